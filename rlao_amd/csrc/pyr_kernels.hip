// Pyramid wave-front sensor: zero-padded 2-D FFT -> 4-facet phase mask -> inverse 2-D FFT -> |.|^2 summed over the
// modulation points -> camera binning -> quadrant slopes maps.
//   OOPAO/Pyramid.py:469-504  pyramid_transform      :516-607  wfs_measure (single wave-front, modulation loop)
//   OOPAO/Pyramid.py:987-1006 camera binning         :682-725, 774-790  signalProcessing / grabQuadrant
//
// nRes = (2 nSub + sep + 2 edge) px is not a power of two (288 = 2^5 3^2 for 20 sub-apertures, 528 = 2^4 3 11 for 40)
// and an nRes^2 complex field (0.66 / 2.2 MB) does not fit in LDS, so the 2-D transforms are done as 1-D passes of
// whole rows / columns held in LDS, three kernels per modulation chunk:
//   P1 rows   : build E = amp exp(i(phi + TT_theta)) [x phasor] for a few pupil rows, zero-pad, FFT along x.  Only the
//               R pupil rows are non-zero, so only R of the nRes rows are transformed and stored (T1: R x nRes).
//   P2 columns: FFT along y of a few columns (zero-padded from R to nRes), fftshift + mask multiply in LDS, and --
//               since the column is already resident -- the inverse FFT along y right away (T2: nRes x nRes).
//   P3 rows   : inverse FFT along x of the nRes/cam rows that feed one camera row, |.|^2 / nRes^4, sum over the
//               modulation points of the chunk, bin to the camera row, accumulate into the frame.
// The 1-D FFT is a Stockham autosort with the radix list chosen on the host (4, 2, 3, 5, then any prime factor by a
// direct butterfly), twiddles from a table w_N^k computed in float64.
#include "common.hpp"
#include "fft.hpp"

namespace ao {

// LDS budget of the two sequence buffers of an FFT workgroup: ~40 KiB keeps 3-4 workgroups (12-16 waves) on a CU, which is what
// hides the LDS / barrier latency of the Stockham stages (at 64 KiB two workgroups = 2 waves per SIMD ran in lock-step)
constexpr size_t kFftLdsTarget = 40 * 1024;

// P1: grid = (ceil(R / RB), chunk, E)
template <typename T, int NFIX>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? 4 : 1) k_pyr_rows(const PyrArgs<T> a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int N = a.N, NP = a.plan.np, R = a.R, RB = a.seq_per_block;
    cx<T>* A = reinterpret_cast<cx<T>*>(lds_raw);
    cx<T>* B = A + RB * NP;
    cx<T>* twl = B + RB * NP;
    fft_load_twiddles<T>(twl, a.tw, N);
    const int e = blockIdx.z, th = blockIdx.y, y0 = blockIdx.x * RB;
    const int nrow = min(RB, R - y0);
    const T* ph = a.phase + (size_t)e * R * R;
    const T* tt = a.tt ? a.tt + (size_t)(a.theta0 + th) * R * R : nullptr;
    const T pi_over_n = (T)(3.14159265358979323846 / N);
    // (loops are (sequence, element) nests: an integer division by a run-time N per element costs as much as the butterflies)
    const int two_n = 2 * N;
    for (int r = 0; r < RB; ++r)
    for (int xg = threadIdx.x; xg < N; xg += blockDim.x) {
        cx<T> v = {0, 0};
        const int x = xg - a.off;
        if (r < nrow && x >= 0 && x < R) {
            const int p = (y0 + r) * R + x;
            const T am = a.amp[p];
            if (am != (T)0) {
                T ang = ph[p];
                if (tt) ang += tt[p];
                // centred mask: the field is multiplied by exp(-i pi (N+1)/N (x + y)) on the padded grid (Pyramid.py:294, 486)
                // the angle pi (N+1) k / N is reduced mod 2 pi in integers (k up to 2N would cost float32 1e-4 rad)
                const T pang = a.phasor_mult ? pi_over_n * (T)((a.phasor_mult * (xg + y0 + r + a.off)) % two_n) : (T)0;
                T s, c;
                if (sizeof(T) == 8 && a.phasor_mult) {
                    // float64: field and phasor multiplied as two complex numbers, as the reference does (support * phasor).
                    // Folding the phasor's angle into the phase rounds it to ulp(|phase|): the WFS calibration measures a
                    // wave-front of 1 m of piston (Pyramid.py:462-466; 8e6 rad, ulp 1e-9 rad), which showed as 1e-9 in the
                    // reference slopes and in the interaction matrix.
                    T ps, pc;
                    sincos_g<T>(ang, &s, &c);
                    sincos_g<T>(pang, &ps, &pc);
                    v = {am * (c * pc + s * ps), am * (s * pc - c * ps)};
                } else {
                    sincos_g<T>(ang - pang, &s, &c);
                    v = {am * c, am * s};
                }
            }
        }
        A[r * NP + fpad(xg)] = v;
    }
    cx<T>* out = fft_any<T, NFIX>(A, B, a.plan, RB, twl, 0);
    cx<T>* t1 = a.t1 + (((size_t)e * a.n_theta_chunk + th) * R + y0) * N;
    for (int r = 0; r < nrow; ++r)
        for (int x = threadIdx.x; x < N; x += blockDim.x) t1[(size_t)r * N + x] = out[r * NP + fpad(x)];
}

// P2: grid = (N / CB, chunk, E); CB columns per workgroup
template <typename T, int NFIX>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? 4 : 1) k_pyr_cols(const PyrArgs<T> a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int N = a.N, NP = a.plan.np, R = a.R, CB = a.seq_per_block;
    cx<T>* A = reinterpret_cast<cx<T>*>(lds_raw);
    cx<T>* B = A + CB * NP;
    cx<T>* twl = B + CB * NP;
    fft_load_twiddles<T>(twl, a.tw, N);
    // Column block of this workgroup.  gridDim.x is a multiple of 8 and workgroups go round-robin over the 8 XCDs, so
    // blockIdx.x % 8 is the XCD: each XCD takes a contiguous range of column blocks.  A 128-byte line of T1 / T2 holds the
    // columns of 16 / CB neighbouring blocks: on one XCD they share its L2 (one HBM read of the line, partial writes merged).
    const int nblk = N / CB, per_xcd = gridDim.x / 8;
    const int blk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (blk >= nblk) return;                                      // padding of the grid (uniform for the workgroup)
    const int e = blockIdx.z, th = blockIdx.y, kx0 = blk * CB;
    const cx<T>* t1 = a.t1 + ((size_t)e * a.n_theta_chunk + th) * R * N;
    // gather: sequence c = column kx0 + c, element y (zero outside the pupil rows)
    for (int i = threadIdx.x; i < CB * NP; i += blockDim.x) A[i] = {0, 0};
    __syncthreads();
    for (int i0 = threadIdx.x; i0 < R * CB; i0 += 4 * (int)blockDim.x) {       // batches of 4 independent loads per lane
        cx<T> v[4];
        int yy[4], cc[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * (int)blockDim.x < R * CB ? i0 + u * (int)blockDim.x : i0;
            yy[u] = CB == 1 ? i : fastdiv(i, a.magic_seq);            // lanes along the columns: contiguous in T1
            cc[u] = i - yy[u] * CB;
            v[u] = t1[(size_t)yy[u] * N + kx0 + cc[u]];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * (int)blockDim.x < R * CB)
                A[cc[u] * NP + fpad(a.off + yy[u])] = v[u];          // (odd sequence stride NP: the CB lanes hit CB banks)
    }
    cx<T>* f = fft_any<T, NFIX>(A, B, a.plan, CB, twl, 0);
    cx<T>* g = (f == A) ? B : A;
    // focal plane: [fftshift] + mask   (Pyramid.py:486-497).  Shifted position i holds frequency (i + N/2) mod N.
    const int h = a.centering ? 0 : N / 2;
    for (int c = 0; c < CB; ++c) {
        int jx = kx0 + c + h;                                     // output (shifted) column of frequency kx0 + c
        jx = jx >= N ? jx - N : jx;
        for (int ky = threadIdx.x; ky < N; ky += blockDim.x) {    // output (shifted) row index ky
            const int src_y = ky + h >= N ? ky + h - N : ky + h;
            const cx<T> v = f[c * NP + fpad(src_y)];
            const T* mk = a.mask + 2 * ((size_t)ky * N + jx);
            g[c * NP + fpad(ky)] = cmul(v, cx<T>{mk[0], mk[1]});
        }
    }
    cx<T>* r = fft_any<T, NFIX>(g, f, a.plan, CB, twl, 1);
    cx<T>* t2 = a.t2 + ((size_t)e * a.n_theta_chunk + th) * N * N;
    const int jx0 = (kx0 + h) % N;                                // CB divides N/2: the block's columns stay contiguous
    for (int i = threadIdx.x; i < N * CB; i += blockDim.x) {
        const int ky = CB == 1 ? i : fastdiv(i, a.magic_seq), c = i - ky * CB;
        t2[(size_t)ky * N + jx0 + c] = r[c * NP + fpad(ky)];
    }
}

// Science PSF, second pass: grid = (N / CB, 1, E); forward FFT along y of CB columns of T1, |.|^2 / N^2 with the output
// fftshift.  EMF = fftshift(fft2(ifftshift(E phasor))) / N (OOPAO/Telescope.py:316-319): the input ifftshift only flips
// signs of the spectrum, which |.|^2 does not see.
template <typename T>
__global__ void __launch_bounds__(256) k_psf_cols(const PyrArgs<T> a, T* __restrict__ psf) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int N = a.N, NP = a.plan.np, R = a.R, CB = a.seq_per_block;
    cx<T>* A = reinterpret_cast<cx<T>*>(lds_raw);
    cx<T>* B = A + CB * NP;
    cx<T>* twl = B + CB * NP;
    fft_load_twiddles<T>(twl, a.tw, N);
    const int e = blockIdx.z, kx0 = blockIdx.x * CB;
    const cx<T>* t1 = a.t1 + (size_t)e * R * N;
    for (int i = threadIdx.x; i < CB * NP; i += blockDim.x) A[i] = {0, 0};
    __syncthreads();
    for (int i = threadIdx.x; i < R * CB; i += blockDim.x) {
        const int y = CB == 1 ? i : fastdiv(i, a.magic_seq), c = i - y * CB;
        A[c * NP + fpad(a.off + y)] = t1[(size_t)y * N + kx0 + c];
    }
    cx<T>* f = fft_lds<T>(A, B, a.plan, CB, twl, 0);
    // 2 x 2 sum-binning of |.|^2 (the reference's oversampling quirk, Telescope.py:303-305, 341-343): the PSF is M x M, M = N / 2
    const int h = N / 2, M = N / 2;
    const T scale = (T)1 / ((T)N * (T)N);
    T* out = psf + (size_t)e * M * M;
    const int jx0 = (kx0 + h) % N;                                // CB (even) divides N / 2: the block's columns stay contiguous
    for (int i = threadIdx.x; i < M * (CB / 2); i += blockDim.x) {
        const int k2 = i / (CB / 2), c2 = i - k2 * (CB / 2);     // output row k2, column (jx0 / 2) + c2
        T acc = 0;
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) {
                const cx<T> v = f[(2 * c2 + dx) * NP + fpad((2 * k2 + dy + h) % N)];   // shifted row ky holds frequency (ky + h) mod N
                acc += (v.re * v.re + v.im * v.im) * scale;
            }
        out[(size_t)k2 * M + jx0 / 2 + c2] = acc;
    }
}

template <typename T>
int launch_psf(const PyrArgs<T>& base, T* psf, hipStream_t st) {
    PyrArgs<T> a = base;
    const int N = a.N, R = a.R;
    const int NP = a.plan.np;
    int rb = (int)(kFftLdsTarget / (2 * (size_t)NP * sizeof(cx<T>)));
    if (rb < 2) rb = (int)std::min<size_t>(2, 160 * 1024 / ((2 * (size_t)NP + N) * sizeof(cx<T>)));   // the column pass bins pairs
    if (rb < 1) return fail("psf: N = %d does not fit two LDS row buffers", N);
    rb = rb > 8 ? 8 : rb;
    int cb = rb & ~1;
    while (cb > 2 && (N / 2) % cb) cb -= 2;
    if (cb < 2 || (N / 2) % cb || (N / 2) % 2) return fail("psf: N = %d has no even column block", N);
    const size_t lds1 = (size_t)(2 * rb * NP + N) * sizeof(cx<T>), lds2 = (size_t)(2 * cb * NP + N) * sizeof(cx<T>);
    if (lds1 > 64 * 1024)
        AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_rows<T, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    if (lds2 > 64 * 1024)
        AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_psf_cols<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    a.theta0 = 0;
    a.n_theta_chunk = 1;
    a.seq_per_block = rb;
    hipLaunchKernelGGL((k_pyr_rows<T, 0>), dim3(cdiv(R, rb), 1, a.n_env), dim3(256), lds1, st, a);
    a.seq_per_block = cb;
    a.magic_seq = fft_magic((unsigned)cb);
    hipLaunchKernelGGL(k_psf_cols<T>, dim3(N / cb, 1, a.n_env), dim3(256), lds2, st, a, psf);
    AO_HIP(hipGetLastError());
    return 0;
}
template int launch_psf<float>(const PyrArgs<float>&, float*, hipStream_t);
template int launch_psf<double>(const PyrArgs<double>&, double*, hipStream_t);

// P3: grid = (cam, E); the nb = N / cam rows of one camera row, all modulation points of the chunk
template <typename T, int NFIX>
__global__ void __launch_bounds__(256, sizeof(T) == 4 ? 4 : 1) k_pyr_rows_inv(const PyrArgs<T> a, int accumulate) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    // the nb rows of a camera row go through LDS in sub-batches of SB = seq_per_block rows (two buffers of SB sequences:
    // ~28 KB instead of 2 nb rows = 54 KB at nRes 528, i.e. 5 resident workgroups per CU instead of 2 -- the kernel waits on
    // barriers and on the HBM read of its rows, not on arithmetic)
    const int N = a.N, NP = a.plan.np, nb = N / a.cam, SB = a.seq_per_block;
    cx<T>* A = reinterpret_cast<cx<T>*>(lds_raw);
    cx<T>* B = A + SB * NP;
    cx<T>* twl = B + SB * NP;
    T* acc = reinterpret_cast<T*>(twl + N);                       // [N] column sums of |.|^2 over the nb rows and the chunk
    fft_load_twiddles<T>(twl, a.tw, N);
    const int e = blockIdx.y, cr = blockIdx.x;
    for (int i = threadIdx.x; i < N; i += blockDim.x) acc[i] = (T)0;
    const T scale = (T)1 / ((T)N * (T)N * (T)N * (T)N);            // ifft2 normalisation 1/N^2 on the amplitude
    for (int th = 0; th < a.n_theta_chunk; ++th) {
        const cx<T>* t2 = a.t2 + (((size_t)e * a.n_theta_chunk + th) * N + (size_t)cr * nb) * N;
        for (int q0 = 0; q0 < nb; q0 += SB) {
            const int ns = min(SB, nb - q0);
            __syncthreads();
            for (int q = 0; q < SB; ++q)
                for (int x = threadIdx.x; x < N; x += blockDim.x)
                    A[q * NP + fpad(x)] = q < ns ? t2[(size_t)(q0 + q) * N + x] : cx<T>{0, 0};
            cx<T>* r = fft_any<T, NFIX>(A, B, a.plan, SB, twl, 1);
            for (int x = threadIdx.x; x < N; x += blockDim.x) {
                T s = 0;
                for (int q = 0; q < ns; ++q) {
                    const cx<T> v = r[q * NP + fpad(x)];
                    s += (v.re * v.re + v.im * v.im) * scale;
                }
                acc[x] += s;
            }
        }
    }
    __syncthreads();
    T* fr = a.frame + (size_t)e * a.cam * a.cam + (size_t)cr * a.cam;
    for (int c = threadIdx.x; c < a.cam; c += blockDim.x) {
        T s = 0;
        for (int q = 0; q < nb; ++q) s += acc[c * nb + q];
        fr[c] = accumulate ? fr[c] + s : s;
    }
}

// Slopes maps, one workgroup per env (Pyramid.py:703-725 'slopesMaps_incidence_flux', :685-701 'slopesMaps').
template <typename T>
__global__ void __launch_bounds__(256) k_pyr_slopes(const PyrSlopeArgs<T> a) {
    __shared__ double red[4];
    const int e = blockIdx.x;
    const T* fr = a.frame + (size_t)e * a.cam * a.cam;
    double s = 0;
    if (!a.norm_valid_mean) {
        for (int i = threadIdx.x; i < a.cam * a.cam; i += blockDim.x) s += (double)fr[i];
    } else {
        for (int k = threadIdx.x; k < a.n_valid; k += blockDim.x) {
            const int r = a.valid_idx[k] / a.n_sub, c = a.valid_idx[k] % a.n_sub;
            s += (double)fr[(a.q_lo + r) * a.cam + a.q_lo + c] + (double)fr[(a.q_lo + r) * a.cam + a.q_hi + c] +
                 (double)fr[(a.q_hi + r) * a.cam + a.q_hi + c] + (double)fr[(a.q_hi + r) * a.cam + a.q_lo + c];
        }
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = s;
    __syncthreads();
    const double tot = red[0] + red[1] + red[2] + red[3];
    const T norma = (T)(a.norm_valid_mean ? tot / a.n_valid : tot / ((double)a.cam * a.cam));
    T* sg = a.signal + (size_t)e * 2 * a.n_valid;
    for (int k = threadIdx.x; k < a.n_valid; k += blockDim.x) {
        const int r = a.valid_idx[k] / a.n_sub, c = a.valid_idx[k] % a.n_sub;
        const T i1 = fr[(a.q_lo + r) * a.cam + a.q_lo + c];      // grabQuadrant(1): rows lo, cols lo
        const T i2 = fr[(a.q_lo + r) * a.cam + a.q_hi + c];      // (2): rows lo, cols hi
        const T i3 = fr[(a.q_hi + r) * a.cam + a.q_hi + c];      // (3): rows hi, cols hi
        const T i4 = fr[(a.q_hi + r) * a.cam + a.q_lo + c];      // (4): rows hi, cols lo
        const T sx = ((i1 - i2) + i4) - i3;
        const T sy = ((i1 - i4) + i2) - i3;
        sg[k] = (sx / norma - a.ref[k]) * a.units;
        sg[a.n_valid + k] = (sy / norma - a.ref[a.n_valid + k]) * a.units;
    }
}

// NFIX: compile-time plan of the inverse row pass; NFIX12: of the row and column passes (0 = run-time plan)
template <typename T, int NFIX, int NFIX12>
int launch_pyramid_n(const PyrArgs<T>& base, int n_theta, int chunk, hipStream_t st) {
    PyrArgs<T> a = base;
    const int N = a.N, R = a.R;
    // sequences per workgroup: keep the two LDS buffers within 64 KiB (P3 may need more: raised explicitly)
    const int NP = a.plan.np;
    int rb = (int)(kFftLdsTarget / (2 * (size_t)NP * sizeof(cx<T>)));
    rb = rb < 1 ? (int)(160 * 1024 / ((2 * (size_t)NP + N) * sizeof(cx<T>))) : rb;
    if (rb < 1) return fail("pyramid: nRes = %d does not fit two LDS row buffers", N);
    rb = rb > 8 ? 8 : rb;
    int cb = rb;
    while (cb > 1 && (N / 2) % cb) --cb;                          // CB must divide N/2 (fftshift keeps a block's columns contiguous)
    const int nb = N / a.cam;
    int sb = (int)(28 * 1024 / (2 * (size_t)NP * sizeof(cx<T>)));   // P3 waits on its HBM rows: 5 workgroups per CU
    sb = sb < 1 ? 1 : (sb > nb ? nb : sb);
    while (sb > 1 && nb % sb) --sb;                               // equal sub-batches of the nb rows of a camera row
    const size_t lds3 = (size_t)(2 * sb * NP + N) * sizeof(cx<T>) + (size_t)N * sizeof(T);
    const size_t lds1 = (size_t)(2 * rb * NP + N) * sizeof(cx<T>), lds2 = (size_t)(2 * cb * NP + N) * sizeof(cx<T>);
    if (lds1 > 64 * 1024)
        AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_rows<T, NFIX12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    if (lds2 > 64 * 1024)
        AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_cols<T, NFIX12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
    if (lds3 > 160 * 1024) return fail("pyramid: %d rows of nRes = %d per camera row do not fit in LDS", nb, N);
    if (lds3 > 64 * 1024)
        AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_pyr_rows_inv<T, NFIX>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3));
    for (int t0 = 0; t0 < n_theta; t0 += chunk) {
        a.theta0 = t0;
        a.n_theta_chunk = (n_theta - t0) < chunk ? (n_theta - t0) : chunk;
        a.seq_per_block = rb;
        hipLaunchKernelGGL((k_pyr_rows<T, NFIX12>), dim3(cdiv(R, rb), a.n_theta_chunk, a.n_env), dim3(256), lds1, st, a);
        a.seq_per_block = cb;
        a.magic_seq = fft_magic((unsigned)cb);
        hipLaunchKernelGGL((k_pyr_cols<T, NFIX12>), dim3(cdiv(N / cb, 8) * 8, a.n_theta_chunk, a.n_env), dim3(256), lds2, st, a);
        a.seq_per_block = sb;
        hipLaunchKernelGGL((k_pyr_rows_inv<T, NFIX>), dim3(a.cam, a.n_env), dim3(256), lds3, st, a, t0 > 0 ? 1 : 0);
        AO_HIP(hipGetLastError());
    }
    return 0;
}

// the float32 transforms of the two lengths the reference's configurations use have compile-time plans (fft.hpp).  At 528 the
// row / column passes keep the run-time plan: fully unrolled around the radix-11 butterfly they need 194 registers (2 waves per
// SIMD instead of 4: measured 5.7 -> 7.4 ms per step at 1024 envs) or spill.
template <typename T>
int launch_pyramid(const PyrArgs<T>& base, int n_theta, int chunk, hipStream_t st) {
    if constexpr (sizeof(T) == 4) {
        if (pyramid528_supported(base)) return launch_pyramid528(base, n_theta, chunk, st);
    }
    if (sizeof(T) == 4 && base.N == 528 && base.plan.n_fac == 3) return launch_pyramid_n<T, 528, 0>(base, n_theta, chunk, st);
    if (sizeof(T) == 4 && base.N == 288 && base.plan.n_fac == 4) return launch_pyramid_n<T, 288, 288>(base, n_theta, chunk, st);
    return launch_pyramid_n<T, 0, 0>(base, n_theta, chunk, st);
}

template <typename T>
int launch_pyramid_slopes(const PyrSlopeArgs<T>& sl, int n_env, hipStream_t st) {
    hipLaunchKernelGGL(k_pyr_slopes<T>, dim3(n_env), dim3(256), 0, st, sl);
    AO_HIP(hipGetLastError());
    return 0;
}
template int launch_pyramid_slopes<float>(const PyrSlopeArgs<float>&, int, hipStream_t);
template int launch_pyramid_slopes<double>(const PyrSlopeArgs<double>&, int, hipStream_t);

template int launch_pyramid<float>(const PyrArgs<float>&, int, int, hipStream_t);
template int launch_pyramid<double>(const PyrArgs<double>&, int, int, hipStream_t);

// radix list for the Stockham transform: 16s and 4s first, then 2, 3, 5, then whatever prime factors remain
int make_fft_plan(int n, FftPlan* pl) {
    pl->n = n;
    pl->np = (fpad(n - 1) + 1) | 1;
    pl->n_fac = 0;
    int m = n;
    auto push = [&](int r) { if (pl->n_fac < 12) pl->fac[pl->n_fac++] = r; };
    while (m % 16 == 0) { push(16); m /= 16; }                    // register-resident 4 x 4 butterflies: half the stages (barriers)
    while (m % 4 == 0) { push(4); m /= 4; }
    while (m % 2 == 0) { push(2); m /= 2; }
    while (m % 3 == 0) { push(3); m /= 3; }
    while (m % 5 == 0) { push(5); m /= 5; }
    for (int p = 7; m > 1; p += 2)
        while (m % p == 0) { push(p); m /= p; }
    int prod = 1;
    for (int i = 0; i < pl->n_fac; ++i) {
        pl->magic_ns[i] = fft_magic((unsigned)prod);
        pl->magic_m[i] = fft_magic((unsigned)(n / pl->fac[i]));
        prod *= pl->fac[i];
    }
    if (n > 8192) return fail("FFT length %d too long for the 16-bit index arithmetic", n);
    return prod == n ? 0 : fail("cannot factor the FFT length %d into at most 12 stages", n);
}

}  // namespace ao
