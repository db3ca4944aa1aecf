// Pyramid passes for nRes = 528 in float32 (the 40x40 Pyramid of BASELINE configs[2], OOPAO/Pyramid.py:469-607, 987-1006) on the
// register-resident 24 x 22 transform of fft528.hpp.  Same three passes, same T1 / T2 layouts and the same arithmetic on the
// pupil field as pyr_kernels.hip (which keeps every other length, float64 and the science PSF); what changes is how a
// 528-point transform is carried out: a lane holds a whole 24- or 22-point factor, a sequence crosses LDS once per transform
// (the Stockham version: three times), and global memory is read into and written from the registers that hold the factors.
//
// A workgroup is 192 lanes = 8 sequences x 24 lanes: "role A" is (sequence, n2 or m1 < 22) -- 176 lanes, the 24-point factor --
// and "role B" is (sequence, k1 < 24), the 22-point factor.
//   P1 rows     : role A builds the field of 8 pupil rows (x = 22 n1 + n2 - off), 24-point DFT, twiddle, exchange; role B 22-point
//                 DFT and stores X[k1 + 24 k2]: 192 contiguous bytes per sequence and store.
//   P2 columns  : role A loads 8 neighbouring columns of T1 (64 contiguous bytes per row), 24-point DFT, twiddle, exchange; role B
//                 22-point DFT, fftshift (k2 -> k2 + 11: a renaming) and mask, inverse 22-point DFT, twiddle, exchange; role A
//                 inverse 24-point DFT and stores T2.  Two exchanges for two transforms.
//   P3 rows^-1  : role B loads rows of T2, inverse 22-point DFT, twiddle, exchange; role A inverse 24-point DFT, |.|^2, summed over
//                 the rows of a camera row and the modulation points in LDS, binned to the camera row.
// LDS layouts are chosen per pass so that the exchange is conflict-free for the lane order that keeps global accesses
// contiguous (bank rules of ds_write_b64 / ds_read_b64, MI355X_MICROARCH.md; scripts/lds_banks_528.py counts them).
#include "common.hpp"
#include "fft.hpp"
#include "fft528.hpp"

namespace ao {

using f528::v2;
constexpr int kLanes528 = 192, kTws = 23;                         // twiddle table [k1][n2] with rows of 23 (odd: see P2's inverse reads)

// w_528^(k1 n2), k1 < 24, n2 < 22, from the N-entry table of the env (k1 n2 <= 483 < 528)
__device__ inline void load_tws(v2* __restrict__ tws, const float* __restrict__ tw, int tid) {
    const v2* t = reinterpret_cast<const v2*>(tw);
    for (int i = tid; i < 24 * kTws; i += kLanes528) {
        const int k1 = i / kTws, n2 = i - kTws * k1;
        tws[i] = t[n2 < 22 ? k1 * n2 : 0];
    }
}

// ---- P1: grid = (ceil(R / 8), chunk, E) -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kLanes528) k_pyr528_rows(const PyrArgs<float> a) {
    constexpr int N = f528::kN, SEQ = 550, S2 = 25;               // ex[c][n2][k1]: rows of 25 (odd) -> conflict-free writes
    __shared__ v2 ex[8 * SEQ];
    __shared__ v2 tws[24 * kTws];
    const int tid = threadIdx.x, R = a.R, off = a.off;
    load_tws(tws, a.tw, tid);
    const int e = blockIdx.z, th = blockIdx.y, y0 = blockIdx.x * 8;
    const int n1_lo = off / 22, n1_hi = (off + R - 1) / 22;       // the 24-point inputs that can be inside the pupil, for any lane
    if (tid < 176) {
        const int c = tid / 22, n2 = tid - 22 * c, row = y0 + c;
        const float* ph = a.phase + (size_t)e * R * R;
        const float* tt = a.tt ? a.tt + (size_t)(a.theta0 + th) * R * R : nullptr;
        const float pi_over_n = (float)(3.14159265358979323846 / N);
        v2 v[24];
#pragma unroll
        for (int n1 = 0; n1 < 24; ++n1) {
            v[n1] = v2{0.f, 0.f};
            if (n1 >= n1_lo && n1 <= n1_hi) {
                const int xg = 22 * n1 + n2, x = xg - off;
                if (row < R && (unsigned)x < (unsigned)R) {
                    const int p = row * R + x;
                    const float am = a.amp[p];
                    if (am != 0.f) {
                        float ang = ph[p];
                        if (tt) ang += tt[p];
                        // centred mask: exp(-i pi (N+1)/N (x + y)) on the padded grid, angle reduced mod 2 pi in integers (k_pyr_rows)
                        const float pang = a.phasor_mult ? pi_over_n * (float)((a.phasor_mult * (xg + row + off)) % (2 * N)) : 0.f;
                        float s, co;
                        sincosf(ang - pang, &s, &co);
                        v[n1] = v2{am * co, am * s};
                    }
                }
            }
        }
        f528::dft24<false>(v);
#pragma unroll
        for (int k1 = 1; k1 < 24; ++k1) v[k1] = f528::cmul_tw<false>(v[k1], tws[k1 * kTws + n2]);
#pragma unroll
        for (int k1 = 0; k1 < 24; ++k1) ex[c * SEQ + n2 * S2 + k1] = v[k1];
    }
    __syncthreads();
    {
        const int c = tid / 24, k1 = tid - 24 * c, row = y0 + c;
        v2 u[22];
#pragma unroll
        for (int n2 = 0; n2 < 22; ++n2) u[n2] = ex[c * SEQ + n2 * S2 + k1];
        f528::dft22<false>(u);
        if (row < R) {
            v2* t1 = reinterpret_cast<v2*>(a.t1) + (((size_t)e * a.n_theta_chunk + th) * R + row) * N + k1;
#pragma unroll
            for (int k2 = 0; k2 < 22; ++k2) t1[24 * k2] = u[k2];
        }
    }
}

// ---- P2: grid = (72 = 66 column blocks padded to a multiple of 8, chunk, E) --------------------------------------------------------
template <bool SHIFT>
__global__ void __launch_bounds__(kLanes528) k_pyr528_cols(const PyrArgs<float> a) {
    constexpr int N = f528::kN, SF = 200, SI = 184;               // ex[n2][k1][c] rows of 200, then ex[k1][m1][c] rows of 184 (both = 8 mod 16)
    __shared__ v2 ex[24 * SI];                                    // 4416 >= 22 * 200
    __shared__ v2 tws[24 * kTws];
    const int tid = threadIdx.x, c = tid & 7, j = tid >> 3, R = a.R, off = a.off;
    load_tws(tws, a.tw, tid);
    // blockIdx.x % 8 is the XCD: each XCD takes a contiguous range of column blocks (k_pyr_cols)
    const int nblk = N / 8, per_xcd = gridDim.x / 8;
    const int blk = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (blk >= nblk) return;                                      // padding of the grid (uniform for the workgroup)
    const int e = blockIdx.z, th = blockIdx.y, kx0 = blk * 8;
    const int jx0 = SHIFT ? (kx0 + N / 2) % N : kx0;              // 8 divides N / 2: the block's shifted columns stay contiguous
    const int n1_lo = off / 22, n1_hi = (off + R - 1) / 22;
    v2 v[24];
    if (j < 22) {
        const v2* t1 = reinterpret_cast<const v2*>(a.t1) + ((size_t)e * a.n_theta_chunk + th) * R * N + kx0 + c;
#pragma unroll
        for (int n1 = 0; n1 < 24; ++n1) {
            v[n1] = v2{0.f, 0.f};
            if (n1 >= n1_lo && n1 <= n1_hi) {
                const int y = 22 * n1 + j - off;
                if ((unsigned)y < (unsigned)R) v[n1] = t1[(size_t)y * N];
            }
        }
        f528::dft24<false>(v);
#pragma unroll
        for (int k1 = 1; k1 < 24; ++k1) v[k1] = f528::cmul_tw<false>(v[k1], tws[k1 * kTws + j]);
#pragma unroll
        for (int k1 = 0; k1 < 24; ++k1) ex[j * SF + k1 * 8 + c] = v[k1];
    }
    // the mask values of this lane's frequencies: shifted position i holds frequency (i + N/2) mod N, so frequency k1 + 24 k2 sits at
    // i = k1 + 24 ((k2 + 11) mod 22)   (N / 2 = 24 x 11; Pyramid.py:486-497)
    v2 m[22];
    {
        const v2* mk = reinterpret_cast<const v2*>(a.mask) + jx0 + c;
#pragma unroll
        for (int i2 = 0; i2 < 22; ++i2) m[i2] = mk[(size_t)(j + 24 * i2) * N];
    }
    __syncthreads();
    v2 u[22];
#pragma unroll
    for (int n2 = 0; n2 < 22; ++n2) u[n2] = ex[n2 * SF + j * 8 + c];
    f528::dft22<false>(u);
    v2 g[22];
#pragma unroll
    for (int k2 = 0; k2 < 22; ++k2) {
        const int i2 = SHIFT ? (k2 + 11) % 22 : k2;
        g[i2] = f528::cmul2(u[k2], m[i2]);
    }
    f528::dft22<true>(g);
#pragma unroll
    for (int m1 = 1; m1 < 22; ++m1) g[m1] = f528::cmul_tw<true>(g[m1], tws[j * kTws + m1]);
    __syncthreads();                                              // every lane has its forward values
#pragma unroll
    for (int m1 = 0; m1 < 22; ++m1) ex[j * SI + m1 * 8 + c] = g[m1];
    __syncthreads();
    if (j < 22) {
#pragma unroll
        for (int k1 = 0; k1 < 24; ++k1) v[k1] = ex[k1 * SI + j * 8 + c];
        f528::dft24<true>(v);
        v2* t2 = reinterpret_cast<v2*>(a.t2) + ((size_t)e * a.n_theta_chunk + th) * N * N + jx0 + c;
#pragma unroll
        for (int m2 = 0; m2 < 24; ++m2) t2[(size_t)(j + 22 * m2) * N] = v[m2];
    }
}

// ---- P3: grid = (cam / G, E): G camera rows = G nb rows of T2 per modulation point, in batches of 8 sequences ------------------------
__global__ void __launch_bounds__(kLanes528) k_pyr528_rows_inv(const PyrArgs<float> a, int accumulate, int G) {
    constexpr int N = f528::kN, SEQ = 568, S1 = 23;               // ex[c][k1][m1]
    __shared__ v2 ex[8 * SEQ];
    __shared__ v2 tws[24 * kTws];
    __shared__ float acc[4 * N];                                  // [G][N] column sums of |.|^2 over the nb rows and the chunk
    float* pw = reinterpret_cast<float*>(ex);                     // [8][N] |.|^2 of the batch (after the exchange has been read)
    const int tid = threadIdx.x, nb = N / a.cam, chunk = a.n_theta_chunk;
    load_tws(tws, a.tw, tid);
    for (int i = tid; i < G * N; i += kLanes528) acc[i] = 0.f;
    const int e = blockIdx.y, cr0 = blockIdx.x * G;
    const int per_g = chunk * nb, S = G * per_g;                  // sequence s = (g chunk + th) nb + q: row (cr0 + g) nb + q of point th
    const float scale = 1.f / ((float)N * (float)N * (float)N * (float)N);   // ifft2 normalisation 1/N^2 on the amplitude
    const int cb = tid / 24, k1 = tid - 24 * cb;
    const int ca = tid / 22, m1 = tid - 22 * ca;
    for (int s0 = 0; s0 < S; s0 += 8) {
        v2 g[22];
        {
            const int s = s0 + cb;
            const bool valid = s < S;
            const int gi = s / per_g, rem = s - gi * per_g, th = rem / nb, q = rem - th * nb;
            const v2* t2 = reinterpret_cast<const v2*>(a.t2) +
                           (valid ? (((size_t)e * chunk + th) * N + (size_t)(cr0 + gi) * nb + q) * N + k1 : 0);
#pragma unroll
            for (int k2 = 0; k2 < 22; ++k2) g[k2] = valid ? t2[24 * k2] : v2{0.f, 0.f};
        }
        f528::dft22<true>(g);
#pragma unroll
        for (int q1 = 1; q1 < 22; ++q1) g[q1] = f528::cmul_tw<true>(g[q1], tws[k1 * kTws + q1]);
        __syncthreads();                                          // the previous batch's sums are taken (first batch: tables are loaded)
#pragma unroll
        for (int q1 = 0; q1 < 22; ++q1) ex[cb * SEQ + k1 * S1 + q1] = g[q1];
        __syncthreads();
        float p[24];
        if (tid < 176) {
            v2 v[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) v[k] = ex[ca * SEQ + k * S1 + m1];
            f528::dft24<true>(v);
#pragma unroll
            for (int m2 = 0; m2 < 24; ++m2) p[m2] = (v[m2].x * v[m2].x + v[m2].y * v[m2].y) * scale;
        }
        __syncthreads();                                          // pw aliases ex
        if (tid < 176) {
#pragma unroll
            for (int m2 = 0; m2 < 24; ++m2) pw[ca * N + m1 + 22 * m2] = p[m2];
        }
        __syncthreads();
        const int n_in = min(8, S - s0);
        for (int x = tid; x < N; x += kLanes528) {
            int g_prev = s0 / per_g;
            float run = 0.f;
            for (int cc = 0; cc < n_in; ++cc) {
                const int gi = (s0 + cc) / per_g;
                if (gi != g_prev) {
                    acc[g_prev * N + x] += run;
                    run = 0.f;
                    g_prev = gi;
                }
                run += pw[cc * N + x];
            }
            acc[g_prev * N + x] += run;
        }
    }
    __syncthreads();
    float* fr = a.frame + (size_t)e * a.cam * a.cam + (size_t)cr0 * a.cam;
    for (int i = tid; i < G * a.cam; i += kLanes528) {
        const int gi = i / a.cam, cc = i - gi * a.cam;
        float s = 0.f;
        for (int q = 0; q < nb; ++q) s += acc[gi * N + cc * nb + q];
        fr[i] = accumulate ? fr[i] + s : s;
    }
}

// 0: this geometry is not covered (the caller runs the Stockham passes of pyr_kernels.hip)
int pyramid528_supported(const PyrArgs<float>& a) {
    return a.N == f528::kN && a.R <= a.N && a.off >= 0 && a.off + a.R <= a.N && a.cam > 0 && a.N % a.cam == 0 && !a.generic_fft;
}

int launch_pyramid528(const PyrArgs<float>& base, int n_theta, int chunk, hipStream_t st) {
    PyrArgs<float> a = base;
    const int N = a.N, R = a.R;
    const int G = a.cam % 4 == 0 ? 4 : (a.cam % 2 == 0 ? 2 : 1);
    for (int t0 = 0; t0 < n_theta; t0 += chunk) {
        a.theta0 = t0;
        a.n_theta_chunk = (n_theta - t0) < chunk ? (n_theta - t0) : chunk;
        hipLaunchKernelGGL(k_pyr528_rows, dim3(cdiv(R, 8), a.n_theta_chunk, a.n_env), dim3(kLanes528), 0, st, a);
        const dim3 g2(cdiv(N / 8, 8) * 8, a.n_theta_chunk, a.n_env);
        if (a.centering)
            hipLaunchKernelGGL(k_pyr528_cols<false>, g2, dim3(kLanes528), 0, st, a);
        else
            hipLaunchKernelGGL(k_pyr528_cols<true>, g2, dim3(kLanes528), 0, st, a);
        hipLaunchKernelGGL(k_pyr528_rows_inv, dim3(a.cam / G, a.n_env), dim3(kLanes528), 0, st, a, t0 > 0 ? 1 : 0, G);
        AO_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace ao
