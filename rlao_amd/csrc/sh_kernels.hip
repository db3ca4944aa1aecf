// Shack-Hartmann kernels (diffractive, single wavefront per env).
//   OOPAO/ShackHartmann.py:340-347  lenslet fields: tile phase.T, embed p x p in 2p x 2p, x sqrt(flux) x phasor
//   OOPAO/ShackHartmann.py:539      I = |FFT2(E) / n|^2            (n = 2p)
//   OOPAO/ShackHartmann.py:565      2 x 2 sum-binning to p x p     (the shannon crop at :560 is overwritten)
//   OOPAO/ShackHartmann.py:349-353  camera frame assembly          (noise-free detector == identity)
//   OOPAO/ShackHartmann.py:314-324  centroid: threshold at thr * max over ALL valid spots, centre of gravity
//   OOPAO/ShackHartmann.py:583-601  NaN -> 0, reference subtraction, slope units, valid selection
#include "common.hpp"
#include "sh_device.hpp"

namespace ao {

// One wavefront per valid lenslet, 4 lenslets per workgroup; grid = (ceil(nValid/4), n_env).
// The 2-D DFT of the zero-padded field is evaluated as two small dense products restricted to the
// p x p non-zero block (rows/columns lo .. lo+p-1 of the n x n array):
//     G[a][v] = sum_b E[a][b] w^((b+lo) v)          F[u][v] = sum_a G[a][v] w^((a+lo) u)
// and only the 2x2-binned |F|^2 is kept.  Tile element E[a][b] comes from phase[i p + b][j p + a]
// (the reference tiles phase.T, so the spot image is transposed inside its camera block).
template <typename T>
__global__ void __launch_bounds__(256) k_sh_spots(const T* __restrict__ phase, const ShConst<T> sc,
                                                  T* __restrict__ frame, T* __restrict__ wfs_max, int R, int n_subap,
                                                  int n_valid) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int p = R / n_subap, n = 2 * p, lo = n / 2 - p / 2;
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    cplx<T>* tw = reinterpret_cast<cplx<T>*>(lds_raw);                       // [n]
    cplx<T>* Et = tw + n + wave * (p * p + p * n);                            // [p][p]
    cplx<T>* G = Et + p * p;                                                  // [p][n]
    for (int k = threadIdx.x; k < n; k += blockDim.x) tw[k] = {sc.tw[2 * k], sc.tw[2 * k + 1]};
    __syncthreads();

    const int s = blockIdx.x * 4 + wave;
    const int e = blockIdx.y;
    const bool active = s < n_valid;          // no early return: the barriers below are workgroup-wide
    const int k = active ? sc.subap_idx[s] : 0, i = k / n_subap, j = k % n_subap;
    const T* ph = phase + (size_t)e * R * R;

    for (int idx = lane; active && idx < p * p; idx += kWave) {
        const int a = idx / p, b = idx % p;
        const int pix = (i * p + b) * R + (j * p + a);
        T sn, cs;
        sincos_t<T>(ph[pix], &sn, &cs);
        const T am = sc.amp[pix];
        // phasor exp(-i pi (n+1)/n (x + y)) = ph[a] * ph[b]
        const T pr = sc.ph[2 * a] * sc.ph[2 * b] - sc.ph[2 * a + 1] * sc.ph[2 * b + 1];
        const T pi = sc.ph[2 * a] * sc.ph[2 * b + 1] + sc.ph[2 * a + 1] * sc.ph[2 * b];
        const T er = am * cs, ei = am * sn;
        Et[idx] = {er * pr - ei * pi, er * pi + ei * pr};
    }
    __syncthreads();

    for (int idx = lane; active && idx < p * n; idx += kWave) {
        const int a = idx / n, v = idx % n;
        T gr = 0, gi = 0;
        int t = (lo * v) % n;
        for (int b = 0; b < p; ++b) {
            const cplx<T> x = Et[a * p + b], w = tw[t];
            gr += x.re * w.re - x.im * w.im;
            gi += x.re * w.im + x.im * w.re;
            t += v;
            t = t >= n ? t - n : t;
        }
        G[idx] = {gr, gi};
    }
    __syncthreads();

    T mx = 0;
    T* fr = frame + (size_t)e * R * R;
    for (int idx = lane; active && idx < p * p; idx += kWave) {
        const int P = idx / p, Q = idx % p;
        T acc = 0;
#pragma unroll
        for (int du = 0; du < 2; ++du) {
            const int u = 2 * P + du;
#pragma unroll
            for (int dv = 0; dv < 2; ++dv) {
                const int v = 2 * Q + dv;
                T fr_ = 0, fi_ = 0;
                int t = (lo * u) % n;
                for (int a = 0; a < p; ++a) {
                    const cplx<T> g = G[a * n + v], w = tw[t];
                    fr_ += g.re * w.re - g.im * w.im;
                    fi_ += g.re * w.im + g.im * w.re;
                    t += u;
                    t = t >= n ? t - n : t;
                }
                // |F / n|^2, as the reference normalises the field before squaring
                const T ar = fr_ / (T)n, ai = fi_ / (T)n;
                acc += ar * ar + ai * ai;
            }
        }
        fr[(i * p + P) * R + (j * p + Q)] = acc;
        mx = acc > mx ? acc : mx;
    }
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    if (active && lane == 0) atomic_max_nonneg(&wfs_max[e], mx);
}

// ---------------------------------------------------------------------------------------------------
// Fast path for P = 6 pixels per lenslet (n = 12): every BASELINE configuration.
//
// One wavefront per ROW of lenslets; P/2 = 3 lanes per lenslet, 21 lenslets per pass.  Lane (j, q) owns
// the spectral columns v in {2q, 2q+1, 2q+6, 2q+7}, i.e. the two camera columns Q = q and Q = q+3.
//   stage 0  the wave reads its P x (21 P) strip of phase / amplitude with coalesced row loads, turns it
//            into E0 = amp e^{i phi} and parks it in LDS laid out per lenslet (stride 37: bank-conflict free)
//   stage 1  G[a][v] = sum_b E0[a][b] (ph_b w^{(b+lo) v})     per-lane twiddle registers (depend on q)
//            columns v and v+6 share their products: w^{(b+lo)(v+6)} = (-1)^{b+lo} w^{(b+lo) v}
//   stage 2  F[u][v] = sum_a (ph_a w^{u (a+lo)}) G[a][v]      compile-time twiddles; rows u and u+6 share
//   binning  I[P][Q] = sum_{du,dv} |F[2P+du][2Q+dv]|^2 / n^2  entirely in the lane's registers
//   output   the strip of the camera frame is staged in LDS and written as whole rows; one atomicMax per wave
// All arithmetic after stage 0 is register-resident FMAs (~2.6 kFMA per lenslet instead of 5.2 k), and the
// centring phasor exp(-i pi (n+1)/n (x+y)) = ph_a ph_b is folded into the twiddles.
// ---------------------------------------------------------------------------------------------------

template <typename T, bool FAST_TRIG>
__global__ void __launch_bounds__(128, 2) k_sh_spots_p6(const T* __restrict__ phase, const ShConst<T> sc,
                                                     const uint8_t* __restrict__ valid2d, T* __restrict__ frame,
                                                     T* __restrict__ wfs_max, int R, int n_subap) {
    using namespace fast6;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    constexpr int W = SPW * P;                                        // strip width in pixels
    cplx<T>* Ew = reinterpret_cast<cplx<T>*>(lds_raw) + wave * (SPW * EST);
    T* Fs = reinterpret_cast<T*>(reinterpret_cast<cplx<T>*>(lds_raw) + 2 * (SPW * EST)) + wave * (P * W);

    const int e = blockIdx.y;
    const int i = blockIdx.x * 2 + wave;                              // lenslet row of this wave
    const bool row_ok = i < n_subap;
    const T* ph = phase + (size_t)e * R * R;
    T* fr = frame + (size_t)e * R * R;
    const int jl = lane / HP, q = lane - jl * HP;

    const T ctab = cos_table_lane<T>();
    T mx = 0;
    for (int j0 = 0; j0 < n_subap; j0 += SPW) {
        // a strip without a valid lenslet (the corners of the lenslet array: 7 % of the strips of an 80 x 80 array) is all zeros
        const int j = j0 + jl;
        const bool ok = row_ok && lane < SPW * HP && j < n_subap && valid2d[i * n_subap + j] != 0;
        const bool lit = __any(ok);                                    // (wave-uniform; the barriers below stay unconditional)
        // ---- stage 0: strip of phase -> E0 in LDS (all loads of the lane issued before the first use) ---------
        constexpr int NP = (P * W + kWave - 1) / kWave;                 // 12 pixels per lane
        T phv[NP], amv[NP];
#pragma unroll
        for (int k = 0; k < NP; ++k) {
            const int t = lane + k * kWave;
            const int b = t / W, c = t - b * W;
            const int col = j0 * P + c;
            phv[k] = (T)0;
            amv[k] = (T)0;
            if (lit && t < P * W && col < R) {
                const int pix = (i * P + b) * R + col;
                amv[k] = sc.amp[pix];
                phv[k] = ph[pix];
            }
        }
        if (lit) {
#pragma unroll
            for (int k = 0; k < NP; ++k) {
                const int t = lane + k * kWave;
                if (t < P * W) {
                    const int b = t / W, c = t - b * W;
                    T sn, cs;
                    if (FAST_TRIG) sincos_fast(phv[k], &sn, &cs); else sincos_t<T>(phv[k], &sn, &cs);
                    const int jj = c / P, a = c - jj * P;
                    Ew[jj * EST + a * P + b] = {amv[k] * cs, amv[k] * sn};
                }
            }
        }
        __syncthreads();

        T Ia[P], Ib[P];
#pragma unroll
        for (int u = 0; u < P; ++u) Ia[u] = Ib[u] = (T)0;
        if (lit)                                                       // wave-uniform: the twiddle shuffles need every lane
            lenslet_spots<T>(Ew + (lane < SPW * HP ? jl : 0) * EST, q, ctab, Ia, Ib);
        if (!ok) {
#pragma unroll
            for (int u = 0; u < P; ++u) Ia[u] = Ib[u] = (T)0;
        } else {
#pragma unroll
            for (int u = 0; u < P; ++u) {
                mx = Ia[u] > mx ? Ia[u] : mx;
                mx = Ib[u] > mx ? Ib[u] : mx;
            }
        }
        // ---- stage the camera strip and write whole rows ------------------------------------------------------
        if (lane < SPW * HP) {
#pragma unroll
            for (int u = 0; u < P; ++u) {
                Fs[u * W + jl * P + q] = Ia[u];
                Fs[u * W + jl * P + q + 3] = Ib[u];
            }
        }
        __syncthreads();
        if (row_ok) {
            for (int t = lane; t < P * W; t += kWave) {
                const int u = t / W, c = t - u * W;
                const int col = j0 * P + c;
                if (col < R) fr[(size_t)(i * P + u) * R + col] = Fs[t];
            }
        }
        __syncthreads();
    }
    for (int off = 32; off > 0; off >>= 1) {
        const T o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    if (row_ok && lane == 0) atomic_max_nonneg(&wfs_max[e], mx);
}

template <typename T>
int launch_sh_spots(const T* phase, const ShConst<T>& sc, T* frame, T* wfs_max, int n_env, int R, int n_subap,
                    int n_valid, hipStream_t st) {
    const int p = R / n_subap, n = 2 * p;
    if (p == fast6::P && sc.valid2d != nullptr) {
        const size_t lds6 = 2 * (sizeof(cplx<T>) * fast6::SPW * fast6::EST + sizeof(T) * fast6::P * fast6::SPW * fast6::P);
        dim3 grid6(cdiv(n_subap, 2), n_env);
        if (sc.fast_trig)
            hipLaunchKernelGGL((k_sh_spots_p6<T, true>), grid6, dim3(128), lds6, st, phase, sc, sc.valid2d, frame,
                               wfs_max, R, n_subap);
        else
            hipLaunchKernelGGL((k_sh_spots_p6<T, false>), grid6, dim3(128), lds6, st, phase, sc, sc.valid2d, frame,
                               wfs_max, R, n_subap);
        AO_HIP(hipGetLastError());
        return 0;
    }
    const size_t lds = sizeof(cplx<T>) * (n + 4 * (p * p + p * n));
    if (lds > 64 * 1024) return fail("sh_spots: %d px per lenslet needs %zu B of LDS", p, lds);
    dim3 grid(cdiv(n_valid, 4), n_env);
    hipLaunchKernelGGL(k_sh_spots<T>, grid, dim3(256), lds, st, phase, sc, frame, wfs_max, R, n_subap, n_valid);
    AO_HIP(hipGetLastError());
    return 0;
}

// Centre of gravity.  grid = (ceil(nValid / 32), n_env), 256 lanes: 8 lanes per lenslet (one camera row each for
// p <= 8; they stride over the rows otherwise), so a lane's reads are the p contiguous floats of its row and
// all of them are in flight at once; the three partial sums are folded across the 8 lanes with DPP shuffles.
template <typename T>
__global__ void __launch_bounds__(256) k_sh_centroid(const T* __restrict__ frame, const T* __restrict__ wfs_max,
                                                     const ShConst<T> sc, T* __restrict__ signal, int R, int n_subap,
                                                     int n_valid, int max_group, int n_env) {
    const int e = blockIdx.y;
    const int p = R / n_subap;
    // envs of one measurement batch share the threshold maximum (ShackHartmann.py:605-660)
    const int g0 = (e / max_group) * max_group;
    const int g1 = min(g0 + max_group, n_env);
    T mx = 0;
    for (int q = g0; q < g1; ++q) mx = wfs_max[q] > mx ? wfs_max[q] : mx;
    const T cut = sc.threshold * mx;
    const T* fr = frame + (size_t)e * R * R;
    T* sg = signal + (size_t)e * 2 * n_valid;
    const int s = blockIdx.x * 32 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
    const bool ok = s < n_valid;
    T norm = 0, m0 = 0, m1 = 0;
    if (ok) {
        const int k = sc.subap_idx[s], i = k / n_subap, j = k % n_subap;
        for (int u = sub; u < p; u += 8) {
            const T* row = fr + (size_t)(i * p + u) * R + j * p;
            T rs = 0, rm = 0;
            if (p == 6) {                                          // every BASELINE geometry: the row's 6 pixels requested together
                T xv[6];                                           // (a loop over a run-time p is a chain of 6 memory latencies)
#pragma unroll
                for (int v = 0; v < 6; ++v) xv[v] = row[v];
#pragma unroll
                for (int v = 0; v < 6; ++v) {
                    const T x = xv[v] < cut ? (T)0 : xv[v];
                    rs += x;
                    rm += x * (T)v;
                }
            } else
            for (int v = 0; v < p; ++v) {
                T x = row[v];
                x = x < cut ? (T)0 : x;
                rs += x;
                rm += x * (T)v;
            }
            norm += rs;
            m0 += rs * (T)u;
            m1 += rm;
        }
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) {
        norm += __shfl_down(norm, off, 8);
        m0 += __shfl_down(m0, off, 8);
        m1 += __shfl_down(m1, off, 8);
    }
    if (ok && sub == 0) {
        T c0 = (T)0, c1 = (T)0;
        if (norm != (T)0) {
            c0 = m0 / norm;
            c1 = m1 / norm;
        }
        sg[s] = (c0 - sc.ref[s]) / sc.units;
        sg[n_valid + s] = (c1 - sc.ref[n_valid + s]) / sc.units;
    }
}

template <typename T>
int launch_sh_centroid(const T* frame, const T* wfs_max, const ShConst<T>& sc, T* signal, int n_env, int R,
                       int n_subap, int n_valid, int max_group, hipStream_t st) {
    hipLaunchKernelGGL(k_sh_centroid<T>, dim3(cdiv(n_valid, 32), n_env), dim3(256), 0, st, frame, wfs_max, sc, signal,
                       R, n_subap, n_valid, max_group, n_env);
    AO_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Fused tail: centre of gravity -> obs = -R s -> reward, integrator and telemetry in ONE launch, one workgroup
// (1024 lanes) per env.  The three separate kernels (centroid, split-K GEMM, epilogue) are mostly launch + memory
// latency (~10 us each for < 1 us of work at a few hundred envs).  The reconstructor is low rank by construction,
// R = M2C (A x K) . M (K x nSig) with K = nModes = 50 (MAIN/OOPAOEnv/OOPAOEnv.py:295, 381), so a per-env product
// through the factors costs 4.5x fewer MACs than R s and re-reads 0.2 MB instead of 0.9 MB from L2 per env:
//   t = M s   : 16 lanes per mode, contiguous reads of the mode's row, DPP fold
//   o = M2C t : one lane per actuator, M2C stored transposed [K][A] (coalesced), t in LDS
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024) k_sh_tail(const T* __restrict__ frame, const T* __restrict__ wfs_max,
                                                   const ShConst<T> sc, T* __restrict__ signal,
                                                   const T* __restrict__ fac_m, const T* __restrict__ fac_m2c_t,
                                                   int n_modes, const FinishArgs<T> f, int R, int n_subap,
                                                   int n_valid, int max_group, int n_env) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* sl = reinterpret_cast<T*>(lds_raw);                     // [2 nValid] slopes
    T* img_s = sl + 2 * n_valid;                               // [nAct^2] observation image
    __shared__ double red[16];
    const int e = blockIdx.x, tid = threadIdx.x;
    const int p = R / n_subap;
    const int g0 = (e / max_group) * max_group, g1 = min(g0 + max_group, n_env);
    T mx = 0;
    for (int q = g0; q < g1; ++q) mx = wfs_max[q] > mx ? wfs_max[q] : mx;
    const T cut = sc.threshold * mx;
    const T* fr = frame + (size_t)e * R * R;
    const int img = f.n_act * f.n_act;
    for (int q = tid; q < img; q += blockDim.x) img_s[q] = (T)0;
    // ---- centroids: 2 lanes per lenslet (p/2 camera rows each) ----------------------------------------------------
    for (int w = tid; w < 2 * n_valid; w += blockDim.x) {
        const int s = w >> 1, half = w & 1;
        const int k = sc.subap_idx[s], i = k / n_subap, j = k % n_subap;
        T norm = 0, m0 = 0, m1 = 0;
        for (int u = half; u < p; u += 2) {
            const T* row = fr + (size_t)(i * p + u) * R + j * p;
            T rs = 0, rm = 0;
            if (p == 6) {                                          // every BASELINE geometry: the row's 6 pixels requested together
                T xv[6];                                           // (a loop over a run-time p is a chain of 6 memory latencies)
#pragma unroll
                for (int v = 0; v < 6; ++v) xv[v] = row[v];
#pragma unroll
                for (int v = 0; v < 6; ++v) {
                    const T x = xv[v] < cut ? (T)0 : xv[v];
                    rs += x;
                    rm += x * (T)v;
                }
            } else
            for (int v = 0; v < p; ++v) {
                T x = row[v];
                x = x < cut ? (T)0 : x;
                rs += x;
                rm += x * (T)v;
            }
            norm += rs;
            m0 += rs * (T)u;
            m1 += rm;
        }
        norm += __shfl_xor(norm, 1);
        m0 += __shfl_xor(m0, 1);
        m1 += __shfl_xor(m1, 1);
        if (half == 0) {
            T c0 = (T)0, c1 = (T)0;
            if (norm != (T)0) {
                c0 = m0 / norm;
                c1 = m1 / norm;
            }
            const T s0 = (c0 - sc.ref[s]) / sc.units, s1 = (c1 - sc.ref[n_valid + s]) / sc.units;
            sl[s] = s0;
            sl[n_valid + s] = s1;
            signal[(size_t)e * 2 * n_valid + s] = s0;
            signal[(size_t)e * 2 * n_valid + n_valid + s] = s1;
        }
    }
    __syncthreads();
    tail_from_slopes<T>(sl, img_s, red, fac_m, fac_m2c_t, n_modes, f, e, n_valid, n_env);
}

template <typename T>
int launch_sh_tail(const T* frame, const T* wfs_max, const ShConst<T>& sc, T* signal, const T* fac_m,
                   const T* fac_m2c_t, int n_modes, const FinishArgs<T>& fa, int n_env, int R, int n_subap, int n_valid,
                   int max_group, hipStream_t st) {
    const size_t lds = sizeof(T) * ((size_t)2 * n_valid + (size_t)fa.n_act * fa.n_act + (size_t)n_modes);
    if (lds > 64 * 1024) return -1;
    hipLaunchKernelGGL(k_sh_tail<T>, dim3(n_env), dim3(1024), lds, st, frame, wfs_max, sc, signal, fac_m, fac_m2c_t,
                       n_modes, fa, R, n_subap, n_valid, max_group, n_env);
    AO_HIP(hipGetLastError());
    return 0;
}

#define INST(T)                                                                                                 \
    template int launch_sh_spots<T>(const T*, const ShConst<T>&, T*, T*, int, int, int, int, hipStream_t);      \
    template int launch_sh_centroid<T>(const T*, const T*, const ShConst<T>&, T*, int, int, int, int, int,      \
                                       hipStream_t);                                                            \
    template int launch_sh_tail<T>(const T*, const T*, const ShConst<T>&, T*, const T*, const T*, int,         \
                                   const FinishArgs<T>&, int, int, int, int, int, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace ao
