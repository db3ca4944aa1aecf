// Shack-Hartmann kernels (diffractive, single wavefront per env).
//   OOPAO/ShackHartmann.py:340-347  lenslet fields: tile phase.T, embed p x p in 2p x 2p, x sqrt(flux) x phasor
//   OOPAO/ShackHartmann.py:539      I = |FFT2(E) / n|^2            (n = 2p)
//   OOPAO/ShackHartmann.py:565      2 x 2 sum-binning to p x p     (the shannon crop at :560 is overwritten)
//   OOPAO/ShackHartmann.py:349-353  camera frame assembly          (noise-free detector == identity)
//   OOPAO/ShackHartmann.py:314-324  centroid: threshold at thr * max over ALL valid spots, centre of gravity
//   OOPAO/ShackHartmann.py:583-601  NaN -> 0, reference subtraction, slope units, valid selection
#include "common.hpp"

namespace ao {

template <typename T> struct cplx { T re, im; };

template <typename T> __device__ inline void sincos_t(T x, T* s, T* c);
template <> __device__ inline void sincos_t<float>(float x, float* s, float* c) { sincosf(x, s, c); }
template <> __device__ inline void sincos_t<double>(double x, double* s, double* c) { sincos(x, s, c); }

__device__ inline void atomic_max_nonneg(float* addr, float v) {
    atomicMax(reinterpret_cast<unsigned int*>(addr), __float_as_uint(v));
}
__device__ inline void atomic_max_nonneg(double* addr, double v) {
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

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

template <typename T>
int launch_sh_spots(const T* phase, const ShConst<T>& sc, T* frame, T* wfs_max, int n_env, int R, int n_subap,
                    int n_valid, hipStream_t st) {
    const int p = R / n_subap, n = 2 * p;
    const size_t lds = sizeof(cplx<T>) * (n + 4 * (p * p + p * n));
    if (lds > 64 * 1024) return fail("sh_spots: %d px per lenslet needs %zu B of LDS", p, lds);
    dim3 grid(cdiv(n_valid, 4), n_env);
    hipLaunchKernelGGL(k_sh_spots<T>, grid, dim3(256), lds, st, phase, sc, frame, wfs_max, R, n_subap, n_valid);
    AO_HIP(hipGetLastError());
    return 0;
}

// One workgroup per env; one lane per valid lenslet.
template <typename T>
__global__ void __launch_bounds__(256) k_sh_centroid(const T* __restrict__ frame, const T* __restrict__ wfs_max,
                                                     const ShConst<T> sc, T* __restrict__ signal, int R, int n_subap,
                                                     int n_valid, int max_group, int n_env) {
    const int e = blockIdx.x;
    const int p = R / n_subap;
    // envs of one measurement batch share the threshold maximum (ShackHartmann.py:605-660)
    const int g0 = (e / max_group) * max_group;
    const int g1 = min(g0 + max_group, n_env);
    T mx = 0;
    for (int q = g0; q < g1; ++q) mx = wfs_max[q] > mx ? wfs_max[q] : mx;
    const T cut = sc.threshold * mx;
    const T* fr = frame + (size_t)e * R * R;
    T* sg = signal + (size_t)e * 2 * n_valid;
    for (int s = threadIdx.x; s < n_valid; s += blockDim.x) {
        const int k = sc.subap_idx[s], i = k / n_subap, j = k % n_subap;
        T norm = 0, m0 = 0, m1 = 0;
        for (int u = 0; u < p; ++u) {
            const T* row = fr + (size_t)(i * p + u) * R + j * p;
            for (int v = 0; v < p; ++v) {
                T x = row[v];
                x = x < cut ? (T)0 : x;
                norm += x;
                m0 += x * (T)u;
                m1 += x * (T)v;
            }
        }
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
    hipLaunchKernelGGL(k_sh_centroid<T>, dim3(n_env), dim3(256), 0, st, frame, wfs_max, sc, signal, R, n_subap,
                       n_valid, max_group, n_env);
    AO_HIP(hipGetLastError());
    return 0;
}

#define INST(T)                                                                                                 \
    template int launch_sh_spots<T>(const T*, const ShConst<T>&, T*, T*, int, int, int, int, hipStream_t);      \
    template int launch_sh_centroid<T>(const T*, const T*, const ShConst<T>&, T*, int, int, int, int, int,      \
                                       hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace ao
