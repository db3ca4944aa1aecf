// Device-side complex helpers and the mixed-radix Stockham FFT that runs entirely in LDS.
// Used by the Pyramid WFS (pyr_kernels.hip) and by the phase-screen generator (screen_kernels.hip).
#pragma once
#include "common.hpp"

namespace ao {

template <typename T> struct cx { T re, im; };
template <typename T> __device__ inline cx<T> cmul(cx<T> a, cx<T> b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
template <typename T> __device__ inline cx<T> cadd(cx<T> a, cx<T> b) { return {a.re + b.re, a.im + b.im}; }
template <typename T> __device__ inline cx<T> csub(cx<T> a, cx<T> b) { return {a.re - b.re, a.im - b.im}; }

template <typename T> __device__ inline void sincos_g(T x, T* s, T* c);
template <> __device__ inline void sincos_g<float>(float x, float* s, float* c) { sincosf(x, s, c); }
template <> __device__ inline void sincos_g<double>(double x, double* s, double* c) { sincos(x, s, c); }

// twiddle w^k, k in [0, n): forward exp(-2 pi i k / n); inverse = conjugate
template <typename T>
__device__ inline cx<T> tw_get(const T* __restrict__ tw, int k, int inverse) {
    cx<T> w = {tw[2 * k], tw[2 * k + 1]};
    if (inverse) w.im = -w.im;
    return w;
}

// One Stockham stage over `nseq` sequences stored [seq][n] in LDS (src -> dst), all lanes of the workgroup cooperate.
template <typename T, int R>
__device__ inline void fft_stage_r(const cx<T>* __restrict__ src, cx<T>* __restrict__ dst, int n, int ns, int nseq,
                                   const T* __restrict__ tw, int inverse) {
    const int m = n / R;
    const int tstep = n / (ns * R);                             // w_{ns R}^{k} = w_n^{k tstep}
    for (int w = threadIdx.x; w < nseq * m; w += blockDim.x) {
        const int seq = w / m, j = w - seq * m;
        const int k = j % ns;
        const cx<T>* s = src + seq * n;
        cx<T> v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v[r] = s[j + r * m];
            if (r > 0) v[r] = cmul(v[r], tw_get(tw, (k * r * tstep) % n, inverse));
        }
        cx<T> y[R];
        if (R == 2) {
            y[0] = cadd(v[0], v[1]);
            y[1] = csub(v[0], v[1]);
        } else if (R == 4) {
            const cx<T> a = cadd(v[0], v[2]), b = csub(v[0], v[2]), c = cadd(v[1], v[3]), d = csub(v[1], v[3]);
            // forward: multiply d by -i ; inverse: by +i
            const cx<T> dj = inverse ? cx<T>{-d.im, d.re} : cx<T>{d.im, -d.re};
            y[0] = cadd(a, c);
            y[1] = cadd(b, dj);
            y[2] = csub(a, c);
            y[3] = csub(b, dj);
        } else {
#pragma unroll
            for (int q = 0; q < R; ++q) {
                cx<T> acc = v[0];
#pragma unroll
                for (int r = 1; r < R; ++r) acc = cadd(acc, cmul(v[r], tw_get(tw, ((q * r) % R) * (n / R), inverse)));
                y[q] = acc;
            }
        }
        cx<T>* d = dst + seq * n + (j / ns) * ns * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) d[q * ns] = y[q];
    }
}

// any other (prime) radix: direct butterfly straight from LDS, no register arrays
template <typename T>
__device__ inline void fft_stage_any(const cx<T>* __restrict__ src, cx<T>* __restrict__ dst, int n, int ns, int nseq,
                                     int R, const T* __restrict__ tw, int inverse) {
    const int m = n / R;
    const int tstep = n / (ns * R);
    for (int w = threadIdx.x; w < nseq * m * R; w += blockDim.x) {
        const int q = w % R, w2 = w / R;
        const int seq = w2 / m, j = w2 - seq * m;
        const int k = j % ns;
        const cx<T>* s = src + seq * n;
        cx<T> acc = {0, 0};
        for (int r = 0; r < R; ++r) {
            const int idx = ((k * r * tstep) % n + ((q * r) % R) * (n / R)) % n;
            acc = cadd(acc, cmul(s[j + r * m], tw_get(tw, idx, inverse)));
        }
        dst[seq * n + (j / ns) * ns * R + k + q * ns] = acc;
    }
}

// Full 1-D transform of `nseq` sequences; returns the buffer that holds the result (a or b).
template <typename T>
__device__ inline cx<T>* fft_lds(cx<T>* a, cx<T>* b, const FftPlan& pl, int nseq, const T* __restrict__ tw, int inverse) {
    int ns = 1;
    cx<T>* src = a;
    cx<T>* dst = b;
    for (int s = 0; s < pl.n_fac; ++s) {
        const int R = pl.fac[s];
        __syncthreads();
        switch (R) {
            case 2: fft_stage_r<T, 2>(src, dst, pl.n, ns, nseq, tw, inverse); break;
            case 3: fft_stage_r<T, 3>(src, dst, pl.n, ns, nseq, tw, inverse); break;
            case 4: fft_stage_r<T, 4>(src, dst, pl.n, ns, nseq, tw, inverse); break;
            case 5: fft_stage_r<T, 5>(src, dst, pl.n, ns, nseq, tw, inverse); break;
            default: fft_stage_any<T>(src, dst, pl.n, ns, nseq, R, tw, inverse); break;
        }
        ns *= R;
        cx<T>* t = src;
        src = dst;
        dst = t;
    }
    __syncthreads();
    return src;
}

}  // namespace ao
