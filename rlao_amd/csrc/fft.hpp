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

// The twiddle table w_n^k = exp(-2 pi i k / n), k in [0, n), lives in LDS for the transform (twl); inverse = conjugate.
template <typename T>
__device__ inline void fft_load_twiddles(cx<T>* __restrict__ twl, const T* __restrict__ tw, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) twl[i] = {tw[2 * i], tw[2 * i + 1]};
}
template <typename T>
__device__ inline cx<T> tw_lds(const cx<T>* __restrict__ twl, int k, int inverse) {
    cx<T> w = twl[k];
    if (inverse) w.im = -w.im;
    return w;
}
// x / d for x < 2^16 with magic = floor(2^32 / d) + 1 (make_fft_plan)
__device__ inline int fastdiv(int x, unsigned magic) { return (int)__umulhi((unsigned)x, magic); }

// One Stockham stage over `nseq` sequences stored [seq][n] in LDS (src -> dst), all lanes of the workgroup cooperate.
// Butterfly j of a sequence reads src[j + r m] (m = n / R), multiplies by w_{ns R}^{k r} = w_n^{k r tstep} with k = j mod ns,
// and writes dst[(j / ns) ns R + k + q ns].  k r tstep < n for every r < R: no reduction of the twiddle index is needed.
template <typename T, int R>
__device__ inline void fft_stage_r(const cx<T>* __restrict__ src, cx<T>* __restrict__ dst, int n, int np, int ns, unsigned magic_ns,
                                   unsigned magic_m, int nseq, const cx<T>* __restrict__ twl, int inverse) {
    const int m = n / R;
    const int tstep = m / ns;                                   // n / (ns R)
    for (int w = threadIdx.x; w < nseq * m; w += blockDim.x) {
        const int seq = fastdiv(w, magic_m), j = w - seq * m;
        const int jd = ns == 1 ? j : fastdiv(j, magic_ns), k = j - jd * ns;
        const cx<T>* s = src + seq * np;
        cx<T> v[R];
        int ti = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v[r] = s[j + r * m];
            if (r > 0) v[r] = cmul(v[r], tw_lds(twl, ti, inverse));
            ti += k * tstep;
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
                for (int r = 1; r < R; ++r) acc = cadd(acc, cmul(v[r], tw_lds(twl, ((q * r) % R) * m, inverse)));
                y[q] = acc;
            }
        }
        cx<T>* d = dst + seq * np + jd * ns * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) d[q * ns] = y[q];
    }
}

// Odd prime radix R (7, 11, 13) with every input in registers.  W_R^{q (R-r)} = conj(W_R^{q r}), so with a_r = v_r + v_{R-r}
// and b_r = v_r - v_{R-r}:  y_q = v_0 + sum_{r<=H} (a_r C_{qr} + i b_r S_{qr}),  y_{R-q} = v_0 + sum (a_r C_{qr} - i b_r S_{qr})
// -- 4 H real multiply-adds per output pair instead of 4 (R - 1) per output.  (528 = 4.4.3.11: the radix-11 stage was 85 %
// of the Pyramid's transform time as a generic O(R^2) sweep with one lane per output.)
template <typename T, int R>
__device__ inline void fft_stage_prime(const cx<T>* __restrict__ src, cx<T>* __restrict__ dst, int n, int np, int ns, unsigned magic_ns,
                                       unsigned magic_m, int nseq, const cx<T>* __restrict__ twl, int inverse) {
    constexpr int H = (R - 1) / 2;
    const int m = n / R;
    const int tstep = m / ns;
    T C[R], S[R];                                               // W_R^p = C[p] + i S[p]
#pragma unroll
    for (int p2 = 1; p2 < R; ++p2) {
        const cx<T> wv = tw_lds(twl, p2 * m, inverse);
        C[p2] = wv.re;
        S[p2] = wv.im;
    }
    for (int w = threadIdx.x; w < nseq * m; w += blockDim.x) {
        const int seq = fastdiv(w, magic_m), j = w - seq * m;
        const int jd = ns == 1 ? j : fastdiv(j, magic_ns), k = j - jd * ns;
        const cx<T>* s = src + seq * np;
        cx<T> v[R];
        int ti = 0;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v[r] = s[j + r * m];
            if (r > 0) v[r] = cmul(v[r], tw_lds(twl, ti, inverse));
            ti += k * tstep;
        }
        cx<T> a[H + 1], b[H + 1];
        cx<T> y0 = v[0];
#pragma unroll
        for (int r = 1; r <= H; ++r) {
            a[r] = cadd(v[r], v[R - r]);
            b[r] = csub(v[r], v[R - r]);
            y0 = cadd(y0, a[r]);
        }
        cx<T>* d = dst + seq * np + jd * ns * R + k;
        d[0] = y0;
#pragma unroll
        for (int q = 1; q <= H; ++q) {
            T cr = v[0].re, ci = v[0].im, sr = 0, si = 0;
#pragma unroll
            for (int r = 1; r <= H; ++r) {
                constexpr int dummy = 0; (void)dummy;
                const int p2 = (q * r) % R;
                cr += a[r].re * C[p2];
                ci += a[r].im * C[p2];
                sr += b[r].re * S[p2];
                si += b[r].im * S[p2];
            }
            // i b S = (-b.im S, b.re S)
            d[q * ns] = {cr - si, ci + sr};
            d[(R - q) * ns] = {cr + si, ci - sr};
        }
    }
}

// Any other (prime) radix, two sweeps: (1) every input is multiplied in place by its stage twiddle; (2) one lane per
// OUTPUT evaluates the length-R DFT  sum_r v[r] W_R^{q r}  with W_R^p = w_n^{p m}  (R broadcast reads from the LDS table).
template <typename T>
__device__ inline void fft_stage_any(cx<T>* __restrict__ src, cx<T>* __restrict__ dst, int n, int np, int ns, unsigned magic_ns,
                                     unsigned magic_m, int nseq, int R, const cx<T>* __restrict__ twl, int inverse) {
    const int m = n / R;
    const int tstep = m / ns;
    for (int w = threadIdx.x; w < nseq * n; w += blockDim.x) {   // element e = j + r m of sequence seq
        const int sr = fastdiv(w, magic_m), seq = sr / R, r = sr - seq * R, j = w - sr * m;   // (w / m) = seq R + r
        if (r > 0) {
            const int jd = ns == 1 ? j : fastdiv(j, magic_ns), k = j - jd * ns;
            cx<T>* x = src + seq * np + r * m + j;
            *x = cmul(*x, tw_lds(twl, k * r * tstep, inverse));
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < nseq * n; w += blockDim.x) {   // output q of butterfly j of sequence seq: w = (seq R + q) m + j
        const int sq = fastdiv(w, magic_m), j = w - sq * m;
        const int seq = sq / R, q = sq - seq * R;
        const int jd = ns == 1 ? j : fastdiv(j, magic_ns), k = j - jd * ns;
        const cx<T>* s = src + seq * np + j;
        cx<T> acc = s[0];
        int qr = 0;
        for (int r = 1; r < R; ++r) {
            qr += q;
            qr = qr >= R ? qr - R : qr;
            acc = cadd(acc, cmul(s[r * m], tw_lds(twl, qr * m, inverse)));
        }
        dst[seq * np + jd * ns * R + k + q * ns] = acc;
    }
}

// Full 1-D transform of `nseq` sequences; returns the buffer that holds the result (a or b).  twl: the n-entry twiddle
// table in LDS (fft_load_twiddles; the first stage's barrier orders it).
template <typename T>
__device__ inline cx<T>* fft_lds(cx<T>* a, cx<T>* b, const FftPlan& pl, int nseq, const cx<T>* __restrict__ twl, int inverse) {
    int ns = 1;
    cx<T>* src = a;
    cx<T>* dst = b;
    for (int s = 0; s < pl.n_fac; ++s) {
        const int R = pl.fac[s];
        __syncthreads();
        switch (R) {
            case 2: fft_stage_r<T, 2>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 3: fft_stage_r<T, 3>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 4: fft_stage_r<T, 4>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 5: fft_stage_r<T, 5>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 7: fft_stage_prime<T, 7>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 11: fft_stage_prime<T, 11>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 13: fft_stage_prime<T, 13>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            default: fft_stage_any<T>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, R, twl, inverse); break;
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
