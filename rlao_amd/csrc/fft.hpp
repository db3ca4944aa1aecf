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
// Position of element i of a sequence in LDS: one slot of padding after every 16.  A first stage of radix R writes its
// outputs R apart (radix 16: 32 dwords = every lane in the same bank); with the padding lane j lands 17 slots after lane
// j - 1.  Every buffer the transform touches -- the caller's input and the result included -- is indexed through fpad().
__host__ __device__ inline int fpad(int i) { return i + (i >> 4); }
// x / d for x < 2^16 with magic = floor(2^32 / d) + 1 (make_fft_plan, fft_magic); d = 1 has no such magic in 32 bits: magic 0 = "x itself"
// (a one-stage plan of a prime length -- the 29-pixel screens of a small telescope with a field of view -- divides by m = 1)
__device__ inline int fastdiv(int x, unsigned magic) { return magic ? (int)__umulhi((unsigned)x, magic) : x; }
__host__ __device__ inline unsigned fft_magic(unsigned d) { return d <= 1u ? 0u : (unsigned)((1ull << 32) / d) + 1u; }

// Length-4 DFT in place (forward: W_4 = -i; inverse: +i).
template <typename T>
__device__ inline void dft4(cx<T>& x0, cx<T>& x1, cx<T>& x2, cx<T>& x3, int inverse) {
    const cx<T> a = cadd(x0, x2), b = csub(x0, x2), c = cadd(x1, x3), d = csub(x1, x3);
    const cx<T> dj = inverse ? cx<T>{-d.im, d.re} : cx<T>{d.im, -d.re};
    x0 = cadd(a, c);
    x1 = cadd(b, dj);
    x2 = csub(a, c);
    x3 = csub(b, dj);
}
// Length-16 DFT in registers as 4 x 4: r = 4 r1 + r0, q = q0 + 4 q1,  W_16^{q r} = W_4^{q0 r1} W_16^{q0 r0} W_4^{q1 r0}.
template <typename T>
__device__ inline void dft16(cx<T> (&v)[16], cx<T> (&y)[16], int inverse) {
    // cos / sin of 2 pi k / 16, k = 1, 2, 3
    constexpr double c1 = 0.92387953251128673848, s1 = 0.38268343236508978178, c2 = 0.70710678118654752440;
#pragma unroll
    for (int r0 = 0; r0 < 4; ++r0) dft4<T>(v[r0], v[4 + r0], v[8 + r0], v[12 + r0], inverse);   // over r1: v[4 q0 + r0] = t[r0][q0]
    // t[r0][q0] *= W_16^{q0 r0}: exponents 1 2 3 / 2 4 6 / 3 6 9
    const T sg = inverse ? (T)1 : (T)-1;                         // forward: exp(-i x)
    auto rot = [&](cx<T>& x, double c, double sn) { x = cmul(x, cx<T>{(T)c, sg * (T)sn}); };
    rot(v[4 * 1 + 1], c1, s1);            // k = 1
    rot(v[4 * 1 + 2], c2, c2);            // k = 2
    rot(v[4 * 1 + 3], s1, c1);            // k = 3
    rot(v[4 * 2 + 1], c2, c2);            // k = 2
    rot(v[4 * 2 + 2], 0.0, 1.0);          // k = 4
    rot(v[4 * 2 + 3], -c2, c2);           // k = 6
    rot(v[4 * 3 + 1], s1, c1);            // k = 3
    rot(v[4 * 3 + 2], -c2, c2);           // k = 6
    rot(v[4 * 3 + 3], -c1, -s1);          // k = 9
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) {
        cx<T> a0 = v[4 * q0], a1 = v[4 * q0 + 1], a2 = v[4 * q0 + 2], a3 = v[4 * q0 + 3];       // over r0
        dft4<T>(a0, a1, a2, a3, inverse);
        y[q0] = a0;
        y[q0 + 4] = a1;
        y[q0 + 8] = a2;
        y[q0 + 12] = a3;
    }
}

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
            v[r] = s[fpad(j + r * m)];
            if (r > 0 && ns > 1) v[r] = cmul(v[r], tw_lds(twl, ti, inverse));      // first stage: k = 0, every twiddle is 1
            ti += k * tstep;
        }
        cx<T> y[R];
        if constexpr (R == 16) {
            dft16<T>(v, y, inverse);
        } else if (R == 2) {
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
        cx<T>* d = dst + seq * np;
        const int o = jd * ns * R + k;
#pragma unroll
        for (int q = 0; q < R; ++q) d[fpad(o + q * ns)] = y[q];
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
            v[r] = s[fpad(j + r * m)];
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
        cx<T>* d = dst + seq * np;
        const int o = jd * ns * R + k;
        d[fpad(o)] = y0;
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
            d[fpad(o + q * ns)] = {cr - si, ci + sr};
            d[fpad(o + (R - q) * ns)] = {cr + si, ci - sr};
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
            cx<T>* x = src + seq * np + fpad(r * m + j);
            *x = cmul(*x, tw_lds(twl, k * r * tstep, inverse));
        }
    }
    __syncthreads();
    for (int w = threadIdx.x; w < nseq * n; w += blockDim.x) {   // output q of butterfly j of sequence seq: w = (seq R + q) m + j
        const int sq = fastdiv(w, magic_m), j = w - sq * m;
        const int seq = sq / R, q = sq - seq * R;
        const int jd = ns == 1 ? j : fastdiv(j, magic_ns), k = j - jd * ns;
        const cx<T>* s = src + seq * np;
        cx<T> acc = s[fpad(j)];
        int qr = 0;
        for (int r = 1; r < R; ++r) {
            qr += q;
            qr = qr >= R ? qr - R : qr;
            acc = cadd(acc, cmul(s[fpad(j + r * m)], tw_lds(twl, qr * m, inverse)));
        }
        dst[seq * np + fpad(jd * ns * R + k + q * ns)] = acc;
    }
}

// ---- compile-time plans ----------------------------------------------------------------------------------------------------
// The same stages with the transform length N, the stage's ns and the direction as template parameters: every division and
// modulo of the index arithmetic is by a constant, the twiddle strides fold, and the conjugation for the inverse is free.  The
// Pyramid's lengths (288 = 16.2.3.3 and 528 = 16.3.11: BASELINE configs[2] and the reference's Papyrus set-up) take this path;
// with run-time plans the index arithmetic was 2/3 of the instructions of the column pass.
template <typename T, bool INV>
__device__ inline cx<T> tw_ct(const cx<T>* __restrict__ twl, int k) {
    cx<T> w = twl[k];
    if (INV) w.im = -w.im;
    return w;
}

template <typename T, int R, int N, int NS, bool INV>
__device__ inline void fft_stage_ct(const cx<T>* __restrict__ src, cx<T>* __restrict__ dst, int nseq, const cx<T>* __restrict__ twl) {
    constexpr int m = N / R, tstep = m / NS, np = (N - 1 + ((N - 1) >> 4) + 1) | 1, H = (R - 1) / 2;
    constexpr bool prime = (R == 3 || R == 5 || R == 7 || R == 11 || R == 13);
    T C[prime ? R : 1], S[prime ? R : 1];
    if constexpr (prime) {
#pragma unroll
        for (int p2 = 1; p2 < R; ++p2) {
            const cx<T> wv = tw_ct<T, INV>(twl, p2 * m);
            C[p2] = wv.re;
            S[p2] = wv.im;
        }
    }
    for (int w = threadIdx.x; w < nseq * m; w += blockDim.x) {
        const int seq = w / m, j = w - seq * m;
        const int jd = j / NS, k = j - jd * NS;
        const cx<T>* s = src + seq * np;
        cx<T> v[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            v[r] = s[fpad(j + r * m)];
            if (r > 0 && NS > 1) v[r] = cmul(v[r], tw_ct<T, INV>(twl, k * r * tstep));
        }
        cx<T>* d = dst + seq * np;
        const int o = jd * NS * R + k;
        if constexpr (R == 16) {
            cx<T> y[R];
            dft16<T>(v, y, INV ? 1 : 0);
#pragma unroll
            for (int q = 0; q < R; ++q) d[fpad(o + q * NS)] = y[q];
        } else if constexpr (R == 2) {
            d[fpad(o)] = cadd(v[0], v[1]);
            d[fpad(o + NS)] = csub(v[0], v[1]);
        } else if constexpr (R == 4) {
            dft4<T>(v[0], v[1], v[2], v[3], INV ? 1 : 0);
#pragma unroll
            for (int q = 0; q < 4; ++q) d[fpad(o + q * NS)] = v[q];
        } else {
            static_assert(prime, "radix without a compile-time stage");
            cx<T> a[H + 1], b[H + 1];
            cx<T> y0 = v[0];
#pragma unroll
            for (int r = 1; r <= H; ++r) {
                a[r] = cadd(v[r], v[R - r]);
                b[r] = csub(v[r], v[R - r]);
                y0 = cadd(y0, a[r]);
            }
            d[fpad(o)] = y0;
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
                d[fpad(o + q * NS)] = {cr - si, ci + sr};
                d[fpad(o + (R - q) * NS)] = {cr + si, ci - sr};
            }
        }
    }
}

// 288 = 16 . 2 . 3 . 3   and   528 = 16 . 3 . 11 (the radix order of make_fft_plan); returns the buffer holding the result
template <typename T, int N, bool INV>
__device__ inline cx<T>* fft_lds_ct(cx<T>* a, cx<T>* b, int nseq, const cx<T>* __restrict__ twl) {
    static_assert(N == 288 || N == 528, "no compile-time plan for this length");
    __syncthreads();
    fft_stage_ct<T, 16, N, 1, INV>(a, b, nseq, twl);
    __syncthreads();
    if constexpr (N == 528) {
        fft_stage_ct<T, 3, N, 16, INV>(b, a, nseq, twl);
        __syncthreads();
        fft_stage_ct<T, 11, N, 48, INV>(a, b, nseq, twl);
        __syncthreads();
        return b;
    } else {
        fft_stage_ct<T, 2, N, 16, INV>(b, a, nseq, twl);
        __syncthreads();
        fft_stage_ct<T, 3, N, 32, INV>(a, b, nseq, twl);
        __syncthreads();
        fft_stage_ct<T, 3, N, 96, INV>(b, a, nseq, twl);
        __syncthreads();
        return a;
    }
}

// run-time plan, or the compile-time one when NFIX = plan length
template <typename T, int NFIX>
__device__ inline cx<T>* fft_any(cx<T>* a, cx<T>* b, const FftPlan& pl, int nseq, const cx<T>* __restrict__ twl, int inverse);

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
            case 3: fft_stage_prime<T, 3>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 4: fft_stage_r<T, 4>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 5: fft_stage_prime<T, 5>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 7: fft_stage_prime<T, 7>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 11: fft_stage_prime<T, 11>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
            case 16: fft_stage_r<T, 16>(src, dst, pl.n, pl.np, ns, pl.magic_ns[s], pl.magic_m[s], nseq, twl, inverse); break;
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

template <typename T, int NFIX>
__device__ inline cx<T>* fft_any(cx<T>* a, cx<T>* b, const FftPlan& pl, int nseq, const cx<T>* __restrict__ twl, int inverse) {
    if constexpr (NFIX == 0) {
        return fft_lds<T>(a, b, pl, nseq, twl, inverse);
    } else {
        return inverse ? fft_lds_ct<T, NFIX, true>(a, b, nseq, twl) : fft_lds_ct<T, NFIX, false>(a, b, nseq, twl);
    }
}

}  // namespace ao
