// 528-point transform of the 40x40 Pyramid (nRes = 528 = 24 x 22, BASELINE configs[2]) with register-resident factors; further down the
// factors of 288 = 16 x 18 (the 20x20 Pyramid), used by the same kernels (pyr528_kernels.hip).
//
// The Stockham transform of fft.hpp moves every point through LDS once per radix (16, 3, 11) and spends three quarters of its
// instructions on addresses, twiddle fetches and loop control.  Here a lane holds a whole factor:
//     x[22 n1 + n2]  --24-point DFT over n1 in the registers of lane n2-->  * w_528^(n2 k1)  --one exchange through LDS-->
//     22-point DFT over n2 in the registers of lane k1  -->  X[k1 + 24 k2]
// and the inverse runs the same two factors in the other order (22-point over k2 in lane k1 first), so that the column pass --
// forward transform, mask, inverse transform -- needs no exchange between the two transforms: the lane that ends the forward one
// with the frequencies k1 + 24 k2 starts the inverse one with them.  24 = 3 x 8 and 22 = 2 x 11 are products of coprime factors:
// inside a lane they are prime-factor (Good-Thomas) transforms, index renaming only, no twiddles.  Complex numbers are
// 2-vectors so that additions and real-coefficient multiply-adds (all of the 11-point butterfly) compile to packed
// v_pk_add_f32 / v_pk_fma_f32.
#pragma once
#include <hip/hip_runtime.h>

namespace ao {
namespace f528 {

typedef float v2 __attribute__((ext_vector_type(2)));             // (re, im)

constexpr int kN = 528, kN1 = 24, kN2 = 22;

__host__ __device__ inline v2 mul_i(v2 a) { return v2{-a.y, a.x}; }                        // a * i
template <bool INV> __host__ __device__ inline v2 rot90(v2 a) {                           // a * (-i) forward, a * (+i) inverse
    return INV ? v2{-a.y, a.x} : v2{a.y, -a.x};
}
__host__ __device__ inline v2 cmul2(v2 a, v2 w) { return a.x * w + a.y * v2{-w.y, w.x}; }
template <bool INV> __host__ __device__ inline v2 cmul_tw(v2 a, v2 w) {                   // a * w forward, a * conj(w) inverse
    return INV ? a.x * v2{w.x, -w.y} + a.y * v2{w.y, w.x} : a.x * w + a.y * v2{-w.y, w.x};
}

// ---- 3 ----  W_3 = -1/2 -+ i sqrt(3)/2
template <bool INV> __host__ __device__ inline void dft3(v2& x0, v2& x1, v2& x2) {
    constexpr float s3 = 0.86602540378443864676f;
    const v2 t = x1 + x2, u = x1 - x2;
    const v2 m = x0 - 0.5f * t;
    const v2 n = rot90<INV>(s3 * u);
    x0 = x0 + t;
    x1 = m + n;
    x2 = m - n;
}
// ---- 4 ----
template <bool INV> __host__ __device__ inline void dft4v(v2& x0, v2& x1, v2& x2, v2& x3) {
    const v2 a = x0 + x2, b = x0 - x2, c = x1 + x3, d = rot90<INV>(x1 - x3);
    x0 = a + c;
    x1 = b + d;
    x2 = a - c;
    x3 = b - d;
}
// ---- 8 ----  in place, natural order
template <bool INV> __host__ __device__ inline void dft8(v2& x0, v2& x1, v2& x2, v2& x3, v2& x4, v2& x5, v2& x6, v2& x7) {
    constexpr float c2 = 0.70710678118654752440f;
    dft4v<INV>(x0, x2, x4, x6);                                  // even samples -> e[k] in x0 x2 x4 x6
    dft4v<INV>(x1, x3, x5, x7);                                  // odd samples  -> o[k] in x1 x3 x5 x7
    // o[k] *= W_8^k: 1, (1 -+ i) / sqrt 2, -+ i, (-1 -+ i) / sqrt 2
    const v2 o1 = INV ? v2{x3.x - x3.y, x3.x + x3.y} * c2 : v2{x3.x + x3.y, x3.y - x3.x} * c2;
    const v2 o2 = rot90<INV>(x5);
    const v2 o3 = INV ? v2{-x7.x - x7.y, x7.x - x7.y} * c2 : v2{x7.y - x7.x, -x7.x - x7.y} * c2;
    const v2 e0 = x0, e1 = x2, e2 = x4, e3 = x6, o0 = x1;
    x0 = e0 + o0;
    x4 = e0 - o0;
    x1 = e1 + o1;
    x5 = e1 - o1;
    x2 = e2 + o2;
    x6 = e2 - o2;
    x3 = e3 + o3;
    x7 = e3 - o3;
}
// ---- 11 ----  y_q = v_0 + sum_r (a_r C_qr +- i b_r S_qr), a_r = v_r + v_(11-r), b_r = v_r - v_(11-r)   (fft.hpp, fft_stage_prime)
// v[i] is addressed as v[P[i]]: the caller's prime-factor renaming.  In place.
template <bool INV, int P0, int P1, int P2, int P3, int P4, int P5, int P6, int P7, int P8, int P9, int P10, int Q0, int Q1, int Q2, int Q3,
          int Q4, int Q5, int Q6, int Q7, int Q8, int Q9, int Q10>
__host__ __device__ inline void dft11(v2 (&v)[22]) {
    constexpr int P[11] = {P0, P1, P2, P3, P4, P5, P6, P7, P8, P9, P10};
    constexpr int Q[11] = {Q0, Q1, Q2, Q3, Q4, Q5, Q6, Q7, Q8, Q9, Q10};
    // cos / sin of 2 pi p / 11, p = 0..10
    constexpr float C[11] = {1.f, 0.84125353283118116886f, 0.41541501300188642553f, -0.14231483827328514044f, -0.65486073394528506406f,
                             -0.95949297361449738989f, -0.95949297361449738989f, -0.65486073394528506406f, -0.14231483827328514044f,
                             0.41541501300188642553f, 0.84125353283118116886f};
    constexpr float S[11] = {0.f, 0.54064081745559758211f, 0.90963199535451837141f, 0.98982144188093273238f, 0.75574957435425828377f,
                             0.28173255684142969771f, -0.28173255684142969771f, -0.75574957435425828377f, -0.98982144188093273238f,
                             -0.90963199535451837141f, -0.54064081745559758211f};
    const v2 v0 = v[P[0]];
    v2 a[6], b[6];
    v2 y0 = v0;
#pragma unroll
    for (int r = 1; r <= 5; ++r) {
        a[r] = v[P[r]] + v[P[11 - r]];
        b[r] = v[P[r]] - v[P[11 - r]];
        y0 = y0 + a[r];
    }
    v[Q[0]] = y0;
#pragma unroll
    for (int q = 1; q <= 5; ++q) {
        v2 c = v0, s = v2{0.f, 0.f};
#pragma unroll
        for (int r = 1; r <= 5; ++r) {
            const int p = (q * r) % 11;
            c = c + a[r] * C[p];
            s = s + b[r] * S[p];
        }
        // forward: W^p = C - i S  ->  y_q = c - i s, y_(11-q) = c + i s ; inverse: the conjugate
        const v2 is = mul_i(s);
        v[Q[q]] = INV ? c + is : c - is;
        v[Q[11 - q]] = INV ? c - is : c + is;
    }
}

// ---- 22 = 2 x 11, prime-factor: n = (11 n1 + 2 n2) mod 22, k = (11 k1 + 12 k2) mod 22.  In place, natural order. ----
template <bool INV> __host__ __device__ inline void dft22(v2 (&v)[22]) {
#pragma unroll
    for (int n2 = 0; n2 < 11; ++n2) {                             // length 2 over n1: positions (2 n2) and (11 + 2 n2) mod 22
        const int p0 = (2 * n2) % 22, p1 = (11 + 2 * n2) % 22;
        const v2 s = v[p0] + v[p1], d = v[p0] - v[p1];
        v[p0] = s;                                                // k1 = 0
        v[p1] = d;                                                // k1 = 1
    }
    // length 11 over n2 at positions (11 k1 + 2 n2) mod 22 -> outputs k2 at (11 k1 + 12 k2) mod 22.  Both position sets are the
    // same 11 slots (the even ones for k1 = 0, the odd ones for k1 = 1), so the butterfly can work in place.
    dft11<INV, 0, 2, 4, 6, 8, 10, 12, 14, 16, 18, 20, 0, 12, 2, 14, 4, 16, 6, 18, 8, 20, 10>(v);
    dft11<INV, 11, 13, 15, 17, 19, 21, 1, 3, 5, 7, 9, 11, 1, 13, 3, 15, 5, 17, 7, 19, 9, 21>(v);
}

// ---- 24 = 3 x 8, prime-factor: n = (8 n1 + 3 n2) mod 24, k = (16 k1 + 9 k2) mod 24.  In place, natural order. ----
template <bool INV> __host__ __device__ inline void dft24(v2 (&v)[24]) {
#pragma unroll
    for (int n2 = 0; n2 < 8; ++n2)                                // length 3 over n1 at positions (8 n1 + 3 n2) mod 24; k1 replaces n1
        dft3<INV>(v[(3 * n2) % 24], v[(8 + 3 * n2) % 24], v[(16 + 3 * n2) % 24]);
    // length 8 over n2 at positions (8 k1 + 3 n2) mod 24 -> output k2 belongs at (16 k1 + 9 k2) mod 24
    v2 y[24];
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        v2 t[8];
#pragma unroll
        for (int n2 = 0; n2 < 8; ++n2) t[n2] = v[(8 * k1 + 3 * n2) % 24];
        dft8<INV>(t[0], t[1], t[2], t[3], t[4], t[5], t[6], t[7]);
#pragma unroll
        for (int k2 = 0; k2 < 8; ++k2) y[(16 * k1 + 9 * k2) % 24] = t[k2];
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) v[i] = y[i];
}


// ---- 288 = 16 x 18 (the 20 x 20 Pyramid of the reference's Papyrus set-up) ------------------------------------------------------------
// ---- 16 = 4 x 4 (Cooley-Tukey, inner twiddles W_16^(q0 r0)), in place, natural order ----
template <bool INV> __host__ __device__ inline void dft16(v2 (&v)[16]) {
    constexpr float c1 = 0.92387953251128673848f, s1 = 0.38268343236508978178f, c2 = 0.70710678118654752440f;
#pragma unroll
    for (int r0 = 0; r0 < 4; ++r0) dft4v<INV>(v[r0], v[4 + r0], v[8 + r0], v[12 + r0]);     // over r1: v[4 q0 + r0] = t[r0][q0]
    // t[r0][q0] *= W_16^(q0 r0) = cos - i sin (forward), cos + i sin (inverse): exponents 1 2 3 / 2 4 6 / 3 6 9
    auto rot = [](v2 x, float c, float sn) { return INV ? v2{x.x * c - x.y * sn, x.x * sn + x.y * c} : v2{x.x * c + x.y * sn, x.y * c - x.x * sn}; };
    v[5] = rot(v[5], c1, s1);
    v[6] = rot(v[6], c2, c2);
    v[7] = rot(v[7], s1, c1);
    v[9] = rot(v[9], c2, c2);
    v[10] = rot90<INV>(v[10]);
    v[11] = rot(v[11], -c2, c2);
    v[13] = rot(v[13], s1, c1);
    v[14] = rot(v[14], -c2, c2);
    v[15] = rot(v[15], -c1, -s1);
    v2 y[16];
#pragma unroll
    for (int q0 = 0; q0 < 4; ++q0) {
        v2 a0 = v[4 * q0], a1 = v[4 * q0 + 1], a2 = v[4 * q0 + 2], a3 = v[4 * q0 + 3];         // over r0
        dft4v<INV>(a0, a1, a2, a3);
        y[q0] = a0;
        y[q0 + 4] = a1;
        y[q0 + 8] = a2;
        y[q0 + 12] = a3;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = y[i];
}
// ---- 9 = 3 x 3 (Cooley-Tukey): n = 3 n1 + n2, k = k1 + 3 k2; on the elements v[P[i]], outputs to v[Q[k]] ----
template <bool INV, int P0, int P1, int P2, int P3, int P4, int P5, int P6, int P7, int P8, int Q0, int Q1, int Q2, int Q3, int Q4, int Q5, int Q6, int Q7,
          int Q8>
__host__ __device__ inline void dft9(v2 (&v)[18]) {
    constexpr int P[9] = {P0, P1, P2, P3, P4, P5, P6, P7, P8};
    constexpr int Q[9] = {Q0, Q1, Q2, Q3, Q4, Q5, Q6, Q7, Q8};
    // cos / sin of 2 pi p / 9, p = 1, 2, 4
    constexpr float c1 = 0.76604444311897803520f, s1 = 0.64278760968653932632f, c2 = 0.17364817766693034885f, s2 = 0.98480775301220805937f,
                    c4 = -0.93969262078590838405f, s4 = 0.34202014332566873304f;
    v2 t[3][3];                                                    // t[n2][k1]
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
        v2 a = v[P[n2]], b = v[P[3 + n2]], c = v[P[6 + n2]];
        dft3<INV>(a, b, c);
        t[n2][0] = a;
        t[n2][1] = b;
        t[n2][2] = c;
    }
    auto rot = [](v2 x, float c, float sn) { return INV ? v2{x.x * c - x.y * sn, x.x * sn + x.y * c} : v2{x.x * c + x.y * sn, x.y * c - x.x * sn}; };
    t[1][1] = rot(t[1][1], c1, s1);                                // W_9^(n2 k1)
    t[1][2] = rot(t[1][2], c2, s2);
    t[2][1] = rot(t[2][1], c2, s2);
    t[2][2] = rot(t[2][2], c4, s4);
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        v2 a = t[0][k1], b = t[1][k1], c = t[2][k1];
        dft3<INV>(a, b, c);                                        // over n2 -> k2
        v[Q[k1]] = a;
        v[Q[k1 + 3]] = b;
        v[Q[k1 + 6]] = c;
    }
}
// ---- 18 = 2 x 9, prime-factor: n = (9 n1 + 2 n2) mod 18, k = (9 k1 + 10 k2) mod 18.  In place, natural order. ----
template <bool INV> __host__ __device__ inline void dft18(v2 (&v)[18]) {
#pragma unroll
    for (int n2 = 0; n2 < 9; ++n2) {
        const int p0 = (2 * n2) % 18, p1 = (9 + 2 * n2) % 18;
        const v2 s = v[p0] + v[p1], d = v[p0] - v[p1];
        v[p0] = s;                                                // k1 = 0
        v[p1] = d;                                                // k1 = 1
    }
    // length 9 over n2 at (9 k1 + 2 n2) mod 18 -> k2 at (9 k1 + 10 k2) mod 18: the even slots for k1 = 0, the odd ones for k1 = 1
    dft9<INV, 0, 2, 4, 6, 8, 10, 12, 14, 16, 0, 10, 2, 12, 4, 14, 6, 16, 8>(v);
    dft9<INV, 9, 11, 13, 15, 17, 1, 3, 5, 7, 9, 1, 11, 3, 13, 5, 15, 7, 17>(v);
}

// the factor of length L in the registers of a lane
template <int L, bool INV> __host__ __device__ inline void dft_len(v2 (&v)[L]) {
    static_assert(L == 24 || L == 22 || L == 16 || L == 18, "no register-resident transform of this length");
    if constexpr (L == 24) dft24<INV>(v);
    else if constexpr (L == 22) dft22<INV>(v);
    else if constexpr (L == 16) dft16<INV>(v);
    else dft18<INV>(v);
}

}  // namespace f528
}  // namespace ao
