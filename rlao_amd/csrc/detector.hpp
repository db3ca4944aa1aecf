// WFS camera noise model on the device:  OOPAO/Detector.py:178-301 (integrate + readout, one frame per read-out).
//   photon noise  Poisson(frame)                      :204-206, 285-286
//   quantum efficiency  frame * QE                    :178-180
//   dark shot noise  + Poisson(darkCurrent * T)       :224-229
//   saturation  clip(frame, 0, FWC)                   :183-187
//   EMCCD gain before, CCD / CMOS gain after the read-out noise   :243-244, 258-259
//   read-out noise  + round(N(0,1) * sigma)           :218-221
//   ADC  trunc(frame / FWC * (2^bits - 1)), clipped to 2^bits - 1   :190-201
// The reference seeds its generators from the wall clock (Detector.py:127-130): a noisy frame is reproducible only in
// distribution.  Here every pixel draws from a counter-based Philox4x32-7 stream keyed by (seed) and indexed by
// (pixel, global env index, frame counter): reproducible, independent of the batch layout and of the kernel variant.
#pragma once
#include "common.hpp"

namespace ao {

struct Philox {
    uint32_t c[4], k[2];
    __device__ inline void round() {
        // one 32 x 32 -> 64 multiply per product (v_mad_u64_u32) instead of a mul-hi and a mul-lo: integer multiplies are quarter rate
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        const uint32_t n0 = hi1 ^ c[1] ^ k[0], n2 = hi0 ^ c[3] ^ k[1];
        c[0] = n0; c[1] = lo1; c[2] = n2; c[3] = lo0;
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
};

// 4 x 32 random bits for (counter, key)
__device__ inline void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t (&out)[4]) {
    Philox p{{c0, c1, c2, c3}, {k0, k1}};
#pragma unroll
    for (int i = 0; i < 7; ++i) p.round();                       // Philox4x32-7 (Salmon et al. 2011: 7 rounds pass BigCrush):
                                                                 // 32-bit integer multiplies are quarter-rate on CDNA
    out[0] = p.c[0]; out[1] = p.c[1]; out[2] = p.c[2]; out[3] = p.c[3];
}

__device__ inline float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }   // (0, 1)

// One pixel's stream: uniforms on demand, 4 per Philox call (a shift register: no dynamically indexed array, which
// the compiler would place in scratch memory).
struct PixelRng {
    uint32_t c0, c1, c2, k0, k1, sub;
    uint32_t b0, b1, b2, b3;
    int have;
    __device__ inline PixelRng(uint32_t pixel, uint32_t env, uint32_t frame, uint32_t s0, uint32_t s1)
        : c0(pixel), c1(env), c2(frame), k0(s0), k1(s1), sub(0), b0(0), b1(0), b2(0), b3(0), have(0) {}
    __device__ inline float next() {
        if (have == 0) {
            uint32_t o[4];
            philox4x32(c0, c1, c2, sub++, k0, k1, o);
            b0 = o[3]; b1 = o[2]; b2 = o[1]; b3 = o[0];        // handed out in the order o[3], o[2], o[1], o[0]
            have = 4;
        }
        const uint32_t v = b0;
        b0 = b1; b1 = b2; b2 = b3;
        --have;
        return u01(v);
    }
};

// log(k!) for integer-valued k >= 0.
// (lgammaf costs ~10x more instructions, and the PTRS acceptance test below runs for every lit pixel of every frame.)
__device__ inline float log_factorial(float k) {
    // k >= 4: Stirling's series with three correction terms, |error| < 2e-8;  k = 0..3 from a 4-entry select
    const float kk = fmaxf(k, 4.f);
    const float r = __builtin_amdgcn_rcpf(kk), r2 = r * r;
    const float st = kk * __logf(kk) - kk + 0.5f * __logf(6.283185307179586f * kk) +
                     r * (0.0833333333f + r2 * (-0.00277777778f + r2 * 0.000793650794f));
    const float small = k < 2.f ? 0.f : (k < 3.f ? 0.693147181f : 1.791759469f);
    return k < 4.f ? small : st;
}

// Poisson(lam): sequential inversion below 12 (exact, ~lam iterations), Hoermann's PTRS above (exact; the algorithm
// NumPy's legacy generator uses for lam >= 10).
__device__ inline float poisson(float lam, PixelRng& g) {
    if (!(lam > 0.f)) return 0.f;
    if (lam < 12.f) {
        const float u = g.next();
        float p = __expf(-lam), cdf = p;
        int k = 0;
        while (u > cdf && k < 200) {
            ++k;
            p *= lam * __builtin_amdgcn_rcpf((float)k);              // 1 ulp reciprocal: the term of the series to 1e-7
            cdf += p;
        }
        return (float)k;
    }
    // (1-ulp hardware reciprocals / square root: an IEEE division expands to ~10 instructions, and the constants of the
    //  hat function do not need the last bit)
    const float slam = __builtin_amdgcn_sqrtf(lam), loglam = __logf(lam);
    const float b = 0.931f + 2.53f * slam, a = -0.059f + 0.02483f * b;
    const float invalpha = 1.1239f + 1.1328f * __builtin_amdgcn_rcpf(b - 3.4f), vr = 0.9277f - 3.6224f * __builtin_amdgcn_rcpf(b - 2.f);
    const float log_invalpha = __logf(invalpha);
    for (int it = 0; it < 64; ++it) {
        const float U = g.next() - 0.5f, V = g.next();
        const float us = 0.5f - fabsf(U);
        const float rus = __builtin_amdgcn_rcpf(us);
        const float kf = floorf((2.f * a * rus + b) * U + lam + 0.43f);
        if (us >= 0.07f && V <= vr) return kf;
        if (kf < 0.f || (us < 0.013f && V > us)) continue;
        if (__logf(V) + log_invalpha - __logf(a * rus * rus + b) <= -lam + kf * loglam - log_factorial(kf)) return kf;
    }
    return floorf(lam + 0.5f);
}

__device__ inline float gaussian(PixelRng& g) {
    const float u1 = g.next(), u2 = g.next();
    return __builtin_amdgcn_sqrtf(-2.f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

// photons in -> camera counts out
__device__ inline float detector_pixel(float photons, const DetectorCfg& d, uint32_t pixel, uint32_t env) {
    PixelRng g(pixel, env + d.env_offset, d.frame_counter, d.seed_lo, d.seed_hi);
    float f = photons;
    if (d.photon_noise) f = poisson(f, g);
    f *= d.qe;
    if (d.dark_e > 0.f) f += poisson(d.dark_e, g);
    if (d.fwc > 0.f) f = fminf(fmaxf(f, 0.f), d.fwc);
    if (d.emccd) f *= d.gain;
    if (d.readout_noise != 0.f) f += rintf(gaussian(g) * d.readout_noise);
    if (!d.emccd) f *= d.gain;
    if (d.bits > 0) {
        const float top = (float)((1u << d.bits) - 1u);
        f = truncf(f / d.fwc * top);                              // astype(int): toward zero
        f = fminf(f, top);                                        // clip(frame, frame.min(), 2^bits - 1)
    }
    return f;
}

// host-side launcher (detector_kernels.hip): applies the camera to frame [E][cam*cam] in place; for a Shack-Hartmann
// frame (valid2d != null) it also writes the maximum over the valid lenslets' pixels to wfs_max[E].
template <typename T>
int launch_detector(T* frame, T* wfs_max, const uint8_t* valid2d, int n_env, int cam, int n_subap, const DetectorCfg& d,
                    hipStream_t st);

}  // namespace ao
