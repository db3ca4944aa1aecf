// WFS camera noise model on the device:  OOPAO/Detector.py:178-301 (integrate + readout, one frame per read-out).
//   photon noise  Poisson(frame)                      :204-206, 285-286
//   quantum efficiency  frame * QE                    :178-180
//   dark shot noise  + Poisson(darkCurrent * T)       :224-229
//   saturation  clip(frame, 0, FWC)                   :183-187
//   EMCCD gain before, CCD / CMOS gain after the read-out noise   :243-244, 258-259
//   read-out noise  + round(N(0,1) * sigma)           :218-221
//   ADC  trunc(frame / FWC * (2^bits - 1)), clipped to 2^bits - 1   :190-201
// The reference seeds its generators from the wall clock (Detector.py:127-130): a noisy frame is reproducible only in
// distribution.  Here the random numbers come from counter-based Philox4x32-7 streams keyed by `seed` and indexed by (pixel
// group, global env index, frame counter, purpose): reproducible, independent of the batch layout and of the kernel variant.
//
// Stream layout (the fused step kernel, k_detector_sh6 and k_detector draw the same numbers):
//   * the frame is cut into QUADS of 4 pixels; one Philox call per (quad, purpose) serves its 4 pixels, slot s gets word s:
//       purpose 0  photon noise: the word of the fine alias draw (poisson_alias.hpp)
//       purpose 3  photon noise: the uniform of the remainder's inversion
//       purpose 4  photon noise: the word of the coarse alias draw -- drawn only where a wave holds a pixel of 32 photons or more
//       purpose 1  dark shot noise: one uniform, inversion (purpose 5: the second uniform of PTRS for dark_e >= 10)
//       purpose 2  read-out noise: slots (0, 1) and (2, 3) share a Box-Muller pair (cos / sin branch)
//   * a pixel at or above the end of the alias table (1024 photons) is drawn by PTRS from the words of purposes 0 and 4; a round-0
//     rejection (~8 %) goes on with a stream of its own, (pixel, env, frame, 16 + j), two uniforms per round.
//   Quads: Shack-Hartmann frames with 6-pixel lenslets use the lane -> pixel map of the fused step kernel (a lane owns rows
//   0..5 of the lenslet columns q and q + 3): rows 0..3 of a column are one quad, rows 4..5 of the columns c and c + 3 another.
//   Any other frame: 4 consecutive pixels of a row.  A quad is named by the frame index of its slot-0 pixel.
// Cost (why it is laid out like this): a Philox call is ~100 issue slots (its 32 x 32 multiplies are quarter rate); one call per
// pixel and one data-dependent sampler per pixel (a wave runs both the faint and the bright branch, for as many rounds as
// its slowest lane) made the camera cost more than the physics (round 1: step kernel 40 -> 92 us with photon noise; round 2, quads
// + lock-step inversion + PTRS with LDS queues: 61 us; round 3: the fixed-cost alias sampler of poisson_alias.hpp).
#pragma once
#include "common.hpp"
#include "poisson_alias.hpp"

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
    for (int i = 0; i < 7; ++i) p.round();                       // Philox4x32-7 (Salmon et al. 2011: 7 rounds pass BigCrush)
    out[0] = p.c[0]; out[1] = p.c[1]; out[2] = p.c[2]; out[3] = p.c[3];
}

// strictly inside (0, 1): 23 bits + 1/2 ((x >> 8) + 0.5 rounds to 2^24 for the top word: a uniform of exactly 1)
__device__ inline float u01(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// (4-vectors indexed by a loop counter are ext_vector registers: the compiler indexes those with v_movrel, while a float[4] --
//  even behind a select chain -- is turned into an indexed array in scratch memory)
typedef float f32x4d __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4d __attribute__((ext_vector_type(4)));

constexpr float kPtrsFrom = 10.f;                              // PTRS is valid from lambda = 10 (NumPy switches there too)
enum { kDrawPhoton = 0, kDrawDark = 1, kDrawReadout = 2, kDrawPhoton2 = 3, kDrawPhoton3 = 4, kDrawDark2 = 5, kDrawPixelStream = 16 };

__device__ inline void quad_bits(uint32_t quad, uint32_t env, const DetectorCfg& d, uint32_t purpose, uint32_t (&o)[4]) {
    philox4x32(quad, env + d.env_offset, d.frame_counter, purpose, d.seed_lo, d.seed_hi, o);
}

// 1 / k, k = lane + 1, held across the wave (the inversion walks k in lock-step: the reciprocal of the wave's k is one
// v_readlane instead of a quarter-rate v_rcp per lane and step)
__device__ inline float recip_table_lane() { return 1.0f / (float)((threadIdx.x & 63) + 1); }

// Poisson(lam) for ONE lam shared by the wave (the dark current), lam < kPtrsFrom: inversion by sequential search with ONE uniform,
// X = #{k >= 0 : u > F(k)}.  The lanes walk k together, 4 steps per vote; a lane that has found its X keeps counting zeros; the
// search ends when the terms have underflowed (cdf no longer grows), at 64 steps at the latest.
__device__ inline float poisson_inversion(float lam, float u, float rtab) {
    float p = __expf(-lam), cdf = p, k = 0.f;
    for (int k0 = 0; k0 < 64; k0 += 4) {
        if (!__any(u > cdf) || !__any(p > 0.f)) break;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            k += (u > cdf && p > 0.f) ? 1.f : 0.f;                            // X > t = k0 + j
            p *= lam * __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rtab), k0 + j));   // P(t + 1) = P(t) lam / (t + 1)
            cdf += p;
        }
    }
    return k;
}

// log(k!) for integer-valued k >= 0.
// (lgammaf costs ~10x more instructions, and the PTRS acceptance test below runs for every undecided bright pixel of every frame.)
__device__ inline float log_factorial(float k) {
    // k >= 4: Stirling's series with three correction terms, |error| < 2e-8, as (k + 1/2) log k - k + log sqrt(2 pi) + ...: one
    // logarithm and one reciprocal (transcendentals issue at a quarter of the rate);  k = 0..3 from a 4-entry select
    const float kk = fmaxf(k, 4.f);
    const float r = __builtin_amdgcn_rcpf(kk), r2 = r * r;
    const float st = (kk + 0.5f) * __logf(kk) - kk + 0.918938533f + r * (0.0833333333f + r2 * (-0.00277777778f + r2 * 0.000793650794f));
    const float small = k < 2.f ? 0.f : (k < 3.f ? 0.693147181f : 1.791759469f);
    return k < 4.f ? small : st;
}

// Poisson(lam), lam >= kPtrsFrom: Hoermann's PTRS (exact; the algorithm NumPy's legacy generator uses from lam = 10).  Used for
// photon counts beyond the alias table and for a dark current of 10 electrons per frame or more.
// Round 0 takes its two uniforms from the pixel's words of two quad draws (photons: kDrawPhoton (U), kDrawPhoton3 (V)); a pixel that
// round 0 rejects (~8 % of them) goes on with a stream of its own: call j gives the uniforms of rounds 1 + 2j, 2 + 2j.
struct PtrsConst { float b, a, lam; };
__device__ inline PtrsConst ptrs_const(float lam) {
    // (1-ulp hardware square root: the constants of the hat function do not need the last bit)
    PtrsConst c;
    c.lam = lam;
    c.b = 0.931f + 2.53f * __builtin_amdgcn_sqrtf(lam);
    c.a = -0.059f + 0.02483f * c.b;
    return c;
}
// the cheap part of a round: proposal k and the squeeze (accepts 35 % of the proposals at 10 photons, 67 % at 100, without a
// logarithm).  V <= vr = 0.9277 - 3.6224 / (b - 2) is tested as (0.9277 - V)(b - 2) >= 3.6224: no division (b > 8.9).
__device__ inline bool ptrs_squeeze(const PtrsConst& c, uint32_t wu, uint32_t wv, float* kf, float* us_out, float* v_out) {
    const float U = u01(wu) - 0.5f, V = u01(wv);
    const float us = 0.5f - fabsf(U);
    *kf = floorf((2.f * c.a * __builtin_amdgcn_rcpf(us) + c.b) * U + c.lam + 0.43f);
    *us_out = us;
    *v_out = V;
    return us >= 0.07f && (0.9277f - V) * (c.b - 2.f) >= 3.6224f;
}
// the full acceptance test of a proposal the squeeze did not accept:
//   log V + log invalpha - log(a / us^2 + b) <= -lam + k log lam - log k!      with the left side as ONE logarithm
struct PtrsLogs { float loglam, invalpha; };
__device__ inline PtrsLogs ptrs_logs(const PtrsConst& c) {
    return {__logf(c.lam), 1.1239f + 1.1328f * __builtin_amdgcn_rcpf(c.b - 3.4f)};
}
__device__ inline bool ptrs_full(const PtrsConst& c, float kf, float us, float V, const PtrsLogs& g) {
    if (kf < 0.f || (us < 0.013f && V > us)) return false;
    const float rus = __builtin_amdgcn_rcpf(us);
    return __logf(V * g.invalpha * __builtin_amdgcn_rcpf(c.a * rus * rus + c.b)) <= -c.lam + kf * g.loglam - log_factorial(kf);
}
// rounds 1, 2, ... of a pixel whose round 0 was rejected: its own stream, until accepted.  Whole waves call this together.
__device__ inline float poisson_ptrs_rounds(const PtrsConst& c, const PtrsLogs& g, bool done, float result, uint32_t pixel, uint32_t env,
                                            const DetectorCfg& d) {
    float kf, us, V;
    for (uint32_t call = 0; call < 32; ++call) {
        if (!__any(!done)) break;
        uint32_t o[4];
        philox4x32(pixel, env + d.env_offset, d.frame_counter, kDrawPixelStream + call, d.seed_lo, d.seed_hi, o);
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (h == 1 && !__any(!done)) break;                  // (the first round of a call settles ~87 % of the lanes)
            bool acc = ptrs_squeeze(c, o[2 * h], o[2 * h + 1], &kf, &us, &V);
            if (__any(!done && !acc)) acc = acc || ptrs_full(c, kf, us, V, g);
            if (!done && acc) { result = kf; done = true; }
        }
    }
    return result;
}
__device__ inline float poisson_ptrs(float lam, uint32_t wu, uint32_t wv, uint32_t pixel, uint32_t env, const DetectorCfg& d) {
    const PtrsConst c = ptrs_const(lam);
    const PtrsLogs g = ptrs_logs(c);
    float kf, us, V;
    bool done = ptrs_squeeze(c, wu, wv, &kf, &us, &V);
    if (__any(!done)) done = done || ptrs_full(c, kf, us, V, g);      // (everyone evaluates it: one wave)
    return poisson_ptrs_rounds(c, g, done, done ? kf : floorf(lam + 0.5f), pixel, env, d);
}

// standard normals of a quad's read-out draw: slots (0, 1) and (2, 3) are the cos / sin branches of one Box-Muller pair each
__device__ inline void quad_normals(const uint32_t (&o)[4], f32x4d& n) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const float r = __builtin_amdgcn_sqrtf(-2.f * __logf(u01(o[2 * h])));
        const float t = u01(o[2 * h + 1]);                        // v_sin / v_cos take revolutions
        n[2 * h] = r * __builtin_amdgcn_cosf(t);
        n[2 * h + 1] = r * __builtin_amdgcn_sinf(t);
    }
}

// everything after the photon count: QE, dark shot noise, saturation, gain, read-out noise, ADC.  `dark` = the pixel's dark
// electrons (already drawn), `normal` = its standard normal.
__device__ inline float detector_finish(float f, const DetectorCfg& d, float dark, float normal) {
    f *= d.qe;
    f += dark;
    if (d.fwc > 0.f) f = fminf(fmaxf(f, 0.f), d.fwc);
    if (d.emccd) f *= d.gain;
    if (d.readout_noise != 0.f) f += rintf(normal * d.readout_noise);
    if (!d.emccd) f *= d.gain;
    if (d.bits > 0) {
        const float top = (float)((1u << d.bits) - 1u);
        f = truncf(f / d.fwc * top);                              // astype(int): toward zero
        f = fminf(f, top);                                        // clip(frame, frame.min(), 2^bits - 1)
    }
    return f;
}

// word `s` of a quad's draw / element `s` of a 4-vector with a run-time (wave-uniform) s: selects, not an indexed array
// (which would live in scratch memory); the slot loops below are NOT unrolled -- one copy of each sampler keeps the kernels that
// inline this small enough for the instruction cache (12 unrolled copies made the fused step kernel 114 KB of code)
__device__ inline uint32_t word_of(const uint32_t (&o)[4], int s) {
    const u32x4d ov = {o[0], o[1], o[2], o[3]};
    return ov[s];
}

// The photon counts of a quad: Poisson(v[s]) from the alias tables (poisson_alias.hpp; `tab` = the table where this kernel keeps it,
// LDS or global) -- straight-line code, the same cost for every lane -- then, for pixels at and above the table's end, PTRS in ONE
// rolled block that a wave enters only if it holds such a pixel.  EVERY lane of a wave must call (PTRS votes across the wave).
// o, o2, o3: the quad draws kDrawPhoton, kDrawPhoton2, kDrawPhoton3 (o3 is only looked at where v[s] >= 32).
__device__ inline void photon_quad(f32x4d& v, const uint32_t (&o)[4], const uint32_t (&o2)[4], const uint32_t (&o3)[4], const u32x4d& pix,
                                   uint32_t env, const DetectorCfg& d, float lmax, const uint32_t* __restrict__ tab) {
    f32x4d k;
    bool over = false;
    {
        float lam[4], out[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bool ov = v[s] >= lmax;
            over = over || ov;
            lam[s] = ov ? 0.f : v[s];
        }
        poisson_alias4(lam, o, o2, o3, tab, out);
#pragma unroll
        for (int s = 0; s < 4; ++s) k[s] = out[s];
    }
    if (__any(over)) {
        const u32x4d ov = {o[0], o[1], o[2], o[3]}, o3v = {o3[0], o3[1], o3[2], o3[3]};
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            const bool big = v[s] >= lmax;
            if (!__any(big)) continue;
            const float kb = poisson_ptrs(big ? v[s] : kPtrsFrom, ov[s], o3v[s], pix[s], env, d);
            k[s] = big ? kb : k[s];
        }
    }
    v = k;
}

// One quad of the camera, photons in -> counts out (used where the caller has no cheaper arrangement: k_detector, the unlit
// lenslets of the fused step kernel).  pix[s]: frame index of the quad's slot-s pixel (the per-pixel stream of a bright pixel).
// PHOTON = false: the quad is known to hold no light (unlit lenslets): only dark current / read-out / ADC.
template <bool PHOTON = true>
__device__ inline void detector_quad(f32x4d& v, const uint32_t (&pix)[4], uint32_t quad, uint32_t env, const DetectorCfg& d, float rtab,
                                     const PoissonAlias& pa) {
    const u32x4d pv = {pix[0], pix[1], pix[2], pix[3]};
    if (PHOTON && d.photon_noise) {
        uint32_t o[4], o2[4], o3[4] = {0u, 0u, 0u, 0u};
        quad_bits(quad, env, d, kDrawPhoton, o);
        quad_bits(quad, env, d, kDrawPhoton2, o2);
        if (__any(v[0] >= palias::kCoarseStep || v[1] >= palias::kCoarseStep || v[2] >= palias::kCoarseStep || v[3] >= palias::kCoarseStep))
            quad_bits(quad, env, d, kDrawPhoton3, o3);
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = fmaxf(v[s], 0.f);
        photon_quad(v, o, o2, o3, pv, env, d, pa.lmax, pa.tab);
    }
    f32x4d dark = {0.f, 0.f, 0.f, 0.f}, nrm = {0.f, 0.f, 0.f, 0.f};
    if (d.dark_e > 0.f) {
        uint32_t o[4], o2[4] = {0u, 0u, 0u, 0u};
        quad_bits(quad, env, d, kDrawDark, o);
        if (d.dark_e >= kPtrsFrom) quad_bits(quad, env, d, kDrawDark2, o2);
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            const uint32_t px = pv[s];
            dark[s] = d.dark_e < kPtrsFrom ? poisson_inversion(d.dark_e, u01(word_of(o, s)), rtab)
                                           : poisson_ptrs(d.dark_e, word_of(o, s), word_of(o2, s), px | 0x80000000u, env, d);
        }
    }
    if (d.readout_noise != 0.f) {
        uint32_t o[4];
        quad_bits(quad, env, d, kDrawReadout, o);
        quad_normals(o, nrm);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = detector_finish(v[s], d, dark[s], nrm[s]);
}

// Quad geometry of a Shack-Hartmann frame with 6-pixel lenslets: quad j = 0..8 of a lenslet whose top-left pixel is (y0, x0).
//   j < 6 : column j, rows 0..3            j >= 6 : rows 4, 5 of the columns c = j - 6 (slots 0, 1) and c + 3 (slots 2, 3)
__device__ inline void sh6_quad_pixels(int j, int y0, int x0, int cam, uint32_t (&pix)[4]) {
    if (j < 6) {
#pragma unroll
        for (int s = 0; s < 4; ++s) pix[s] = (uint32_t)((y0 + s) * cam + x0 + j);
    } else {
        const int c = j - 6;
        pix[0] = (uint32_t)((y0 + 4) * cam + x0 + c);
        pix[1] = (uint32_t)((y0 + 5) * cam + x0 + c);
        pix[2] = (uint32_t)((y0 + 4) * cam + x0 + c + 3);
        pix[3] = (uint32_t)((y0 + 5) * cam + x0 + c + 3);
    }
}

// host-side launcher (detector_kernels.hip): applies the camera to frame [E][cam*cam] in place; for a Shack-Hartmann
// frame (valid2d != null) it also writes the maximum over the valid lenslets' pixels to wfs_max[E].
template <typename T>
int launch_detector(T* frame, T* wfs_max, const uint8_t* valid2d, int n_env, int cam, int n_subap, const DetectorCfg& d,
                    const PoissonAlias& pa, hipStream_t st);

}  // namespace ao
