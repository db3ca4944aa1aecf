// Photon-noise sampler of the WFS camera: Poisson(lambda) for a per-pixel lambda with a FIXED, branch-free cost.
//
//   OOPAO/Detector.py:204-206   frame = self.random_state_photon_noise.poisson(frame)
//
// The reference draws with NumPy's legacy generator (inversion below 10 photons, Hoermann's PTRS above) from a wall-clock seed, so a
// noisy frame is reproducible in distribution only; what has to match is the Poisson law.  Rounds 1-2 ran NumPy's two algorithms on
// the device: their trip counts depend on the data, and a wave runs as long as its slowest lane (243 lane-instructions per pixel,
// 22.6 us of the 60.9 us fused step at 256 envs, two LDS queues to finish the rejected PTRS rounds densely).  This sampler uses the
// additivity of the Poisson law instead:
//
//   lambda = 32 c + j / 4 + d,   c = floor(lambda / 32),  j = floor(4 (lambda - 32 c)) in 0..127,  d in [0, 1/4)
//   X = Xc + Xf + Xd,   Xc ~ Poisson(32 c),  Xf ~ Poisson(j / 4),  Xd ~ Poisson(d)   independent  =>  X ~ Poisson(lambda)  exactly
//
//   Xc, Xf: Walker / Vose ALIAS tables of the grid values (host, float64; rows cut where the tail mass is below 2^-34): one 32-bit
//           word w picks the cell floor(w n / 2^32) and compares the product's low bits with the cell's 23-bit threshold -- ONE
//           table read per draw, no loop.  128 fine rows + 32 coarse rows = 14.9 k words (58 KB): resident in LDS where a
//           workgroup draws thousands of pixels (fused step kernel, k_detector_sh6), read through the caches otherwise.
//   Xd:     inversion with 7 unrolled steps (P(Poisson(1/4) > 7) = 1e-10, below the resolution of the uniform).
//
// Three words per pixel (two below 32 photons), ~45 lane-instructions, no votes, no queues, no divergence: the lanes of a wave
// finish together whatever their pixels hold.  Probabilities are exact to the quantisation of the thresholds, 2^-23 / n of a row's
// mass per outcome (n >= 1 cells) -- the level of the float32 inversion it replaces.  Pixels at or above the table's end (`lmax`,
// 1024 photons at full size) go through PTRS (detector.hpp), whole waves at a time: rare at the flux of the BASELINE configs
// (brightest pixel of the 8 m / 20x20 loop at magnitude 8: ~860 photons).
#pragma once
#include <stdint.h>

#include <vector>

#include <hip/hip_runtime.h>

namespace ao {

namespace palias {
constexpr int kFineRows = 128;                 // lambda0 = j / 4, j = 0 .. 127
constexpr float kFineStep = 0.25f, kCoarseStep = 32.f;
constexpr int kMaxCoarseRows = 32;             // lambda0 = 32 c, c = 0 .. 31: the table ends at 1024 photons
constexpr int kHeader = 4;                     // words: fine rows, coarse rows, total words, reserved
constexpr int kMaxWords = 15360;               // capacity a kernel reserves for the whole table (60 KB)
}  // namespace palias

// by value into kernels
struct PoissonAlias {
    const uint32_t* tab;    // device copy of the table: header, row descriptors {first entry, kmin << 16 | cells}, entries {threshold << 9 | alias}
    int words;              // words of it the kernel may use (a kernel with less LDS copies a prefix: whole coarse rows are dropped from the end)
    float lmax;             // lambda < lmax is drawn from the first `words` words
};

// host: the table and, for every count of coarse rows kept, the words needed (words_upto[c] = size with coarse rows 0 .. c - 1)
struct PoissonAliasHost {
    std::vector<uint32_t> tab;
    std::vector<int> words_upto;   // [kMaxCoarseRows + 1]
    // the largest prefix that fits in `budget` words: its size and the lambda it reaches (0 rows of the coarse table = nothing usable)
    void prefix(int budget, int* words, float* lmax) const;
};
void build_poisson_alias(PoissonAliasHost& out);
const PoissonAliasHost& poisson_alias_host();   // built once per process

#ifdef __HIPCC__
#ifndef AO_ABL
#define AO_ABL(bit) 0                                             // (timing ablations, scripts/diag_cam_ablate.sh: wrong frames)
#endif
// one alias draw from row `row` with the 32-bit word w
__device__ inline uint32_t alias_draw(const uint32_t* __restrict__ tab, int row, uint32_t w) {
    uint2 d = *reinterpret_cast<const uint2*>(tab + palias::kHeader + 2 * row);
    if (AO_ABL(2)) d = uint2{(uint32_t)(palias::kHeader + 400 + 64 * row), 64u};
    if (AO_ABL(3)) return (uint32_t)(((uint64_t)w * d.y) >> 32);
    const uint32_t n = d.y & 0xffffu, kmin = d.y >> 16;
    const uint64_t prod = (uint64_t)w * n;                         // v_mad_u64_u32: cell and the fraction inside it in one instruction
    const uint32_t cell = (uint32_t)(prod >> 32), frac = (uint32_t)prod;
    const uint32_t en = tab[d.x + cell];
    return kmin + ((frac >> 9) < (en >> 9) ? cell : (en & 511u));
}

__device__ inline float u01_23(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }   // strictly inside (0, 1)

// Poisson(lam), 0 <= lam < lmax of the table `tab` points to (LDS or global).  wf, wr, wc: independent 32-bit words (wc is only
// looked at where lam >= 32).  COARSE = false: the caller knows lam < 32 for every lane.
template <bool COARSE = true>
__device__ inline float poisson_alias(float lam, uint32_t wf, uint32_t wr, uint32_t wc, const uint32_t* __restrict__ tab) {
    const float c = floorf(lam * (1.0f / palias::kCoarseStep));
    const float r = fmaf(-palias::kCoarseStep, c, lam);                               // exact: [0, 32)
    const float jf = fminf(floorf(r * (1.0f / palias::kFineStep)), (float)(palias::kFineRows - 1));
    const float dl = fmaxf(fmaf(-palias::kFineStep, jf, r), 0.f);                      // exact: [0, 1/4)
    uint32_t k = alias_draw(tab, (int)jf, wf);
    if (COARSE) k += alias_draw(tab, palias::kFineRows + (int)c, wc);
    // the remainder: inversion, P(t + 1) = P(t) d / (t + 1)
    float p = __expf(-dl), cdf = p;
    const float u = u01_23(wr);
#pragma unroll
    for (int t = 0; t < (AO_ABL(4) ? 1 : 7); ++t) {
        k += u > cdf ? 1u : 0u;
        p *= dl * (1.0f / (float)(t + 1));
        cdf += p;
    }
    return (float)k;
}
#endif

}  // namespace ao
