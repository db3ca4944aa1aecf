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
//   Xd:     inversion, at most 7 steps (P(Poisson(1/4) > 7) = 1e-10, below the resolution of the uniform), the lanes of a wave in
//           lock-step with a vote per step: P(Xd > 1) < 3 %, so a wave usually leaves after two or three.
//
// Three words per pixel (two below 32 photons), ~50 lane-instructions, no queues, no divergence: the lanes of a wave finish
// together whatever their pixels hold.  Probabilities are exact to the quantisation of the thresholds, 2^-23 / n of a row's
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
    return kmin + (frac < (en & ~511u) ? cell : (en & 511u));       // (frac >> 9) < threshold
}

__device__ inline float u01_23(uint32_t x) { return ((float)(x >> 9) + 0.5f) * (1.0f / 8388608.0f); }   // strictly inside (0, 1)

// The three parts of one draw, split so that a caller with several pixels per lane can run the remainders of all of them in ONE
// loop with a wave vote per step (below): lam = 32 c + j / 4 + d.
struct AliasParts { int fine_row, coarse_row; float d; };
__device__ inline AliasParts alias_parts(float lam) {
    const float c = floorf(lam * (1.0f / palias::kCoarseStep));
    const float r = fmaf(-palias::kCoarseStep, c, lam);                               // exact: [0, 32)
    const float jf = fminf(floorf(r * (1.0f / palias::kFineStep)), (float)(palias::kFineRows - 1));
    return {(int)jf, palias::kFineRows + (int)c, fmaxf(fmaf(-palias::kFineStep, jf, r), 0.f)};   // d exact: [0, 1/4)
}

// Poisson(d[s]), d < 1/4, for the NP pixels of a lane by inversion, P(t + 1) = P(t) d / (t + 1), the lanes of the wave in
// lock-step: a step is taken only while some pixel of some lane is still above its running sum (one scalar branch per step).
// P(X > 1) < 3 % at d = 1/4, so a wave typically leaves after two or three of the seven steps that cover the uniform's
// resolution (P(Poisson(1/4) > 7) = 1e-10).  EVERY lane of the wave must call.
template <int NP>
__device__ inline void poisson_small(const float (&d)[NP], const uint32_t (&w)[NP], uint32_t (&k)[NP]) {
    float p[NP], cdf[NP], u[NP];
#pragma unroll
    for (int s = 0; s < NP; ++s) {
        p[s] = __expf(-d[s]);
        cdf[s] = p[s];
        u[s] = u01_23(w[s]);
    }
#pragma unroll
    for (int t = 0; t < (AO_ABL(4) ? 1 : 7); ++t) {
        bool more = false;
#pragma unroll
        for (int s = 0; s < NP; ++s) more = more || u[s] > cdf[s];
        if (!__any(more)) break;
        const float rt = 1.0f / (float)(t + 1);
#pragma unroll
        for (int s = 0; s < NP; ++s) {
            k[s] += u[s] > cdf[s] ? 1u : 0u;
            p[s] *= d[s] * rt;
            cdf[s] += p[s];
        }
    }
}

// Poisson(v[s]), 0 <= v[s] < lmax of the table `tab` points to (LDS or global), for the four pixels of a quad.  wf, wr, wc: the
// quad's three draws (wc is only looked at where v[s] >= 32).  EVERY lane of the wave must call.
__device__ inline void poisson_alias4(const float (&v)[4], const uint32_t (&wf)[4], const uint32_t (&wr)[4], const uint32_t (&wc)[4],
                                      const uint32_t* __restrict__ tab, float (&out)[4]) {
    uint32_t k[4];
    float d[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const AliasParts a = alias_parts(v[s]);
        d[s] = a.d;
        k[s] = alias_draw(tab, a.fine_row, wf[s]) + alias_draw(tab, a.coarse_row, wc[s]);
    }
    poisson_small<4>(d, wr, k);
#pragma unroll
    for (int s = 0; s < 4; ++s) out[s] = (float)k[s];
}
#endif

}  // namespace ao
