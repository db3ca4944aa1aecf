// The camera on the 12 pixels a lane owns in a Shack-Hartmann frame with 6-pixel lenslets (rows 0..5 of the lenslet columns q and
// q + 3: the lane -> pixel map of the fused step kernel), shared by that kernel and by k_detector_sh6: photon noise through the
// quad draws of detector.hpp ("Stream layout"), then QE, dark current, saturation, read-out noise and ADC.
//   pxv:    the lane's pixels in quad-major order (camera_pack): photons in, counts out; its three quads: t = 0 rows 0..3 of
//           column q, t = 1 rows 0..3 of column q + 3, t = 2 rows 4, 5 of both columns
//   ok:     the lane has pixels (a lenslet of its own); every lane of a WAVE must call (the PTRS hand-over votes across the wave)
//   px0:    frame index of the lane's pixel (row 0, column q);  cam: frame width;  e: env index of the noise streams
//   tab:    the alias tables of poisson_alias.hpp where the calling kernel keeps them (LDS), complete and visible on entry;
//           lmax: the photon count its copy reaches
// Photon counts come from the fixed-cost alias sampler: three table reads and ~45 instructions per pixel whatever it holds, so the
// lanes of a wave finish together.  (Rounds 1-2 ran inversion / PTRS here, a wave as long as its slowest lane, with two workgroup
// queues in LDS and three barriers to finish the rejected PTRS rounds densely: 22.6 us of the 60.9 us step at 256 envs.)
// PTRS, for pixels beyond the table, is ONE rolled block behind a wave vote.
#pragma once
#include "detector.hpp"
#include "sh_device.hpp"

#ifndef AO_ABL
#define AO_ABL(bit) 0                                             // (timing ablations of the fused step kernel, scripts/diag_cam_ablate.sh)
#endif

namespace ao {

typedef float f32x16s __attribute__((ext_vector_type(16)));

// The lane's 12 pixels in QUAD-MAJOR order, pxv[4 t + s] = slot s of quad t:
//   t = 0: rows 0..3 of column q;  t = 1: rows 0..3 of column q + 3;  t = 2: rows 4, 5 of column q then rows 4, 5 of column q + 3
// (Ia[u] / Ib[u] = row u of the columns q / q + 3, the layout of the spots and of the centroid)
__device__ inline f32x16s camera_pack(const float (&Ia)[6], const float (&Ib)[6]) {
    return f32x16s{Ia[0], Ia[1], Ia[2], Ia[3], Ib[0], Ib[1], Ib[2], Ib[3], Ia[4], Ia[5], Ib[4], Ib[5], 0.f, 0.f, 0.f, 0.f};
}
__device__ inline void camera_unpack(const f32x16s& v, float (&Ia)[6], float (&Ib)[6]) {
    Ia[0] = v[0]; Ia[1] = v[1]; Ia[2] = v[2]; Ia[3] = v[3]; Ib[0] = v[4]; Ib[1] = v[5]; Ib[2] = v[6]; Ib[3] = v[7];
    Ia[4] = v[8]; Ia[5] = v[9]; Ib[4] = v[10]; Ib[5] = v[11];
}
// the lane's pixels one quad on: what was quad t + 1 is now in elements 0..3 (three turns restore the order)
__device__ inline void camera_rotate(f32x16s& v) {
    v = f32x16s{v[4], v[5], v[6], v[7], v[8], v[9], v[10], v[11], v[0], v[1], v[2], v[3], 0.f, 0.f, 0.f, 0.f};
}

__device__ inline void camera_sh6_lane(f32x16s& pxv, bool ok, uint32_t px0, int cam, uint32_t e, const DetectorCfg& det,
                                       const uint32_t* __restrict__ tab, float lmax) {
    // frame index of the pixel in element p = 4 t + s of the (unrotated) vector
    auto pix_of = [&](int p) {
        const int t = p >> 2, s4 = p & 3;
        return px0 + (uint32_t)((t < 2 ? s4 : 4 + (s4 & 1)) * cam + ((t == 1 || (t == 2 && s4 >= 2)) ? 3 : 0));
    };
    auto quad_id = [&](int t) { return px0 + (t == 1 ? 3u : (t == 2 ? (uint32_t)(4 * cam) : 0u)); };
    if (det.photon_noise) {
        // (a wave without light -- lenslets outside the pupil in the stand-alone kernel -- draws nothing: Poisson(0) = 0)
        float brightest = 0.f;
#pragma unroll
        for (int p = 0; p < 12; ++p) brightest = fmaxf(brightest, pxv[p]);
        const int n_turns = __any(ok && brightest > 0.f) ? 3 : 0;
        if (n_turns == 0) {
#pragma unroll
            for (int p = 0; p < 12; ++p) pxv[p] = 0.f;
        }
        // One quad per turn of a ROLLED loop (one copy of the sampler in the instruction cache); the turn works on elements 0..3
        // and rotates the vector: static register indices (picking the quad by the loop counter compiled to ladders of scalar
        // branches, unrolling the turns to spills).
        uint32_t over = 0;                                            // pixels at or above the table's end: PTRS, below
#pragma unroll 1
        for (int t = 0; t < n_turns; ++t) {
            const uint32_t qid = quad_id(t);
            float v4[4];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) v4[sl] = ok ? fmaxf(pxv[sl], 0.f) : 0.f;
            uint32_t o[4], o2[4], o3[4] = {0u, 0u, 0u, 0u};
            if (AO_ABL(0)) {
#pragma unroll
                for (int z = 0; z < 4; ++z) { o[z] = (qid + z) * 2654435761u + det.frame_counter; o2[z] = o[z] * 40503u + z; o3[z] = o2[z] * 69069u + 1u; }
            } else {
                quad_bits(qid, e, det, kDrawPhoton, o);
                quad_bits(qid, e, det, kDrawPhoton2, o2);
                if (__any(fmaxf(fmaxf(v4[0], v4[1]), fmaxf(v4[2], v4[3])) >= palias::kCoarseStep)) quad_bits(qid, e, det, kDrawPhoton3, o3);
            }
            float lam4[4], k4[4];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                const bool ov = v4[sl] >= lmax;
                over |= ov ? 1u << (4 * t + sl) : 0u;
                lam4[sl] = ov ? 0.f : v4[sl];
            }
            if (AO_ABL(1)) {
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) k4[sl] = v4[sl];
            } else {
                poisson_alias4(lam4, o, o2, o3, tab, k4);
            }
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) pxv[sl] = v4[sl] >= lmax ? v4[sl] : k4[sl];
            camera_rotate(pxv);
        }
        if (__any(over != 0u)) {                                      // (a very bright star: rare at the flux of the BASELINE configs)
#pragma unroll 1
            for (int p = 0; p < 12; ++p) {
                if (!__any((over >> p) & 1u)) continue;
                const uint32_t qid = quad_id(p >> 2);
                uint32_t o[4], o3[4];
                quad_bits(qid, e, det, kDrawPhoton, o);
                quad_bits(qid, e, det, kDrawPhoton3, o3);
                const bool big = (over >> p) & 1u;
                const float kb = poisson_ptrs(big ? pxv[p] : kPtrsFrom, word_of(o, p & 3), word_of(o3, p & 3), pix_of(p), e, det);
                pxv[p] = big ? kb : pxv[p];
            }
        }
    }
    // QE, dark shot noise, saturation, gain, read-out noise, ADC: quad by quad, the same rotation
    const bool has_dark = det.dark_e > 0.f, has_read = det.readout_noise != 0.f;
    if (has_dark || has_read || det.qe != 1.f || det.gain != 1.f || det.fwc > 0.f || det.bits > 0) {
        const float rtab = recip_table_lane();
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {
            const uint32_t qid = quad_id(t);
            f32x4d dark = {0.f, 0.f, 0.f, 0.f}, nrm = {0.f, 0.f, 0.f, 0.f};
            if (has_dark) {
                uint32_t o[4], o2[4] = {0u, 0u, 0u, 0u};
                quad_bits(qid, e, det, kDrawDark, o);
                if (det.dark_e >= kPtrsFrom) quad_bits(qid, e, det, kDrawDark2, o2);
#pragma unroll 1
                for (int sl4 = 0; sl4 < 4; ++sl4)
                    dark[sl4] = det.dark_e < kPtrsFrom ? poisson_inversion(det.dark_e, u01(word_of(o, sl4)), rtab)
                                                      : poisson_ptrs(det.dark_e, word_of(o, sl4), word_of(o2, sl4), pix_of(4 * t + sl4) | 0x80000000u, e, det);
            }
            if (has_read) {
                uint32_t o[4];
                quad_bits(qid, e, det, kDrawReadout, o);
                quad_normals(o, nrm);
            }
#pragma unroll
            for (int sl4 = 0; sl4 < 4; ++sl4) pxv[sl4] = detector_finish(pxv[sl4], det, dark[sl4], nrm[sl4]);
            camera_rotate(pxv);
        }
    }
}

// the alias tables into LDS by direct loads (global_load_lds: no registers, the wave goes on): every wave of the workgroup copies
// its share of the first `words` (a multiple of 4) words; complete for the workgroup after each wave's `s_waitcnt vmcnt(0)` and
// a barrier.  `lds` is 16-byte aligned.
typedef __attribute__((address_space(1))) const void* ao_gptr_t;
typedef __attribute__((address_space(3))) void* ao_lptr_t;
__device__ inline void alias_table_to_lds(const uint32_t* __restrict__ tab, int words, uint32_t* lds, int wave, int n_waves, int lane) {
    for (int i0 = wave * 256; i0 < words; i0 += n_waves * 256) {
        const int i = i0 + 4 * lane;                                // (the LDS address is the wave's base + 16 lane: a lane past the end is masked off)
        if (i < words) __builtin_amdgcn_global_load_lds((ao_gptr_t)(tab + i), (ao_lptr_t)(lds + i0), 16, 0, 0);
    }
}

}  // namespace ao
