// The camera on the 12 pixels a lane owns in a Shack-Hartmann frame with 6-pixel lenslets (rows 0..5 of the lenslet columns q and
// q + 3: the lane -> pixel map of the fused step kernel), shared by that kernel and by k_detector_sh6: photon noise through the
// quad draws of detector.hpp ("Stream layout"), then QE, dark current, saturation, read-out noise and ADC.
//   pxv[p]: pixel p of the lane, p < 6 = (row p, column q), p >= 6 = (row p - 6, column q + 3): photons in, counts out; its three
//           quads: t = 0 rows 0..3 of column q, t = 1 rows 0..3 of column q + 3, t = 2 rows 4, 5 of both columns
//   ok:     the lane has pixels (a lenslet of its own); every lane of the WORKGROUP must call (barriers, wave votes)
//   px0:    frame index of the lane's pixel (row 0, column q);  cam: frame width;  e: env index of the noise streams
//   Q:      LDS of the workgroup's queues; Q.counts[WAVES + 1] must be zero on entry (and visible: a barrier after zeroing them)
// Faint pixels (lambda < 10) are drawn where they are met, by lock-step inversion; bright ones get the proposal and the squeeze of
// their first PTRS round there too, and the ones it leaves undecided (35-65 % at 10-100 photons) are finished by the whole workgroup in
// two dense passes -- per lane they kept a wave in the rejection loop until its slowest lane was through, with most lanes idle:
//   Q0 {lambda, frame pixel, U word, V word}: one entry per undecided pixel, in wave-private segments (no atomics); the pixel's register
//      keeps the entry's index.  Pass 1: the full acceptance test of round 0, three entries per lane;
//   Q1 {lambda, frame pixel, index in Q0}: what round 0 rejects (~40 % of Q0) goes on with its own stream until accepted (pass 2);
//   res[index in Q0]: the counts, picked up by the owners.
// The pixel loops are rolled (one copy of each sampler: 12 unrolled copies were 114 KB of code) and the lane's pixels are ONE vector
// register group indexed by the wave-uniform loop counter (v_movrel; arrays picked by select chains were demoted to scratch).
#pragma once
#include "detector.hpp"
#include "sh_device.hpp"

#ifndef AO_ABL
#define AO_ABL(bit) 0                                             // (timing ablations of the fused step kernel, scripts/diag_cam_ablate.sh)
#endif

namespace ao {

typedef float f32x16s __attribute__((ext_vector_type(16)));

struct CameraLds {
    f32x4d* q0;      // [q0_cap] entries of 16 bytes
    int q0_cap;
    float* res;      // [q0_cap]
    float* q1;       // [q1_cap][3]
    int q1_cap;
    int* counts;     // [WAVES + 1]: entries of every wave's segment of Q0, entries of Q1
};

template <int WAVES>
__device__ inline void camera_sh6_lane(f32x16s& pxv, bool ok, uint32_t px0, int cam, uint32_t e, const DetectorCfg& det, const CameraLds& Q,
                                       int w, int lane) {
    const float rtab = recip_table_lane();
    auto pix_of = [&](int p) { return px0 + (uint32_t)((p < 6 ? p : p - 6) * cam + (p < 6 ? 0 : 3)); };
    auto pixel_of_slot = [](int t, int sl4) { return t == 0 ? sl4 : (t == 1 ? 6 + sl4 : (sl4 < 2 ? 4 + sl4 : 8 + sl4)); };
    auto pick = [&](int p) { return pxv[p]; };
    auto put = [&](int p, float v) { pxv[p] = v; };
    // (fused step kernel: Q0 lives where Gy C and the layer tiles were, res and Q1 where E0 was -- dead once every wave has its spots:
    //  the barrier after the pixels are met)
    f32x4d* q0 = Q.q0;
    const int q0_cap = Q.q0_cap, q1_cap = Q.q1_cap;
    float* res = Q.res;
    float* q1 = Q.q1;                                                      // 3 words per entry
    int* q_count_s = Q.counts;
    int* q1_count = Q.counts + WAVES;
    uint32_t pending = 0, spilled = 0;                                   // spilled: bright, undecided, and the wave's part of Q0 was full
    // Q0 is cut into WAVES wave-private segments: a wave hands out its indices from a counter of its own (a wave-uniform
    // register) -- a shared counter was one LDS atomic round trip, with the wave waiting on it, per pixel position.
    const int seg_cap = q0_cap / WAVES, seg0 = w * seg_cap;
    int seg_n = 0;
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
#pragma unroll 1
        for (int t = 0; t < n_turns; ++t) {
            const uint32_t qid = px0 + (t == 1 ? 3u : (t == 2 ? (uint32_t)(4 * cam) : 0u));
            uint32_t o[4], o2[4];
            if (AO_ABL(0)) {
#pragma unroll
                for (int z = 0; z < 4; ++z) { o[z] = (qid + z) * 2654435761u + det.frame_counter; o2[z] = o[z] * 40503u + z; }
            } else {
                quad_bits(qid, e, det, kDrawPhoton, o);
                quad_bits(qid, e, det, kDrawPhoton2, o2);
            }
            int p4[4];
            f32x4d v4, out4 = {0.f, 0.f, 0.f, 0.f};
            bool faint[4];
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) {
                p4[sl] = pixel_of_slot(t, sl);
                v4[sl] = pick(p4[sl]);
                faint[sl] = !ok || !(v4[sl] >= kPtrsFrom);
            }
            // faint pixels: inversion, two pixels per turn as packed pairs
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const f32x2d lam2 = {faint[2 * h] && ok ? fmaxf(v4[2 * h], 0.f) : 0.f, faint[2 * h + 1] && ok ? fmaxf(v4[2 * h + 1], 0.f) : 0.f};
                f32x2d k2 = {0.f, 0.f};
                if (!AO_ABL(1) && __any(lam2.x > 0.f || lam2.y > 0.f)) k2 = poisson_inversion2(lam2, f32x2d{u01(o[2 * h]), u01(o[2 * h + 1])}, rtab);
                out4[2 * h] = k2.x;
                out4[2 * h + 1] = k2.y;
            }
            // bright pixels: proposal + squeeze of the first PTRS round for the quad's four pixels at once (no logarithm; four
            // independent dependency chains: a single one left the vector unit waiting on its square root and reciprocals)
            if (!AO_ABL(2) && __any(!faint[0] || !faint[1] || !faint[2] || !faint[3])) {
                bool und[4];
                unsigned long long bal[4];
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    float kb, us, V;
                    const bool sq = ptrs_squeeze(ptrs_const(faint[sl] ? kPtrsFrom : v4[sl]), o[sl], o2[sl], &kb, &us, &V);
                    und[sl] = !AO_ABL(3) && !faint[sl] && !sq;
                    out4[sl] = faint[sl] ? out4[sl] : kb;
                }
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) bal[sl] = __ballot(und[sl]);
                if ((bal[0] | bal[1] | bal[2] | bal[3]) != 0ull) {
#pragma unroll
                    for (int sl = 0; sl < 4; ++sl) {
                        if (und[sl]) {
                            const int idx = seg_n + __popcll(bal[sl] & ((1ull << lane) - 1ull));
                            if (idx < seg_cap) {
                                q0[seg0 + idx] = f32x4d{v4[sl], __uint_as_float(pix_of(p4[sl])), __uint_as_float(o[sl]), __uint_as_float(o2[sl])};
                                out4[sl] = __int_as_float(seg0 + idx);
                                pending |= 1u << p4[sl];
                            } else {
                                out4[sl] = v4[sl];
                                spilled |= 1u << p4[sl];
                            }
                        }
                        seg_n += __popcll(bal[sl]);
                    }
                }
            }
#pragma unroll
            for (int sl = 0; sl < 4; ++sl) put(p4[sl], out4[sl]);
        }
        if (lane == 0) q_count_s[w] = min(seg_n, seg_cap);
        if (__any(spilled != 0)) {                                        // Q0 full (a very bright star): these lanes' own work
#pragma unroll 1
            for (int p = 0; p < 12; ++p)
                if (spilled >> p & 1u) {
                    const int t = p < 4 ? 0 : (p >= 6 && p < 10 ? 1 : 2), sl4 = t == 0 ? p : (t == 1 ? p - 6 : (p < 6 ? p - 4 : p - 8));
                    const uint32_t qid = px0 + (t == 1 ? 3u : (t == 2 ? (uint32_t)(4 * cam) : 0u));
                    uint32_t o[4], o2[4];
                    quad_bits(qid, e, det, kDrawPhoton, o);
                    quad_bits(qid, e, det, kDrawPhoton2, o2);
                    put(p, poisson_ptrs(pick(p), word_of(o, sl4), word_of(o2, sl4), pix_of(p), e, det));
                }
        }
        lds_barrier();                                                    // every wave has its spots: E0 is free; Q0 is complete
        int n_q0 = 0;
#pragma unroll
        for (int w2 = 0; w2 < WAVES; ++w2) n_q0 += q_count_s[w2];
        if (AO_ABL(4)) n_q0 = 0;
        // pass 1: round 0's full test, THREE entries per lane and turn (independent chains of logarithms and reciprocals; ~2500 entries: one turn);
        // accepted -> res, rejected -> Q1 (or, Q1 full, finished on the spot)
        for (int i0 = 64 * w; i0 < n_q0; i0 += 3 * 64 * WAVES) {
            int slot2[3];
            bool live2[3], done2[3];
            float kf2[3], lam2[3], pix2[3];
#pragma unroll
            for (int z = 0; z < 3; ++z) {
                const int i = i0 + 64 * WAVES * z + lane;
                live2[z] = i < n_q0;
                int slot = -1, rem = i, base = 0;                            // dense index -> (segment, entry)
#pragma unroll
                for (int w2 = 0; w2 < WAVES; ++w2) {
                    const int cw = q_count_s[w2];
                    slot = (slot < 0 && rem < cw) ? base + rem : slot;
                    rem -= cw;
                    base += seg_cap;
                }
                slot2[z] = live2[z] ? slot : 0;
            }
#pragma unroll
            for (int z = 0; z < 3; ++z) {
                const f32x4d it = q0[slot2[z]];
                lam2[z] = live2[z] ? it[0] : kPtrsFrom;
                pix2[z] = it[1];
                const PtrsConst c = ptrs_const(lam2[z]);
                float us, V;
                ptrs_squeeze(c, __float_as_uint(it[2]), __float_as_uint(it[3]), &kf2[z], &us, &V);       // (known: not accepted)
                done2[z] = ptrs_full(c, kf2[z], us, V, ptrs_logs(c));
            }
            bool rej3[3];
            unsigned long long bal3[3];
#pragma unroll
            for (int z = 0; z < 3; ++z) {
                rej3[z] = live2[z] && !done2[z];
                bal3[z] = __ballot(rej3[z]);
            }
            int b1 = 0;
            if ((bal3[0] | bal3[1] | bal3[2]) != 0ull && lane == 0) b1 = atomicAdd(q1_count, __popcll(bal3[0]) + __popcll(bal3[1]) + __popcll(bal3[2]));
            b1 = __builtin_amdgcn_readfirstlane(b1);
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (int z = 0; z < 3; ++z) {
                const bool rej = rej3[z];
                const int slot1 = rej ? b1 + __popcll(bal3[z] & below) : -1;
                b1 += __popcll(bal3[z]);
                const bool queued = slot1 >= 0 && slot1 < q1_cap;
                if (queued) {
                    q1[3 * slot1] = lam2[z];
                    q1[3 * slot1 + 1] = pix2[z];
                    q1[3 * slot1 + 2] = __int_as_float(slot2[z]);
                }
                if (__any(rej && !queued)) {                                   // Q1 full: finished here, by the lanes that drew a slot beyond it
                    const PtrsConst c = ptrs_const(lam2[z]);
                    kf2[z] = poisson_ptrs_rounds(c, ptrs_logs(c), !rej || queued, kf2[z], __float_as_uint(pix2[z]), e, det);
                }
                if (live2[z] && !queued) res[slot2[z]] = kf2[z];
            }
        }
        lds_barrier();
        const int n_q1 = min(*q1_count, q1_cap);
        for (int i0 = 64 * w; i0 < n_q1; i0 += 64 * WAVES) {                     // pass 2: whole waves, the sampler votes across the wave
            const int i = i0 + lane;
            const bool live = i < n_q1;
            const float lam1 = live ? q1[3 * i] : kPtrsFrom;
            const uint32_t pix1 = __float_as_uint(q1[3 * (live ? i : 0) + 1]);
            const int orig = __float_as_int(q1[3 * (live ? i : 0) + 2]);
            const PtrsConst c = ptrs_const(lam1);
            const float kf = poisson_ptrs_rounds(c, ptrs_logs(c), !live, floorf(lam1 + 0.5f), pix1, e, det);
            if (live) res[orig] = kf;
        }
        lds_barrier();
        // the owners pick their counts up: all 12 reads in flight at once (the register holds the entry's index)
        if (__any(pending != 0u)) {
            float got[12];
#pragma unroll
            for (int p = 0; p < 12; ++p) got[p] = res[(pending >> p & 1u) ? __float_as_int(pxv[p]) : 0];
#pragma unroll
            for (int p = 0; p < 12; ++p) pxv[p] = (pending >> p & 1u) ? got[p] : pxv[p];
        }
    }
    // QE, dark shot noise, saturation, gain, read-out noise, ADC: quad by quad (dark_e < kPtrsFrom: checked by the host)
    const bool has_dark = det.dark_e > 0.f, has_read = det.readout_noise != 0.f;
    if (has_dark || has_read || det.qe != 1.f || det.gain != 1.f || det.fwc > 0.f || det.bits > 0) {
#pragma unroll 1
        for (int t = 0; t < 3; ++t) {
            const uint32_t qid = px0 + (t == 1 ? 3u : (t == 2 ? (uint32_t)(4 * cam) : 0u));
            f32x4d dark = {0.f, 0.f, 0.f, 0.f}, nrm = {0.f, 0.f, 0.f, 0.f};
            if (has_dark) {
                uint32_t o[4];
                quad_bits(qid, e, det, kDrawDark, o);
#pragma unroll 1
                for (int sl4 = 0; sl4 < 4; ++sl4)
                    dark[sl4] = det.dark_e < kPtrsFrom ? poisson_inversion(det.dark_e, u01(word_of(o, sl4)), rtab)
                                                      : poisson_ptrs(det.dark_e, word_of(o, sl4), ~word_of(o, sl4) * 0x9E3779B9u, pix_of(pixel_of_slot(t, sl4)) | 0x80000000u, e, det);
            }
            if (has_read) {
                uint32_t o[4];
                quad_bits(qid, e, det, kDrawReadout, o);
                quad_normals(o, nrm);
            }
#pragma unroll 1
            for (int sl4 = 0; sl4 < 4; ++sl4) {
                const int p = pixel_of_slot(t, sl4);
                put(p, detector_finish(pick(p), det, dark[sl4], nrm[sl4]));
            }
        }
    }
}

}  // namespace ao
