// The whole env.step() of one AO loop in ONE workgroup: one launch per step, one workgroup (16 waves) per env.
//
//   MAIN/OOPAOEnv/OOPAOEnv.py:485-536 (step)  =  atmosphere sub-pixel translation + footprint + layer sum
//   (OOPAO/Atmosphere.py:406-407, 439-450, 474-477)  ->  DM surface and residual phase (DeformableMirror.py:556, 469;
//   Telescope.py:404-412, 540-542)  ->  Shack-Hartmann spots, camera frame, thresholded centre of gravity, slopes
//   (ShackHartmann.py:340-347, 539-601)  ->  reconstruction, reward, leaky integrator, Strehl / rms telemetry
//   (OOPAOEnv.py:491-536).
//
// Why one workgroup per env: at the BASELINE sizes (R = 120: 57.6 KB of phase, 316 valid lenslets) every intermediate
// of an env fits in the 160 KB of LDS of one CU, so the residual phase never has to be re-read from HBM by the WFS,
// the camera frame never by the centroid, and the five dependent launches of the separate-kernel path (each mostly
// launch + memory latency at a few hundred envs) collapse into one.  256 envs = one workgroup on each of the 256 CUs.
//
//   stage A (2 passes of 64 rows): layer tile -> LDS, separable Catmull-Rom, DM surface s1 . Gx^T on the matrix
//            cores (v_mfma_f32_16x16x4_f32, whose output layout is the lane -> pixel map), pupil, phase store,
//            E0 = amp e^{i phi} parked in LDS per lenslet, float64 telemetry sums
//   stage B  3 lanes per valid lenslet (948 of 1024 lanes at 316 lenslets): register-resident 12 x 12 DFT + 2 x 2
//            binning (sh_device.hpp), frame store, workgroup max -> threshold -> centre of gravity in registers
//   stage C  t = M s, o = -M2C t, integrator, observation image, reward, telemetry (tail_from_slopes)
//
// Arithmetic is the same as in the separate kernels (phase_kernel.hip, sh_kernels.hip); only summation orders of the
// centroid and of the telemetry differ (float32 / float64 rounding level).  No atomics: bitwise reproducible.
// diagnostic ablation of the camera block (scripts/diag_cam_ablate.sh): -DAO_CAM_ABLATE=<bit mask>; wrong frames, timing only
#ifdef AO_CAM_ABLATE
#define AO_ABL(bit) (((AO_CAM_ABLATE) >> (bit)) & 1)
#else
#define AO_ABL(bit) 0
#endif
#include "common.hpp"

// Diagnostic build only (-DAO_STEP_STAMPS): wave 0 of each workgroup stamps s_memtime at the stage boundaries.
#ifdef AO_STEP_STAMPS
namespace ao { __device__ unsigned long long g_stamps[1024 * 32]; __device__ unsigned long long g_wstamps[256 * 16 * 8]; }
#define AO_WSTAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) ::ao::g_wstamps[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define AO_STAMP(i) do { if (tid == 0 && e < 1024) ::ao::g_stamps[e * 32 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AO_STAMP(i) do { } while (0)
#define AO_WSTAMP(i) do { } while (0)
#endif

#include "sh_device.hpp"
#include "detector.hpp"
#include "ring_device.hpp"

#include "camera_sh6.hpp"


namespace ao {

typedef float f32x4s __attribute__((ext_vector_type(4)));

// p[idx] if ok else 0, as an UNCONDITIONAL load from a clamped (always valid) index: a predicated load compiles to a
// branch around the load, and a run of them to a chain of dependent memory latencies.
template <typename P>
__device__ inline float ld_or0(const P* __restrict__ p, long idx, bool ok) {
    const float v = (float)p[ok ? idx : 0];
    return ok ? v : 0.f;
}

namespace fstep {
constexpr int TX = 128, PR = 64;                       // widest pupil, rows per pass
constexpr int WR = 16 + 3, WC = 32 + 4;                // a wave's private layer tile: 19 rows x 35 columns (stride 36)
constexpr int WC4 = WC / 4;                            // float4 per staged row of the wave tile
constexpr int NV4 = (WR * WC4 + 63) / 64;              // 16-byte staging loads per lane and layer
}  // namespace fstep

struct StepLds {
    int cimg, s1, mapt, slot, e0, sl, img, total;      // offsets in 4-byte words
    int tab, tab_cap;                                  // the camera's alias tables (stage B) and the words there are for them
};

static StepLds step_lds_layout(int n_act, int n_subap, int n_valid, int n_modes) {
    using namespace fstep;
    StepLds L;
    const int nAp = (n_act + 3) & ~3, SS = nAp + 1;
    int o = 0;
    auto take = [&](int words) { const int at = o; o += (words + 3) & ~3; return at; };
    // What stage A leaves behind is dead at the barrier in front of stage B: the command image, Gy C (s1) and the waves' layer
    // tiles make room for the alias tables of the camera's photon draw (poisson_alias.hpp; copied there by direct loads while the
    // spots are computed).  The slopes and the observation image / modal coefficients of stages B-C live where the lenslet
    // fields (E0) were: written only behind the barrier that follows the spots of every wave.
    const int sl_words = (2 * n_valid + 3) & ~3, img_words = n_act * n_act + n_modes;
    L.cimg = take(n_act * n_act);
    L.s1 = take(2 * PR * SS);
    L.mapt = take(16 * WR * WC);
    L.tab = L.cimg;
    L.tab_cap = o - L.tab;
    L.slot = take((n_subap * n_subap + 1) / 2);
    L.e0 = take(std::max(2 * n_valid * fast6::EST, sl_words + img_words));
    L.sl = L.e0;
    L.img = L.e0 + sl_words;
    L.total = o;
    return L;
}

template <bool PE>
__device__ inline const LayerTaps& step_taps(const PhaseArgs& pa, int l, int e) {
    if constexpr (PE) return pa.env_taps[(size_t)l * pa.n_env + e];
    else return pa.taps[l];
}

// KS: k steps (of 4) of the DM product whose B operands are held in registers, n_act <= 4 KS
// PE: per-env clocks (aoenv_set_wind_env): the layer taps come from the env's own record in global memory instead of the kernel
// arguments (a variant of its own: read through a pointer chosen at run time, the shared-clock path lost 3 %)
template <int KS, bool PE>
__global__ void __launch_bounds__(1024) k_env_step_sh6(const StepArgs a, const StepLds L) {
    using namespace fstep;
    using fast6::EST;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float* lds = reinterpret_cast<float*>(lds_raw);
    const KArgs<float>& k = a.k;
    const int R = k.R, nA = k.n_act, S = k.pa.S;
    const int nAp = (nA + 3) & ~3, SS = nAp + 1;
    float* cimg = lds + L.cimg;                                  // [nA][nA] command image
    float* s1 = lds + L.s1;                                      // [2 PR][SS]  Gy C, every row
    const int w_ = threadIdx.x >> 6;
    float* mapt = lds + L.mapt + w_ * (WR * WC);                   // [WR][WC] this wave's private layer tile
    short* slot_s = reinterpret_cast<short*>(lds + L.slot);      // [nSub^2] lenslet -> compact valid index or -1
    cplx<float>* E0 = reinterpret_cast<cplx<float>*>(lds + L.e0);    // [nValid][EST]
    float* sl = lds + L.sl;                                      // [2 nValid] slopes            (aliases E0: behind the threshold barrier)
    float* img_s = lds + L.img;                                  // [nA^2 + n_modes]             (aliases E0: stage C)
    const uint32_t* tab_s = reinterpret_cast<const uint32_t*>(lds + L.tab);   // camera: alias tables (alias cimg, s1, the layer tiles: stage B)
    __shared__ double red[4][16];
    __shared__ double red_tail[16];
    __shared__ float red_mx[16];

    const int e = blockIdx.x;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int lc = lane & 15, lq = lane >> 4;                    // MFMA lane decomposition
    const int band = w >> 2, cg = w & 3;                         // 16-row band of the pass, 32-column group
    const size_t pix0 = (size_t)e * R * R;
    const int n_sub = a.n_subap, n_valid = a.n_valid;

    AO_STAMP(0);
    AO_WSTAMP(0);
    // ---- prologue: every global load of the prologue is issued before the first barrier ------------------------------------
    // (the barriers are compiler fences for memory operations: a load written after one is issued after it)
    // A operands of the DM product, Gx[x][k = lane >> 4 + 4 step]: the lane's two column tiles are the same in every
    // pass, so they are loaded once, long before the matrix cores need them.  The host re-lays the influence factors
    // out as operand tables (ga_index(), common.hpp): a lane's 6..8 values are two coalesced 16-byte loads, not 6..8 dwords.
    const float ctab = fast6::cos_table_lane<float>();           // stage B twiddles, fetched by wave shuffles there
    f32x4s gx_raw[2][2], gy_raw[2];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const f32x4s* src = reinterpret_cast<const f32x4s*>(a.gxa) + (2 * cg + tt) * 2 * 64 + (tid & 63);   // ga_index(), stride 8
        gx_raw[tt][0] = src[0];
        gx_raw[tt][1] = src[64];
    }
    const int rt = w >> 1, ct = w & 1;                           // s1 tile of this wave: rows 16 rt.., command columns 16 ct..
    {
        const f32x4s* src = reinterpret_cast<const f32x4s*>(a.gya) + rt * 2 * 64 + (tid & 63);   // gy[16 rt + lc][lq + 4 step]
        gy_raw[0] = src[0];
        gy_raw[1] = src[64];
    }
    float mm_pre[kMaxLayer];                                     // stored range of every layer's screen (lane & 1: min / max)
#pragma unroll
    for (int l = 0; l < kMaxLayer; ++l)
        mm_pre[l] = l < k.pa.n_layer ? static_cast<const float*>(k.pa.minmax[l])[2 * e + (tid & 1)] : 0.f;
    const bool has_act = tid < k.n_valid_act;                    // n_valid_act <= 1024 (n_act <= 32)
    const int act_px = k.pb.act_idx[has_act ? tid : 0];
    const float act_c = k.pb.coefs[(size_t)e * k.n_valid_act + (has_act ? tid : 0)];
    const short slot_v = a.slot_of[tid < n_sub * n_sub ? tid : 0];
    for (int i = tid; i < nA * nA; i += 1024) cimg[i] = 0.f;
    for (int i = tid; i < 2 * PR * SS; i += 1024) s1[i] = 0.f;
    if (tid < n_sub * n_sub) slot_s[tid] = slot_v;
    for (int i = tid + 1024; i < n_sub * n_sub; i += 1024) slot_s[i] = a.slot_of[i];
    lds_barrier();
    AO_STAMP(24);
    if (has_act) cimg[act_px] = act_c;
    float breg[2][KS];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const float t[8] = {gx_raw[tt][0][0], gx_raw[tt][0][1], gx_raw[tt][0][2], gx_raw[tt][0][3],
                            gx_raw[tt][1][0], gx_raw[tt][1][1], gx_raw[tt][1][2], gx_raw[tt][1][3]};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) breg[tt][ks] = t[ks];
    }

    // ---- s1[y][ix] = sum_iy gy[y][iy] C[iy][ix] for every row, on the matrix cores: 8 x 2 tiles of 16 x 16, one per wave ---
    lds_barrier();                                             // cimg complete
    AO_STAMP(25);
    {
        const int ix = 16 * ct + lc;
        float av[KS], bv[KS];
        {
            const float t[8] = {gy_raw[0][0], gy_raw[0][1], gy_raw[0][2], gy_raw[0][3], gy_raw[1][0], gy_raw[1][1], gy_raw[1][2], gy_raw[1][3]};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) av[ks] = t[ks];                                   // A[i = lane & 15][k = lane >> 4]
        }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int iy = lq + 4 * ks;
            bv[ks] = ld_or0(cimg, iy * nA + ix, iy < nA && ix < nA);                      // B[k = lane >> 4][j = lane & 15]
        }
        f32x4s d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], d, 0, 0, 0);
        if (ix < nA) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s1[(16 * rt + 4 * lq + r) * SS + ix] = d[r];
        }
    }

    AO_STAMP(26);
    // ---- deferred ring scatter: map_full[outerMask] = X of a layer that crossed a pixel this step ----------------------------
    // The ring values are the fixed-order sum of the GEMM's split-K slabs, written through the torus origin.  This
    // workgroup is the only reader of this env's map in this launch, and its vector L1 holds no line of it yet (no load of
    // the map has been issued in this kernel): after the stores are acknowledged (vmcnt 0) and the barrier, the loads below
    // read them back from L2.
    {
        bool any = false;
        for (int l = 0; l < k.pa.n_layer; ++l) {
            const float* x = a.ring_x[l];
            if (x == nullptr) continue;
            any = true;                                           // (uniform over the workgroup: the barrier below)
            const LayerTaps& tl = step_taps<PE>(k.pa, l, e);
            if (PE && !tl.ring) continue;              // per-env clocks: this env's layer did not cross a pixel
            float* map = const_cast<float*>(static_cast<const float*>(k.pa.screen[l])) + (size_t)e * S * S;
            const size_t slab = (size_t)a.n_env * a.n_outer;
            const int oy = tl.oy, ox = tl.ox;
            for (int q = tid; q < a.n_outer; q += 1024) {
                float xv[kMaxSplits];                             // all slabs in flight at once, summed in slab order
#pragma unroll
                for (int z = 0; z < kMaxSplits; ++z)
                    xv[z] = x[(size_t)(z < a.ring_splits[l] ? z : 0) * slab + (size_t)e * a.n_outer + q];
                float v = xv[0];
#pragma unroll
                for (int z = 1; z < kMaxSplits; ++z) v += z < a.ring_splits[l] ? xv[z] : 0.f;
                const int idx = a.outer_idx[q];
                int pr = idx / S + oy, pc = idx % S + ox;
                pr = pr >= S ? pr - S : pr;
                pc = pc >= S ? pc - S : pc;
                map[pr * S + pc] = v;
            }
        }
        if (any) __syncthreads();                                // s_waitcnt vmcnt(0) lgkmcnt(0) + barrier: stores acknowledged
        // Z of the layer's next crossing (env.hip: ring pipeline): the two rings inside the new border, through the new origin
        if (!PE) {
            for (int l = 0; l < k.pa.n_layer; ++l)
                if (a.next_zx[l] != nullptr)
                    gather_ring<float>(static_cast<const float*>(k.pa.screen[l]), a.next_zx[l], a.inner_idx, S, a.n_inner, a.zx_ld,
                                       a.next_sx[l], a.next_sy[l], k.pa.taps[l].oy, k.pa.taps[l].ox, e, tid, 1024);
        }
    }

    AO_STAMP(27);
    // ---- range of every layer's screen (the warp clips to it): read back, or recomputed here after a ring extrusion -------
    __shared__ float lohi[kMaxLayer][2];
    __shared__ float red_lo[16], red_hi[16];
    for (int l = 0; l < k.pa.n_layer; ++l) {
        float* mm = const_cast<float*>(static_cast<const float*>(k.pa.minmax[l])) + 2 * e;
        const bool dirty = k.pa.minmax_dirty[l] || (PE && a.ring_x[l] != nullptr && step_taps<PE>(k.pa, l, e).ring);
        if (!dirty) {
            float pre = 0.f;
#pragma unroll
            for (int q = 0; q < kMaxLayer; ++q) pre = q == l ? mm_pre[q] : pre;
            if (tid < 2) lohi[l][tid] = pre;
            continue;
        }
        // the torus is a permutation of the (N+2)^2 pixels: scan it physically with 16-byte loads
        const float* map = static_cast<const float*>(k.pa.screen[l]) + (size_t)e * S * S;
        const int n4 = (S * S) / 4;                              // env maps are 16-byte aligned when S*S % 4 == 0
        float lo = 3.0e38f, hi = -3.0e38f;
        if ((S * S) % 4 == 0) {
            for (int i0 = tid; i0 < n4; i0 += 4 * 1024) {           // 4 independent 16-byte loads per lane in flight
                f32x4s v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f32x4s*>(map + 4 * (i0 + 1024 * u < n4 ? i0 + 1024 * u : i0));
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int d = 0; d < 4; ++d) { lo = v[u][d] < lo ? v[u][d] : lo; hi = v[u][d] > hi ? v[u][d] : hi; }
            }
        } else {
            for (int i = tid; i < S * S; i += 1024) { const float v = map[i]; lo = v < lo ? v : lo; hi = v > hi ? v : hi; }
        }
        for (int off = 32; off > 0; off >>= 1) {
            const float ol = __shfl_down(lo, off), oh = __shfl_down(hi, off);
            lo = ol < lo ? ol : lo;
            hi = oh > hi ? oh : hi;
        }
        lds_barrier();                                           // red_lo / red_hi of the previous layer consumed
        if (lane == 0) { red_lo[w] = lo; red_hi[w] = hi; }
        lds_barrier();
        if (tid == 0) {
            for (int i = 1; i < 16; ++i) { lo = red_lo[i] < lo ? red_lo[i] : lo; hi = red_hi[i] > hi ? red_hi[i] : hi; }
            lohi[l][0] = lo; lohi[l][1] = hi;
            mm[0] = lo; mm[1] = hi;
        }
    }

    // Lane -> pixel map of a pass (64 rows): wave = (band = w >> 2: 16 rows, cg = w & 3: 32 columns = 2 tiles of 16);
    // in a tile the lane owns ONE row y = 16 band + (lane & 15) and FOUR CONSECUTIVE columns x = 16 tile + 4 (lane >> 4) + r.
    // That is the output layout of the matrix cores for D[x][y] = sum_k Gx^T[k][x] s1[y][k] (rows 4 (lane >> 4) + r = x,
    // column lane & 15 = y), and it makes every global access of the pass a 16-byte one (the vector-memory path takes
    // ~16 cycles per wave-instruction whatever the width: dword accesses were 3/4 of this kernel's time).
    double s_atm = 0.0, q_atm = 0.0, s_res = 0.0, q_res = 0.0;
    for (int pass = 0; pass < (R + PR - 1) / PR; ++pass) {
        const int y0 = pass * PR;
        const int tye = min(PR, R - y0);
        const int yl = 16 * band + lc, y = y0 + yl;              // the lane's row
        const bool row_ok = yl < tye;
        if (pass == 0) { AO_WSTAMP(1); lds_barrier(); AO_WSTAMP(2); }   // s1 and the screen ranges complete
        AO_STAMP(1 + 6 * pass);
        // pupil + WFS amplitude of the lane's 2 x 4 pixels (one table: amplitude, or -1 outside the pupil), long before use
        f32x4s apv[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int x = 16 * (2 * cg + tt) + 4 * lq;
            const bool okp = row_ok && x < R;
            apv[tt] = *reinterpret_cast<const f32x4s*>(a.amp_pupil + (okp ? (size_t)y * R + x : 0));
            if (!okp) apv[tt] = f32x4s{-1.f, -1.f, -1.f, -1.f};
        }

        // ---- atmosphere: every layer's tile through LDS (16-byte loads), separable Catmull-Rom ----------------------------
        f32x4s sup[2];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) sup[tt] = f32x4s{0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < k.pa.n_layer; ++l) {
            const LayerTaps& tp = step_taps<PE>(k.pa, l, e);
            const float* map = static_cast<const float*>(k.pa.screen[l]) + (size_t)e * S * S;
            const int r0 = y0 + k.pa.foot + tp.dy - 1, c0 = k.pa.foot + tp.dx - 1;
            // Every wave stages ITS tile (16 rows x 32 columns + the 4 x 4 stencil apron) in a private LDS slice: no
            // workgroup barrier in the atmosphere / DM part, so the 16 waves drift apart and one wave's loads overlap
            // another's arithmetic.  LDS operations of one wave execute in order: a compiler fence is all that is needed
            // between the tile writes and the reads of other lanes' data.
            asm volatile("" ::: "memory");
            AO_STAMP(3 + 6 * pass);
            {
                // tile element (r, c) = map[r0 + 16 band + r][c0 + 32 cg + c], staged as rows of WC4 float4
                const int rw0 = r0 + 16 * band, cw0 = c0 + 32 * cg;
                f32x4s v[NV4];
#pragma unroll
                for (int q = 0; q < NV4; ++q) {
                    const int idx = lane + 64 * q;
                    const int r = idx / WC4, c = 4 * (idx - r * WC4);
                    const int rr = rw0 + r, cc = cw0 + c;
                    const bool need = idx < WR * WC4 && 16 * band + r < tye + 3 && 32 * cg + c < R + 3 && rr >= 0 && rr < S && cc >= 0 && cc < S;
                    int pr = rr + tp.oy, pc = cc + tp.ox;         // torus: physical = (logical + origin) mod S
                    pr = pr >= S ? pr - S : pr;
                    pc = pc >= S ? pc - S : pc;
                    const bool ok = need && cc + 3 < S && pc + 3 < S;      // the 4 columns are contiguous in memory
                    const float* src = map + (ok ? (size_t)pr * S + pc : 0);
                    f32x4s t;                                     // the screen rows are only 4-byte aligned (S = R + 6)
                    __builtin_memcpy(&t, src, 16);
                    v[q] = ok ? t : f32x4s{0.f, 0.f, 0.f, 0.f};
                    if (need && !ok) {
                        // a float4 that straddles the wrap of the torus or the edge of the screen: element by element
#pragma unroll
                        for (int d = 0; d < 4; ++d) {
                            int pd = pc + d;
                            pd = pd >= S ? pd - S : pd;
                            if (cc + d < S) v[q][d] = map[(size_t)pr * S + pd];
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < NV4; ++q) {
                    const int idx = lane + 64 * q;
                    if (idx < WR * WC4) *reinterpret_cast<f32x4s*>(mapt + 4 * idx) = v[q];
                }
            }
            asm volatile("" ::: "memory");
            AO_STAMP(4 + 6 * pass);
            const float wx0 = (float)tp.wx[0], wx1 = (float)tp.wx[1], wx2 = (float)tp.wx[2], wx3 = (float)tp.wx[3];
            const float wy0 = (float)tp.wy[0], wy1 = (float)tp.wy[1], wy2 = (float)tp.wy[2], wy3 = (float)tp.wy[3];
            const float lo = lohi[l][0], hi = lohi[l][1], wl = (float)tp.weight;
            const bool zero_outside = (lo > 0.f || hi < 0.f);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int xt = 16 * tt + 4 * lq;                  // wave-tile column of the first tap of the lane's first pixel
                // the 4 rows x 8 columns of taps (7 used) as 16-byte LDS reads; horizontal pass per row, then vertical
                float h[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float* m = mapt + (lc + q) * WC + xt;
                    const f32x4s m0 = *reinterpret_cast<const f32x4s*>(m), m1 = *reinterpret_cast<const f32x4s*>(m + 4);
                    const float t[7] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[q][r] = ((wx0 * t[r] + wx1 * t[r + 1]) + wx2 * t[r + 2]) + wx3 * t[r + 3];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = ((wy0 * h[0][r] + wy1 * h[1][r]) + wy2 * h[2][r]) + wy3 * h[3][r];
                    // skimage clip=True: clamp to the input range, keep exact zeros when 0 is outside it
                    if (!(zero_outside && v == 0.f)) v = v < lo ? lo : (v > hi ? hi : v);
                    sup[tt][r] += v * wl;
                }
            }
        }
        AO_STAMP(5 + 6 * pass);

        // ---- DM surface on the matrix cores, pupil, phase store, E0 -> LDS, telemetry sums -----------------------------
        const int iy = y / 6, by = y - 6 * iy;
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int x = 16 * (2 * cg + tt) + 4 * lq;
            f32x4s dmv = {0.f, 0.f, 0.f, 0.f};
            const float* bp = s1 + (size_t)(row_ok ? y : 0) * SS + lq;   // B[k = lane >> 4][j = lane & 15 -> row y]   (LDS)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)                              // A[i = lane & 15 -> column][k = lane >> 4] = breg
                if (4 * ks < nAp) dmv = __builtin_amdgcn_mfma_f32_16x16x4f32(breg[tt][ks], bp[4 * ks], dmv, 0, 0, 0);
            if (row_ok && x < R) {
                const size_t q = (size_t)y * R + x;
                f32x4s atm, phi;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    atm[r] = sup[tt][r] * k.atm_scale;
                    const bool in = apv[tt][r] >= 0.f;
                    const float res = in ? (atm[r] + dmv[r]) : 0.f;
                    phi[r] = res * k.src_scale;
                    if (in) {
                        const double da = (double)atm[r], dr = (double)res;
                        s_atm += da;
                        q_atm += da * da;
                        s_res += dr;
                        q_res += dr * dr;
                    }
                }
                if (k.pa.store_atm) *reinterpret_cast<f32x4s*>(k.pb.opd_atm + pix0 + q) = atm;
                *reinterpret_cast<f32x4s*>(k.pb.phase + pix0 + q) = phi;
                // lenslet (i, j) element E[a][b] = phase[i p + b][j p + a]: the reference tiles phase.T
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jx = (x + r) / 6, ax = (x + r) - 6 * jx;
                    const int slot = slot_s[iy * n_sub + jx];
                    if (slot >= 0) {                                      // every pixel of a valid lenslet, lit or not
                        const float am = apv[tt][r] >= 0.f ? apv[tt][r] * (1.f / 12.f) : 0.f;   // 1/n of |FFT2(E)/n|^2 (ShackHartmann.py:539)
                        float sn = 0.f, cs = 0.f;
                        if (am != 0.f) {
                            if (a.sc.fast_trig) sincos_fast(phi[r], &sn, &cs); else sincosf(phi[r], &sn, &cs);
                        }
                        E0[slot * EST + ax * 6 + by] = {am * cs, am * sn};
                    }
                }
            }
        }
    }
    AO_STAMP(13);
    AO_WSTAMP(3);
    // telemetry: waves in a fixed order
    s_atm = wave_sum_f64(s_atm);
    q_atm = wave_sum_f64(q_atm);
    s_res = wave_sum_f64(s_res);
    q_res = wave_sum_f64(q_res);
    if (lane == 0) {
        red[0][w] = s_atm;
        red[1][w] = q_atm;
        red[2][w] = s_res;
        red[3][w] = q_res;
    }
    lds_barrier();                                             // E0 complete, red complete, cimg / s1 / layer tiles dead
    AO_STAMP(14);
    AO_WSTAMP(4);
    // the camera's tables on their way into LDS (no registers, nobody waits): they land while the spots are computed
    const bool cam_photons = a.det.active && a.det.photon_noise;
    if (cam_photons) alias_table_to_lds(a.pa.tab, a.pa.words, reinterpret_cast<uint32_t*>(lds + L.tab), w, 16, lane);
    if (tid == 1023) {                                           // an idle lane: runs beside the spots of the other waves
        double v[4];
        for (int c = 0; c < 4; ++c) {
            double t = 0;
            for (int q = 0; q < 16; ++q) t += red[c][q];
            v[c] = t;
        }
        double* pp = k.pb.part + (size_t)e * 4;                  // kept for state inspection (FinishArgs.n_tiles == 1)
        for (int c = 0; c < 4; ++c) pp[c] = v[c];
        telemetry_scalars<float>(a.fa, e, a.n_env, v);
    }

    // ---- stage B: spots, frame, threshold, centre of gravity ---------------------------------------------------------------
    const int sl_i = lane / 3, q3 = lane - 3 * sl_i;
    const int s = 21 * w + sl_i;
    const bool ok = lane < 63 && s < n_valid;
    float Ia[6], Ib[6];
#pragma unroll
    for (int u = 0; u < 6; ++u) Ia[u] = Ib[u] = 0.f;
    float mx = 0.f;
    int li = 0, lj = 0;
    if (21 * w < n_valid) {                                      // wave-uniform: the twiddle shuffles need every lane
        fast6::lenslet_spots<float, 2, true>(E0 + (ok ? s : 0) * EST, q3, ctab, Ia, Ib);
        if (!ok) {
#pragma unroll
            for (int u = 0; u < 6; ++u) Ia[u] = Ib[u] = 0.f;
        }
    }
    if (ok) {
        const int kk = a.sc.subap_idx[s];
        li = kk / n_sub;
        lj = kk - li * n_sub;
    }
    AO_STAMP(22);
    if (a.det.active) {
        // ---- self*self.cam: the camera on the lane's 12 pixels (detector.hpp, "Stream layout") --------------------------------
        // (camera_sh6.hpp: shared with the stand-alone camera kernel)
        f32x16s pxv = camera_pack(Ia, Ib);
        AO_STAMP(23);
        if (cam_photons) {                                        // this wave's share of the tables has landed; then everybody's
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            AO_STAMP(6);
            lds_barrier();
        }
        AO_STAMP(8);
        camera_sh6_lane(pxv, ok, (uint32_t)((li * 6) * R + lj * 6 + q3), R, (uint32_t)e, a.det, tab_s, a.pa.lmax);
        camera_unpack(pxv, Ia, Ib);
    }
    if (ok) {
        float* fr = a.frame + pix0 + (size_t)(li * 6) * R + lj * 6 + q3;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            fr[(size_t)u * R] = Ia[u];
            fr[(size_t)u * R + 3] = Ib[u];
            mx = Ia[u] > mx ? Ia[u] : mx;
            mx = Ib[u] > mx ? Ib[u] : mx;
        }
    }
    AO_STAMP(15);
    if (a.det.active && (a.det.dark_e > 0.f || a.det.readout_noise != 0.f)) {
        // the camera also reads out the pixels of the lenslets that are not valid (no light): dark + read-out noise, ADC
        const float rtab = recip_table_lane();
        for (int idx = tid; idx < n_sub * n_sub * 9; idx += 1024) {
            const int kk = idx / 9, j = idx - 9 * kk;
            if (slot_s[kk] < 0) {
                const int i2 = kk / n_sub, j2 = kk - i2 * n_sub;
                uint32_t pix[4];
                sh6_quad_pixels(j, i2 * 6, j2 * 6, R, pix);
                f32x4d v = {0.f, 0.f, 0.f, 0.f};
                detector_quad<false>(v, pix, pix[0], (uint32_t)e, a.det, rtab, a.pa);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) a.frame[pix0 + pix[s4]] = v[s4];
            }
        }
    }
    // Stage C operands that do not depend on the slopes are requested now: the rows of M (16 lanes per mode, 16-byte
    // loads) and the reference slopes; their latency overlaps the threshold barrier and the centre of gravity.
    // (an opaque zero in the address pins these loads here: loads from read-only kernel arguments may otherwise be
    //  hoisted to the top of the kernel, where 40 more live registers spill)
    int late0;
    asm volatile("v_mov_b32 %0, 0" : "=v"(late0) : "v"(mx));
    const int n_sig = 2 * n_valid, n4 = n_sig / 4, A = k.n_valid_act;
    const int km = tid >> 4, l16 = tid & 15;                     // (host: n_modes <= 52, nSig % 4 == 0, nSig <= 640)
    f32x4s mv[10];
    // (a wave without a valid mode requests nothing: a load of a dummy address still takes its turn in the wave-instruction queue of the
    //  vector-memory path, in front of the loads somebody waits for -- the 160 such loads of M and M2C^T cost the step 1.3 us)
    if (4 * w < a.n_modes) {                                     // (wave-uniform: a wave holds the modes 4 w .. 4 w + 3)
        const f32x4s* row = reinterpret_cast<const f32x4s*>(a.fac_m + (size_t)(km < a.n_modes ? km : 0) * n_sig);
#pragma unroll
        for (int j = 0; j < 10; ++j) mv[j] = row[(l16 + 16 * j < n4 ? l16 + 16 * j : 0) + late0];
    } else {
#pragma unroll
        for (int j = 0; j < 10; ++j) mv[j] = f32x4s{0.f, 0.f, 0.f, 0.f};
    }
    const float ref0 = a.sc.ref[ok ? s : 0], ref1 = a.sc.ref[ok ? n_valid + s : 0];
    // the integrator's inputs of this lane's actuator -- the previous observation (or the caller's action) and env.dm_prev
    // (OOPAOEnv.py:508) -- are requested here too: held from the prologue on they were two more live registers across stages A and B
    float act_prev = 0.f, dm_prev_c = 0.f;
    if (a.fa.do_integrate) {
        const size_t io = (size_t)e * (nA * nA) + (has_act ? act_px : 0) + late0;
        act_prev = (a.fa.gain_from_obs != 0.f) ? a.fa.obs[io] : a.fa.action[io];
        dm_prev_c = a.fa.dm_prev[(size_t)e * k.n_valid_act + (has_act ? tid : 0) + late0];
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) red_mx[w] = mx;
    AO_WSTAMP(5);
    lds_barrier();
    AO_STAMP(16);
    mx = red_mx[0];
#pragma unroll
    for (int q = 1; q < 16; ++q) mx = red_mx[q] > mx ? red_mx[q] : mx;
    if (tid == 0) a.wfs_max[e] = mx;
    // (aliases E0: every wave has its spots by now) zero at non-actuators: vec_to_img.  Stage C writes the actuators' values two
    // barriers from here.
    for (int i = tid; i < nA * nA; i += 1024) img_s[i] = 0.f;
    const float cut = a.sc.threshold * mx;
    {
        float norm = 0.f, m0 = 0.f, m1 = 0.f;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            const float xa = Ia[u] < cut ? 0.f : Ia[u], xb = Ib[u] < cut ? 0.f : Ib[u];
            const float rs = xa + xb;
            norm += rs;
            m0 += rs * (float)u;
            m1 += xa * (float)q3 + xb * (float)(q3 + 3);
        }
        norm += __shfl_down(norm, 1) + __shfl_down(norm, 2);
        m0 += __shfl_down(m0, 1) + __shfl_down(m0, 2);
        m1 += __shfl_down(m1, 1) + __shfl_down(m1, 2);
        if (ok && q3 == 0) {
            float c0 = 0.f, c1 = 0.f;
            if (norm != 0.f) {
                c0 = m0 / norm;
                c1 = m1 / norm;
            }
            const float v0 = (c0 - ref0) / a.sc.units, v1 = (c1 - ref1) / a.sc.units;
            sl[s] = v0;
            sl[n_valid + s] = v1;
            a.signal[(size_t)e * 2 * n_valid + s] = v0;
            a.signal[(size_t)e * 2 * n_valid + n_valid + s] = v1;
        }
    }
    lds_barrier();
    AO_STAMP(17);

    // ---- stage C ------------------------------------------------------------------------------------------------------------
    // Same arithmetic as tail_from_slopes (sh_device.hpp), with every operand that does not depend on the slopes already
    // in registers: t = M s ; o = -M2C t 1e6 ; integrator ; obs image ; reward          (MAIN/OOPAOEnv/OOPAOEnv.py:491-536)
    const int img = nA * nA;
    float* tm = img_s + img;                                     // [K] modal coefficients
    {
        const f32x4s* s4 = reinterpret_cast<const f32x4s*>(sl);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < 10; ++j) {
            const bool okj = l16 + 16 * j < n4;
            const f32x4s sv = s4[okj ? l16 + 16 * j : 0];
            const float d = ((mv[j][0] * sv[0] + mv[j][1] * sv[1]) + mv[j][2] * sv[2]) + mv[j][3] * sv[3];
            acc += okj ? d : 0.f;
        }
#pragma unroll
        for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
        if (km < a.n_modes && l16 == 0) tm[km] = acc;
    }
    // M2C^T rows for the lane's group of 4 actuators (4 lanes per group, lane part p takes the modes p, p + 4, ...):
    // requested before the barrier, they do not depend on t
    const int n_grp = (A + 3) / 4, g = tid >> 2, part = tid & 3;
    const bool g_ok = g < n_grp;
    f32x4s cv[13];
    int late1;
    asm volatile("v_mov_b32 %0, 0" : "=v"(late1) : "v"(late0));
    if (16 * w < n_grp) {                                        // (wave-uniform: a wave holds the groups 16 w .. 16 w + 15)
#pragma unroll
        for (int j = 0; j < 13; ++j)
            __builtin_memcpy(&cv[j], a.fac_m2c_t + (part + 4 * j < a.n_modes ? (size_t)(part + 4 * j) * A + 4 * (g_ok ? g : 0) : 0) + late1, 16);
    } else {
#pragma unroll
        for (int j = 0; j < 13; ++j) cv[j] = f32x4s{0.f, 0.f, 0.f, 0.f};
    }
    lds_barrier();
    AO_STAMP(19);
    double ss = 0.0;
    {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 13; ++j) {
            const bool okj = part + 4 * j < a.n_modes;
            const float t = tm[okj ? part + 4 * j : 0];
#pragma unroll
            for (int d = 0; d < 4; ++d) acc[d] += okj ? cv[j][d] * t : 0.f;
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            acc[d] += __shfl_xor(acc[d], 1, 4);
            acc[d] += __shfl_xor(acc[d], 2, 4);
        }
        // lane tid finishes actuator k = tid (= 4 g + part): its command, image index and integrator input are in registers
        if (has_act) {
            const float mine = part == 0 ? acc[0] : (part == 1 ? acc[1] : (part == 2 ? acc[2] : acc[3]));
            if (a.fa.do_integrate) {
                const float act = (a.fa.gain_from_obs != 0.f) ? a.fa.gain_from_obs * act_prev : act_prev;
                const float cn = dm_prev_c * a.fa.leak + act * 1e-6f;                    // float32 increment (k_recon_finish)
                a.fa.coefs[(size_t)e * A + tid] = cn;
                a.fa.dm_prev[(size_t)e * A + tid] = cn;
            }
            const float o = -mine * 1e6f;
            img_s[act_px] = o;
            ss += (double)o * (double)o;
        }
    }
    lds_barrier();
    AO_STAMP(20);
    {
        float* ob = a.fa.obs + (size_t)e * img;
        for (int q = tid; q < img; q += 1024) ob[q] = img_s[q];
    }
    ss = wave_sum_f64(ss);
    if (lane == 0) red_tail[w] = ss;
    lds_barrier();
    AO_STAMP(21);
    if (tid == 0) {
        double tot = 0;
        for (int q = 0; q < 16; ++q) tot += red_tail[q];
        if (a.fa.reward) a.fa.reward[e] = (float)(-sqrt(tot));
        if (a.fa.ret && a.fa.do_integrate) a.fa.ret[e] += (float)(-sqrt(tot));
    }
    AO_STAMP(18);
    AO_WSTAMP(6);
}

#ifdef AO_STEP_STAMPS
extern "C" int aoenv_debug_wstamps(unsigned long long* h_out) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_wstamps), sizeof(unsigned long long) * 256 * 16 * 8) == hipSuccess ? 0 : 1;
}
extern "C" int aoenv_debug_stamps(unsigned long long* h_out, int n_env) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 32 * (size_t)n_env) == hipSuccess ? 0 : 1;
}
#endif

int step_fused_supported(int R, int n_subap, int n_valid, int n_act, int n_modes) {
    if (R % n_subap || R / n_subap != fast6::P) return 0;
    if (R > fstep::TX || R % 4) return 0;
    if (n_valid > 16 * 21 || n_subap * n_subap > 32767 || n_act > 32) return 0;
    if (n_modes < 1 || n_modes > 52 || (2 * n_valid) % 4 != 0 || 2 * n_valid > 640) return 0;   // stage C register blocking
    const StepLds L = step_lds_layout(n_act, n_subap, n_valid, n_modes);
    return (size_t)L.total * 4 <= 160 * 1024 - 1024 ? 1 : 0;     // static __shared__ of the kernel: < 1 KB
}

// words of LDS the fused kernel has for the camera's alias tables (where stage A's buffers were): the env keeps the part of the
// tables that fits there for ALL its camera kernels, so that a pixel is drawn the same way whichever kernel meets it
int step_alias_capacity(int n_act) { return step_lds_layout(n_act, 1, 1, 1).tab_cap; }

int launch_env_step(const StepArgs& a, hipStream_t st) {
    const StepLds L = step_lds_layout(a.k.n_act, a.n_subap, a.n_valid, a.n_modes);
    if (a.det.active && a.det.photon_noise && (!a.pa.tab || a.pa.words > L.tab_cap || a.pa.lmax < palias::kCoarseStep))
        return fail("fused step: the photon-noise tables are missing or do not fit (%d words, room for %d)", a.pa.words, L.tab_cap);
    const size_t lds = (size_t)L.total * 4;
    const int v = a.k.n_act <= 24 ? 0 : 1;
    const bool pe = a.k.pa.env_taps != nullptr;
    const void* fn = v == 0 ? (pe ? reinterpret_cast<const void*>(k_env_step_sh6<6, true>) : reinterpret_cast<const void*>(k_env_step_sh6<6, false>))
                            : (pe ? reinterpret_cast<const void*>(k_env_step_sh6<8, true>) : reinterpret_cast<const void*>(k_env_step_sh6<8, false>));
    static size_t attr_set[4] = {0, 0, 0, 0};
    const int vi = 2 * v + (pe ? 1 : 0);
    if (lds > 64 * 1024 && lds > attr_set[vi]) {
        AO_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[vi] = lds;
    }
    if (v == 0 && !pe) hipLaunchKernelGGL((k_env_step_sh6<6, false>), dim3(a.n_env), dim3(1024), lds, st, a, L);
    else if (v == 0) hipLaunchKernelGGL((k_env_step_sh6<6, true>), dim3(a.n_env), dim3(1024), lds, st, a, L);
    else if (!pe) hipLaunchKernelGGL((k_env_step_sh6<8, false>), dim3(a.n_env), dim3(1024), lds, st, a, L);
    else hipLaunchKernelGGL((k_env_step_sh6<8, true>), dim3(a.n_env), dim3(1024), lds, st, a, L);
    AO_HIP(hipGetLastError());
    return 0;
}

}  // namespace ao
