// Host side of libaoenv: the AoEnv object, the atmosphere clock and the C ABI of include/aoenv.h.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>

#include "common.hpp"
#include "detector.hpp"
#include "sh_device.hpp"

namespace ao {

static thread_local std::string g_err;

int fail(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return 1;
}

struct LayerClock {
    double ratio[2] = {0, 0};   // pixels per frame (x, y)          OOPAO/Atmosphere.py:362-363
    double buff[2] = {0, 0};    // sub-pixel accumulator            OOPAO/Atmosphere.py:392-404
};

}  // namespace ao

using namespace ao;

struct AoEnv {
    AoCfg c{};
    int device = 0;
    size_t esz = 4;
    int R = 0, N = 0, S = 0, A = 0, nAct = 0, E = 0, L = 0, nin = 0, nout = 0, K = 0, nSig = 0, nSub = 0, nVal = 0;
    int p = 0, n = 0, n_pupil = 0;
    // per-layer screen grids (fov != 0: a layer at altitude h lives on a grid of its own, OOPAO/Atmosphere.py:216-218); N, S, nin,
    // nout, K above are layer 0's (every layer's when `uniform`: the only case the fused step kernel takes)
    int Nl[kMaxLayer] = {0}, Sl[kMaxLayer] = {0}, ninl[kMaxLayer] = {0}, noutl[kMaxLayer] = {0}, Kl[kMaxLayer] = {0};
    size_t scr_off[kMaxLayer + 1] = {0};     // element offset of layer l's [E][S_l^2] block in `screen`
    int Kmax = 0, noutmax = 0, Smax = 0;
    bool uniform = true;
    void* ab_l[kMaxLayer] = {nullptr};       // [nout_l][K_l] ring operators [A | B] of layer l
    int* inner_idx_l[kMaxLayer] = {nullptr};
    int* outer_idx_l[kMaxLayer] = {nullptr};
    bool have_ab[kMaxLayer] = {false}, have_in[kMaxLayer] = {false}, have_out[kMaxLayer] = {false};
    int last_zx_layer = 0;
    LayerClock clk[kMaxLayer];
    int org[kMaxLayer][2] = {{0, 0}};        // torus origin (oy, ox) of every layer: logical (r, c) at ((r + oy) % S, (c + ox) % S)
    int ring_pending[kMaxLayer] = {0};       // > 0: split count of a ring extrusion whose scatter the next fused step kernel will do
    // per-env clocks (aoenv_set_wind_env): every env its own wind vector per layer; clocks, origins and taps live on the device
    bool per_env_wind = false;
    EnvClock* env_clk[2] = {nullptr, nullptr};   // [L][E] each: current / next (k_ring_prepare_env reads one, writes the other)
    int clk_cur = 0;
    LayerTaps* env_taps = nullptr;           // [L][E] taps of the current step
    bool use_coefs_img = false;              // aoenv_set_option(AOENV_OPT_COEFS_IMAGE); always on above 1024 actuators
    bool defer_ring = true;                  // aoenv_set_option(AOENV_OPT_DEFER_RING)
    bool minmax_dirty[kMaxLayer] = {false};  // the layer's min / max table is stale (ring extruded without the min / max pass)
    bool have[AOENV_C_COUNT] = {false};
    double units = 1.0;
    // device memory (element type = dtype unless noted)
    void* screen[1] = {nullptr};            // [L][E][S*S], every screen a torus (see atm_kernels.hip)
    void* minmax = nullptr;                 // [L][E][2]
    uint32_t* mt_state = nullptr;           // [2][L][E][624]: per layer one committed copy and one the ring look-ahead writes
    int* mt_pos = nullptr;                  // [2][L][E]
    uint32_t* mt_cur[kMaxLayer] = {nullptr};   // committed MT19937 state of every layer's ring stream ...
    int* pos_cur[kMaxLayer] = {nullptr};
    uint32_t* mt_alt[kMaxLayer] = {nullptr};   // ... and the other copy (swapped in when a look-ahead is consumed)
    int* pos_alt[kMaxLayer] = {nullptr};
    // Ring pipeline (float32 fused path, shared clock): the operand [Z | xi] of a layer's NEXT pixel crossing is put together while
    // the current crossing is being served -- xi by extra workgroups of the ring GEMM's launch (the stream position only moves at
    // crossings), Z by the fused step kernel right after it has written the ring (the screen does not change until the next
    // crossing) -- so a crossing step launches the GEMM and nothing else in front of the step kernel: k_ring_prepare (9 us, the
    // Gaussian draw) leaves the critical path.  [An earlier form ran prepare + GEMM one crossing ahead on a second stream: with one
    // 1024-lane workgroup resident on every CU the side-stream kernels found no free CU and time-sliced with the step kernel: slower.]
    struct RingAhead { bool valid = false; int sx = 0, sy = 0, buf = 0; };   // zx_pipe[buf][l] holds [Z | xi] for a crossing in direction (sx, sy); mt_alt[l] the stream after it
    RingAhead ahead[kMaxLayer];
    bool gather_next[kMaxLayer] = {false};  // the next fused step kernel is to gather Z into zx_pipe[ahead.buf][l]
    void* zx_pipe = nullptr;                // [2][L][E][K]
    const void* last_zx = nullptr;          // operand of the last ring GEMM (AOENV_B_XI)
    const void* ring_src[kMaxLayer] = {nullptr};   // slabs of the pending (deferred) ring of every layer
    bool use_lookahead = true;              // aoenv_set_option(AOENV_OPT_RING_LOOKAHEAD): the ring pipeline
    void* zx = nullptr;                     // [E][K]  [Z | xi]
    void* xbuf = nullptr;                   // [splits][E][nout] split-K slabs of the ring GEMM
    void* ab = nullptr;                     // = ab_l[0] (fused path)
    int* inner_idx = nullptr;               // = inner_idx_l[0]
    int* outer_idx = nullptr;
    double layer_weight[kMaxLayer] = {0};
    void* gx = nullptr;
    void* gy = nullptr;
    void* gxt = nullptr;                    // [nActPad4][Rpad128] transpose of gx, zero padded
    void* modes = nullptr;                  // [R*R][A]
    void* dm_opd = nullptr;                 // [E][R*R] dense path
    int* act_idx = nullptr;
    uint8_t* pupil = nullptr;
    void* opd_atm = nullptr;
    void* coefs = nullptr;
    void* dm_prev = nullptr;                // [E][A] env.dm_prev, the integrator state (not touched by aoenv_set_coefs)
    float* dm_rows = nullptr;               // [E][Rpad128][4][ga_stride] Gy C per env (float32; k_dm_rows), the same switch
    void* coefs_img = nullptr;              // [E][nAct^2] command images for the phase kernels of large DMs (A > 1024)
    void* phase = nullptr;
    void* scal = nullptr;                   // [E][4]
    double* part = nullptr;                 // [E][tiles][4] telemetry partial sums of the phase kernel
    int n_tiles = 0;
    void* total = nullptr;
    void* residual = nullptr;
    void* wfs_max = nullptr;
    void* amp = nullptr;
    int* subap_idx = nullptr;
    uint8_t* valid2d = nullptr;             // [nSub*nSub]
    short* slot_of = nullptr;               // [nSub*nSub] lenslet -> compact valid index, -1 = not valid
    float* amp_pupil = nullptr;             // [R*R] fused step kernel: amplitude inside the pupil, -1 outside
    float* gxa = nullptr;                   // gx / gy re-laid out as MFMA operand tables (ga_index(), common.hpp), zero padded to Rpad128
    float* gya = nullptr;
    int ga_stride = 8;                      // k steps per (row, q), a multiple of 4 (16-byte loads), >= ceil(nAct / 4)
    std::vector<uint8_t> h_pupil;           // host copies, to rebuild amp_pupil
    std::vector<double> h_amp;
    bool amp_pupil_dirty = true;
    void* sh_ref = nullptr;
    void* tw = nullptr;
    void* phs = nullptr;
    void* frame = nullptr;
    void* signal = nullptr;
    void* recon = nullptr;                  // [A][nSig]
    void* fac_m = nullptr;                  // [K][nSig] modal command matrix calib.M          (AOENV_C_RECON_FACTORS)
    void* fac_m2c_t = nullptr;              // [K][A]    transposed M2C
    void* fac_m2c = nullptr;                // [A][Kp]   M2C, rows zero padded to Kp = n_modes rounded up to 4 (16-byte rows)
    void* tbuf = nullptr;                   // [splits][E][Kp] modal coefficients t = M s of the factored reconstruction
    size_t fac_cap = 0;                     // modes the two buffers above are sized for
    bool use_factored_recon = true;         // aoenv_set_option(AOENV_OPT_FACTORED_RECON)
    int n_modes = 0;
    // Pyramid
    void* pyr_mask = nullptr;               // [N*N][2]
    void* pyr_tt = nullptr;                 // [nTheta][R*R]
    void* pyr_tw = nullptr;                 // [N][2]
    void* pyr_t1 = nullptr;                 // [E][chunk][R][N] complex
    void* pyr_t2 = nullptr;                 // [E][chunk][N][N] complex
    FftPlan pyr_plan{};
    int pyr_chunk = 1;
    void* vbuf = nullptr;                   // [E][A]
    DetectorCfg det{};                      // aoenv_set_detector(); det.active = 0: ideal camera
    uint32_t* alias_tab = nullptr;          // alias tables of the photon-noise sampler (poisson_alias.hpp), whole
    int alias_words = 0;                    // the part of them this env's kernels use, and the photon count it reaches: geometries
    float alias_lmax = 0.f;                 // the fused step kernel can take keep what fits in ITS LDS, for every camera kernel
    PoissonAlias alias() const { return PoissonAlias{alias_tab, alias_words, alias_lmax}; }
    bool det_seeded = false;                // a seed has been set: the frame counter survives later aoenv_set_detector calls
    void* ret_acc = nullptr;                // caller-owned [E] episode-return accumulator (aoenv_set_return_accumulator)
    std::vector<void*> allocs;
    // optional per-kernel timing with HIP events on the launch stream (bench.py roofline leg)
    bool use_fast_wfs = true;               // aoenv_set_option(AOENV_OPT_FAST_WFS)
    bool use_mfma = true;                   // aoenv_set_option(AOENV_OPT_MFMA_GEMM)
    bool use_fast_trig = true;              // aoenv_set_option(AOENV_OPT_FAST_TRIG)
    bool use_fused_tail = true;             // aoenv_set_option(AOENV_OPT_FUSED_TAIL); needs AOENV_C_RECON_FACTORS
    bool use_fused_step = true;             // aoenv_set_option(AOENV_OPT_FUSED_STEP): the whole step in one kernel
    bool store_opd_atm = false;             // aoenv_set_option(AOENV_OPT_STORE_ATM_OPD): write atm.OPD every step
    bool atm_user_defined = false;          // aoenv_set_atm_opd() until the next step / new screens
    int debug_ablate = 0;                   // aoenv_set_option(99): skip kernel sections (timing diagnosis only)
    bool prof_on = false;
    struct ProfEv { int stage; hipEvent_t a, b; };
    std::vector<ProfEv> prof_ev;

    template <typename T> T* as(void* p_) const { return static_cast<T*>(p_); }
    void* screen_ptr(int which, int l) const {
        return static_cast<char*>(screen[which]) + scr_off[l] * esz;
    }
    void* minmax_ptr(int l) const { return static_cast<char*>(minmax) + (size_t)l * E * 2 * esz; }
    void* xbuf_ptr(int l) const { return static_cast<char*>(xbuf) + (size_t)l * kMaxSplits * E * noutmax * esz; }
    void* zx_pipe_ptr(int buf, int l) const { return static_cast<char*>(zx_pipe) + ((size_t)buf * L + l) * E * Kmax * esz; }
};

namespace {

int dmalloc(AoEnv* env, void** p, size_t bytes, bool zero = true);
// per-env DM command products shared by the tiles of the phase kernels (AOENV_OPT_COEFS_IMAGE): float32 shards keep Gy C in
// MFMA operand layout (k_dm_rows; its padding stays zero from here on), float64 shards the scattered command image
static int alloc_dm_rows(AoEnv* env) {
    if (env->esz == 4 && env->nAct <= 128) {
        if (!env->dm_rows)
            AO_TRY(dmalloc(env, (void**)&env->dm_rows, (size_t)env->E * (cdiv(env->R, 128) * 128) * 4 * env->ga_stride * sizeof(float)));
    } else if (!env->coefs_img) {
        AO_TRY(dmalloc(env, &env->coefs_img, (size_t)env->E * env->nAct * env->nAct * env->esz));
    }
    return 0;
}
int dmalloc(AoEnv* env, void** p, size_t bytes, bool zero) {
    if (bytes == 0) bytes = 16;
    AO_HIP(hipMalloc(p, bytes));
    env->allocs.push_back(*p);
    if (zero) AO_HIP(hipMemset(*p, 0, bytes));
    return 0;
}

// Wraps one kernel launch in a hipEvent pair when profiling is enabled (events are recorded on the same
// stream as the kernel, so the elapsed time is that kernel's device time plus its launch gap).
struct ProfScope {
    AoEnv* env; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; int stage;
    ProfScope(AoEnv* e, int stage_, hipStream_t s) : env(e), st(s), stage(stage_) {
        if (!env->prof_on) return;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) { a = b = nullptr; return; }
        (void)hipEventRecord(a, st);
    }
    ~ProfScope() {
        if (!a) return;
        (void)hipEventRecord(b, st);
        env->prof_ev.push_back({stage, a, b});
    }
};
#define AO_PROF(env, stage, st) ProfScope prof_scope_##stage(env, AOENV_K_##stage, st)

template <typename T>
int upload_f64(void* dst, const double* src, size_t n) {
    std::vector<T> tmp(n);
    for (size_t i = 0; i < n; ++i) tmp[i] = (T)src[i];
    AO_HIP(hipMemcpy(dst, tmp.data(), n * sizeof(T), hipMemcpyHostToDevice));
    return 0;
}

int upload_real(AoEnv* env, void* dst, const double* src, size_t n) {
    return env->c.dtype == AOENV_F32 ? upload_f64<float>(dst, src, n) : upload_f64<double>(dst, src, n);
}

double sgn(double v) { return (v > 0) - (v < 0); }

// Catmull-Rom tap weights of skimage's cubic_interpolation() for fractional offset x in [0, 1)
void mt_seed(uint32_t seed, uint32_t* key) {          // numpy legacy mt19937_seed / init_genrand
    for (int pos = 0; pos < kMtN; ++pos) {
        key[pos] = seed;
        seed = 1812433253u * (seed ^ (seed >> 30)) + (uint32_t)pos + 1u;
    }
}

template <typename T>
ShConst<T> sh_const(const AoEnv* env) {
    ShConst<T> sc;
    sc.amp = env->as<T>(env->amp);
    sc.subap_idx = env->subap_idx;
    sc.valid2d = env->use_fast_wfs ? env->valid2d : nullptr;
    sc.ref = env->as<T>(env->sh_ref);
    sc.tw = env->as<T>(env->tw);
    sc.ph = env->as<T>(env->phs);
    sc.units = (T)env->units;
    sc.threshold = (T)env->c.threshold_cog;
    sc.fast_trig = (env->use_fast_trig && sizeof(T) == 4) ? 1 : 0;
    return sc;
}

// C[M][N] (+ split-K slabs) = X . W^T: float32 shards use the MFMA kernel, float64 the generic one.
template <typename T>
int gemm_dispatch(AoEnv* env, const T* X, const T* W, T* C, int M, int N, int K, int* splits, hipStream_t st);
template <>
int gemm_dispatch<float>(AoEnv* env, const float* X, const float* W, float* C, int M, int N, int K, int* splits,
                         hipStream_t st) {
    if (!env->use_mfma) { *splits = 1; return launch_gemm_nt<float>(X, W, C, M, N, K, K, K, N, st); }
    *splits = gemm_splits(M, N, K);
    return launch_gemm_nt_mfma(X, W, C, M, N, K, K, K, *splits, st);
}
template <>
int gemm_dispatch<double>(AoEnv*, const double* X, const double* W, double* C, int M, int N, int K, int* splits,
                          hipStream_t st) {
    *splits = 1;
    return launch_gemm_nt<double>(X, W, C, M, N, K, K, K, N, st);
}

// The operand prepared for a layer's next crossing will not be used (the screens, the stream or the wind changed): nothing of it
// was committed -- the stream copy it advanced is the alternate one -- so it is simply forgotten; the crossing draws in place.
int drop_lookahead(AoEnv* env, int l, hipStream_t) {
    env->ahead[l].valid = false;
    env->gather_next[l] = false;
    return 0;
}
int drop_lookaheads(AoEnv* env, hipStream_t st) {
    for (int l = 0; l < env->L; ++l) AO_TRY(drop_lookahead(env, l, st));
    return 0;
}
int sync_lookaheads(AoEnv* env) { return drop_lookaheads(env, nullptr); }

// ---- add_row on the device (OOPAO/Atmosphere.py:301-311) for every env of the shard ---------------
// lean = true: the ring is scattered without the min / max pass; the fused step kernel recomputes the range from the map
template <typename T>
int flush_ring(AoEnv* env, int l, hipStream_t st) {
    if (!env->ring_pending[l]) return 0;
    if (env->gather_next[l]) AO_TRY(drop_lookahead(env, l, st));   // (the step kernel that would have gathered the next Z is not coming)
    AO_PROF(env, SCATTER, st);
    // (per-env clocks: only the envs that crossed, each through its own origin, and their range right away -- there is no
    //  per-env "dirty" flag on the host)
    const LayerTaps* et = env->per_env_wind ? env->env_taps + (size_t)l * env->E : nullptr;
    AO_TRY(launch_scatter_minmax<T>(env->as<T>(env->screen_ptr(0, l)), static_cast<const T*>(env->ring_src[l]), env->outer_idx_l[l],
                                    env->as<T>(env->minmax_ptr(l)), env->E, env->Sl[l], env->noutl[l], env->ring_pending[l],
                                    env->org[l][0], env->org[l][1], et ? 1 : 0, st, et));
    env->ring_pending[l] = 0;
    return 0;
}
template <typename T>
int flush_rings(AoEnv* env, hipStream_t st) {
    for (int l = 0; l < env->L; ++l) AO_TRY(flush_ring<T>(env, l, st));
    return 0;
}


// lean: no min / max pass (the fused step kernel recomputes the range from the map);  defer: not even the scatter -- the
// fused step kernel of this step writes the ring itself (one launch less per crossing)
template <typename T>
int extrude(AoEnv* env, int l, int sx, int sy, bool lean, hipStream_t st, bool defer = false) {
    AO_TRY(drop_lookahead(env, l, st));                            // computed from the screen and the stream as they were
    AO_TRY(flush_ring<T>(env, l, st));                             // an earlier extrusion of this layer in the same step
    T* map = env->as<T>(env->screen_ptr(0, l));
    T* zx = env->as<T>(env->zx);
    const int S = env->Sl[l];
    const int oy = env->org[l][0], ox = env->org[l][1];
    {
        AO_PROF(env, SHIFT_GATHER, st);                           // Z gather + xi draw, one launch
        AO_TRY(launch_ring_prepare<T>(map, zx, env->inner_idx_l[l], env->mt_cur[l], env->pos_cur[l], env->mt_cur[l], env->pos_cur[l], env->E, S,
                                      env->ninl[l], env->noutl[l], env->Kl[l], sx, sy, oy, ox, st));
    }
    int splits = 1;
    {
        AO_PROF(env, GEMM_RING, st);
        AO_TRY(gemm_dispatch<T>(env, zx, env->as<T>(env->ab_l[l]), env->as<T>(env->xbuf_ptr(l)), env->E, env->noutl[l], env->Kl[l], &splits,
                                st));
    }
    // the shift itself: move the origin of the torus
    env->org[l][0] = ((oy - sy) % S + S) % S;
    env->org[l][1] = ((ox - sx) % S + S) % S;
    env->last_zx = zx;
    env->last_zx_layer = l;
    if (defer && lean) {
        env->ring_pending[l] = splits;
        env->ring_src[l] = env->xbuf_ptr(l);
    } else {
        AO_PROF(env, SCATTER, st);
        AO_TRY(launch_scatter_minmax<T>(map, env->as<T>(env->xbuf_ptr(l)), env->outer_idx_l[l], env->as<T>(env->minmax_ptr(l)), env->E, S,
                                        env->noutl[l], splits, env->org[l][0], env->org[l][1], lean ? 0 : 1, st));
    }
    env->minmax_dirty[l] = lean;
    return 0;
}

// make every layer's min / max table current (consumers other than the fused step kernel)
template <typename T>
int refresh_minmax(AoEnv* env, hipStream_t st) {
    AO_TRY(flush_rings<T>(env, st));
    for (int l = 0; l < env->L; ++l)
        if (env->minmax_dirty[l]) {
            AO_TRY(launch_minmax<T>(env->as<T>(env->screen_ptr(0, l)), env->as<T>(env->minmax_ptr(l)), env->E, env->Sl[l], st));
            env->minmax_dirty[l] = false;
        }
    return 0;
}

// ---- atm.update(): host clock of updateLayer (OOPAO/Atmosphere.py:350-407) -----------------------
// The direction of a layer's next sub-pixel crossing, by running its clock forward (same arithmetic as advance_atmosphere).
bool next_crossing(const LayerClock& k0, int* sx, int* sy) {
    if ((int)std::fabs(k0.ratio[0]) > 0 || (int)std::fabs(k0.ratio[1]) > 0) return false;   // whole-pixel shifts every step: no look-ahead
    if (k0.ratio[0] == 0 && k0.ratio[1] == 0) return false;
    double b[2] = {k0.buff[0], k0.buff[1]};
    for (int it = 0; it < 1000000; ++it) {
        for (int d = 0; d < 2; ++d) b[d] += std::fmod(std::fabs(k0.ratio[d]), 1.0) * sgn(k0.ratio[d]);
        if (std::fabs(b[0]) >= 1 || std::fabs(b[1]) >= 1) {
            *sx = std::fabs(b[0]) < 1 ? 0 : (int)sgn(b[0]);
            *sy = std::fabs(b[1]) < 1 ? 0 : (int)sgn(b[1]);
            return true;
        }
        for (int d = 0; d < 2; ++d) b[d] = std::fmod(std::fabs(b[d]), 1.0) * sgn(b[d]);
    }
    return false;
}

// A crossing of the float32 fused path through the ring pipeline (AoEnv::RingAhead): if the operand [Z | xi] of this crossing was
// put together ahead, commit its stream copy and launch the GEMM alone; else prepare it in place as extrude() does.  Either way
// the GEMM's launch also draws the innovations of the NEXT crossing, and the step kernel of this step is asked to gather its Z.
// The ring itself is left to that kernel (deferred scatter).  Bit-identical to extrude(): the same Z, the same xi, the same product.
int extrude_pipelined(AoEnv* env, int l, int sx, int sy, hipStream_t st) {
    AoEnv::RingAhead& ah = env->ahead[l];
    const int S = env->Sl[l], oy = env->org[l][0], ox = env->org[l][1];
    const int cur = ah.buf;
    float* op = static_cast<float*>(env->zx_pipe_ptr(cur, l));
    if (ah.valid && ah.sx == sx && ah.sy == sy) {
        std::swap(env->mt_cur[l], env->mt_alt[l]);                 // the draw made ahead becomes the layer's stream
        std::swap(env->pos_cur[l], env->pos_alt[l]);
    } else {
        AO_PROF(env, SHIFT_GATHER, st);
        AO_TRY(launch_ring_prepare<float>(env->as<float>(env->screen_ptr(0, l)), op, env->inner_idx_l[l], env->mt_cur[l], env->pos_cur[l],
                                          env->mt_cur[l], env->pos_cur[l], env->E, S, env->ninl[l], env->noutl[l], env->Kl[l], sx, sy, oy, ox, st));
    }
    ah.valid = false;
    env->gather_next[l] = false;
    const int splits = gemm_splits(env->E, env->noutl[l], env->Kl[l]);
    MtAhead m{env->mt_cur[l], env->pos_cur[l], env->mt_alt[l], env->pos_alt[l], static_cast<float*>(env->zx_pipe_ptr(1 - cur, l)),
              env->Kl[l], env->ninl[l], env->noutl[l], env->E};
    {
        AO_PROF(env, GEMM_RING, st);
        AO_TRY(launch_ring_gemm_draw_ahead(op, env->as<float>(env->ab_l[l]), static_cast<float*>(env->xbuf_ptr(l)), env->E, env->noutl[l], env->Kl[l],
                                           splits, m, st));
    }
    env->last_zx = op;
    env->last_zx_layer = l;
    env->org[l][0] = ((oy - sy) % S + S) % S;                      // the shift itself: move the origin of the torus
    env->org[l][1] = ((ox - sx) % S + S) % S;
    env->ring_pending[l] = splits;
    env->ring_src[l] = env->xbuf_ptr(l);
    env->minmax_dirty[l] = true;
    int nsx = 0, nsy = 0;
    if (next_crossing(env->clk[l], &nsx, &nsy)) {                   // (the clock has been advanced for this step already)
        ah.valid = true;
        ah.sx = nsx;
        ah.sy = nsy;
        ah.buf = 1 - cur;
        env->gather_next[l] = true;
    }
    return 0;
}

// Per-env clocks: one launch per layer advances every env's clock on the device and prepares [Z | xi] of the envs that cross a
// pixel; the ring GEMM runs over the whole shard (rows of the other envs are computed and never used: which envs cross is
// not known on the host, and with independent winds some env crosses on nearly every step anyway).
template <typename T>
int advance_atmosphere_env(AoEnv* env, bool lean, hipStream_t st) {
    for (int l = 0; l < env->L; ++l) {
        AO_TRY(flush_ring<T>(env, l, st));
        T* map = env->as<T>(env->screen_ptr(0, l));
        T* zx = env->as<T>(env->zx);
        const size_t row = (size_t)l * env->E;
        {
            AO_PROF(env, SHIFT_GATHER, st);
            AO_TRY(launch_ring_prepare_env<T>(map, zx, env->inner_idx_l[l], env->mt_cur[l], env->pos_cur[l], env->env_clk[env->clk_cur] + row,
                                              env->env_clk[1 - env->clk_cur] + row, env->env_taps + row, env->layer_weight[l], env->E,
                                              env->Sl[l], env->ninl[l], env->noutl[l], env->Kl[l], st));
        }
        int splits = 1;
        {
            AO_PROF(env, GEMM_RING, st);
            AO_TRY(gemm_dispatch<T>(env, zx, env->as<T>(env->ab_l[l]), env->as<T>(env->xbuf_ptr(l)), env->E, env->noutl[l], env->Kl[l], &splits, st));
        }
        env->last_zx = zx;
        env->last_zx_layer = l;
        env->ring_pending[l] = splits;
        env->ring_src[l] = env->xbuf_ptr(l);
        if (!(lean && env->defer_ring)) AO_TRY(flush_ring<T>(env, l, st));
    }
    env->clk_cur = 1 - env->clk_cur;
    return 0;
}

template <typename T>
int advance_atmosphere(AoEnv* env, bool lean, hipStream_t st) {
    if (env->per_env_wind) return advance_atmosphere_env<T>(env, lean, st);
    for (int l = 0; l < env->L; ++l) {
        LayerClock& k = env->clk[l];
        if (k.ratio[0] == 0 && k.ratio[1] == 0) continue;
        const int ns[2] = {(int)std::fabs(k.ratio[0]), (int)std::fabs(k.ratio[1])};
        const int mn = ns[0] < ns[1] ? ns[0] : ns[1], mx = ns[0] > ns[1] ? ns[0] : ns[1];
        const int s0 = (int)sgn(k.ratio[0]), s1 = (int)sgn(k.ratio[1]);
        for (int i = 0; i < mn; ++i) AO_TRY(extrude<T>(env, l, s0, s1, lean, st));
        for (int j = 0; j < mx - mn; ++j)
            AO_TRY(extrude<T>(env, l, ns[0] == mn ? 0 : s0, ns[1] == mn ? 0 : s1, lean, st));
        int b0, b1;
        if (clock_subpixel(k.ratio, k.buff, &b0, &b1)) {          // (the arithmetic the per-env device clocks share, common.hpp)
            if (lean && env->defer_ring && env->use_lookahead && env->zx_pipe && !env->ring_pending[l]) {
                AO_TRY(extrude_pipelined(env, l, b0, b1, st));
            } else {
                AO_TRY(extrude<T>(env, l, b0, b1, lean, st, lean && env->defer_ring));
            }
        }
    }
    return 0;
}

template <typename T>
void fill_phase_args(AoEnv* env, PhaseArgs& pa, PhaseBuffers<T>& pb, int update_atm, int store_atm, int store_phase);

template <typename T>
int run_phase(AoEnv* env, int update_atm, int store_atm, hipStream_t st, int store_phase = 1) {
    AO_TRY(refresh_minmax<T>(env, st));
    if (env->use_coefs_img && env->c.dm_separable) {               // ELT-size DMs: once per env instead of once per tile
        if constexpr (std::is_same<T, float>::value) {
            if (env->dm_rows)                                       // Gy C on the matrix cores, in operand layout
                AO_TRY(launch_dm_rows(env->as<float>(env->coefs), env->act_idx, env->gya, env->dm_rows, env->E, env->R, env->nAct, env->A,
                                      env->ga_stride, st));
        }
        if (env->coefs_img)
            AO_TRY(launch_coefs_image<T>(env->as<T>(env->coefs), env->act_idx, env->as<T>(env->coefs_img), env->E, env->nAct, env->A, st));
    }
    PhaseArgs pa;
    PhaseBuffers<T> pb;
    fill_phase_args<T>(env, pa, pb, update_atm, store_atm, store_phase);
    AO_PROF(env, PHASE, st);
    return launch_phase<T>(pa, pb, env->E, env->R, env->nAct, env->A, env->c.atm_wavelength, env->c.src_wavelength,
                           (env->use_mfma ? 1 : 0) | (env->debug_ablate << 8), st);
}

// the WFS camera on the frame in HBM (detector.hpp); every measurement is a new frame of the noise streams
template <typename T>
int apply_detector(AoEnv* env, bool sh, hipStream_t st) {
    if (!env->det.active) return 0;
    env->det.frame_counter += 1;
    AO_PROF(env, DETECTOR, st);
    return launch_detector<T>(env->as<T>(env->frame), env->as<T>(env->wfs_max), sh ? env->valid2d : nullptr, env->E,
                              env->c.cam_res, env->nSub, env->det, env->alias(), st);
}

// Shack-Hartmann spots, then the camera on the frame (self*self.cam, OOPAO/ShackHartmann.py:539-576).  [Round 3 measured the camera
// INSIDE the spots kernel once more (16-wave workgroups, alias tables in LDS, the noisy frame written once, bit-identical): 1140 us
// against 611 + 463 us for the two kernels at the ELT size -- both are bound by vector instructions, not by the frame's round trip.]
template <typename T>
int run_spots_and_camera(AoEnv* env, const ShConst<T>& sc, hipStream_t st) {
    {
        AO_PROF(env, SH_SPOTS, st);
        AO_TRY(launch_sh_spots<T>(env->as<T>(env->phase), sc, env->as<T>(env->frame), env->as<T>(env->wfs_max), env->E,
                                  env->R, env->nSub, env->nVal, st));
    }
    return apply_detector<T>(env, true, st);
}

template <typename T>
int run_wfs(AoEnv* env, hipStream_t st) {
    if (env->c.wfs_type == AOENV_WFS_PYRAMID) {
        if (!env->have[AOENV_C_PYR_MASK] || (env->c.pyr_n_theta > 1 && !env->have[AOENV_C_PYR_TT]))
            return fail("pyramid mask / modulation table has not been uploaded");
        PyrArgs<T> pa{};
        pa.phase = env->as<T>(env->phase);
        pa.amp = env->as<T>(env->amp);
        pa.tt = env->c.pyr_n_theta > 1 ? env->as<T>(env->pyr_tt) : nullptr;
        pa.mask = env->as<T>(env->pyr_mask);
        pa.tw = env->as<T>(env->pyr_tw);
        pa.t1 = reinterpret_cast<cx<T>*>(env->pyr_t1);
        pa.t2 = reinterpret_cast<cx<T>*>(env->pyr_t2);
        pa.frame = env->as<T>(env->frame);
        pa.plan = env->pyr_plan;
        pa.R = env->R;
        pa.N = env->c.pyr_n_res;
        pa.cam = env->c.cam_res;
        pa.off = env->c.pyr_n_res / 2 - env->R / 2;
        pa.centering = env->c.pyr_centering;
        pa.phasor_mult = env->c.pyr_centering ? env->c.pyr_n_res + 1 : 0;
        pa.n_env = env->E;
        pa.generic_fft = env->debug_ablate & (512 | 1024);
        PyrSlopeArgs<T> sl{};
        sl.frame = env->as<T>(env->frame);
        sl.valid_idx = env->subap_idx;
        sl.ref = env->as<T>(env->sh_ref);
        sl.signal = env->as<T>(env->signal);
        sl.cam = env->c.cam_res;
        sl.n_sub = env->nSub;
        sl.n_valid = env->nVal;
        sl.q_lo = env->c.pyr_q_lo;
        sl.q_hi = env->c.pyr_q_hi;
        sl.norm_valid_mean = env->c.pyr_norm_valid;
        sl.units = (T)env->units;
        AO_PROF(env, PYRAMID, st);
        AO_TRY(launch_pyramid<T>(pa, env->c.pyr_n_theta, env->pyr_chunk, st));
        AO_TRY(apply_detector<T>(env, false, st));                 // self*self.cam (OOPAO/Pyramid.py:987-1006)
        return launch_pyramid_slopes<T>(sl, env->E, st);
    }
    const ShConst<T> sc = sh_const<T>(env);
    AO_TRY(run_spots_and_camera<T>(env, sc, st));                  // spots, self*self.cam (OOPAO/ShackHartmann.py:539-576)
    {
        AO_PROF(env, SH_CENTROID, st);
        AO_TRY(launch_sh_centroid<T>(env->as<T>(env->frame), env->as<T>(env->wfs_max), sc, env->as<T>(env->signal),
                                     env->E, env->R, env->nSub, env->nVal, env->c.max_group, st));
    }
    return 0;
}

// dm.OPD = modes @ coefs for a dense (non-separable, e.g. two chained mirrors) DM: [E][R^2] = coefs [E][A] . modes [R^2][A]^T
template <typename T>
int refresh_dense_dm(AoEnv* env, hipStream_t st) {
    if (env->c.dm_separable) return 0;
    if (sizeof(T) == 4 && env->use_mfma)                           // one K slice: the product lands in dm_opd directly
        return launch_gemm_nt_mfma(reinterpret_cast<const float*>(env->coefs), reinterpret_cast<const float*>(env->modes),
                                   reinterpret_cast<float*>(env->dm_opd), env->E, env->R * env->R, env->A, env->A, env->A, 1, st);
    return launch_gemm_nt<T>(env->as<T>(env->coefs), env->as<T>(env->modes), env->as<T>(env->dm_opd), env->E,
                             env->R * env->R, env->A, env->A, env->A, env->R * env->R, st);
}

template <typename T>
FinishArgs<T> finish_args(AoEnv* env, const T* d_action, T* d_obs, T* d_reward, T* d_strehl, int telemetry_index,
                          int integrate, double gain, int splits) {
    FinishArgs<T> fa{};
    fa.v = env->as<T>(env->vbuf);
    fa.splits = splits;
    fa.act_idx = env->act_idx;
    fa.action = d_action;
    fa.coefs = env->as<T>(env->coefs);
    fa.dm_prev = env->as<T>(env->dm_prev);
    fa.obs = d_obs;
    fa.reward = d_reward;
    fa.ret = env->as<T>(env->ret_acc);
    fa.strehl = d_strehl;
    fa.scal = env->as<T>(env->scal);
    fa.total = env->as<T>(env->total);
    fa.residual = env->as<T>(env->residual);
    fa.part = env->part;
    fa.n_tiles = env->n_tiles;
    fa.n_pupil = env->n_pupil;
    fa.telemetry_index = telemetry_index;
    fa.n_act = env->nAct;
    fa.n_valid_act = env->A;
    fa.do_integrate = integrate;
    fa.leak = (T)env->c.leak;
    fa.gain_from_obs = (T)gain;
    fa.src_scale = 6.283185307179586476925286766559 / env->c.src_wavelength;
    return fa;
}

// v = R s.  With the factors of the reconstructor on the device (R = M2C calib.M, MAIN/OOPAOEnv/OOPAOEnv.py:295, 381) the
// product is chained, t = M s then v = M2C t: K (nSig + A) multiply-adds and bytes per env instead of A nSig -- 11x fewer at the
// ELT size (A = 5209, nSig = 10048, K = 300: 209 MB of dense reconstructor streamed per step), 4.5x at 40x40 Pyramid size.
template <typename T>
int recon_product(AoEnv* env, int* splits, hipStream_t st) {
    const int Kp = (env->n_modes + 3) & ~3;
    if (env->n_modes > 0 && env->use_factored_recon && env->fac_m2c && env->tbuf &&
        (size_t)Kp * (env->nSig + env->A) < (size_t)env->A * env->nSig) {
        T* t = env->as<T>(env->tbuf);
        if constexpr (std::is_same<T, float>::value) {
            if (env->use_mfma) {
                const int s1 = gemm_splits(env->E, Kp, env->nSig);
                AO_TRY(launch_gemm_nt_mfma(env->as<float>(env->signal), env->as<float>(env->fac_m), t, env->E, Kp, env->nSig, env->nSig,
                                           env->nSig, s1, st));
                // (32 and more column tiles of 64 actuators fill the chip for any shard: no split of the short K, one output slab instead of
                //  K / 64 of them -- at the ELT size 11 MB written and read back by the epilogue instead of 43; the choice does not depend on E)
                *splits = (cdiv(env->A, 64) >= 32 && !(env->debug_ablate & 8192)) ? 1 : gemm_splits(env->E, env->A, Kp);
                // a long chain read by many column tiles is summed once first (same order of the additions: same bits)
                int xs = s1;
                if (s1 > 1 && cdiv(env->A, 64) >= 8 && !(env->debug_ablate & 4096)) {
                    AO_TRY(launch_sum_slabs(t, (size_t)env->E * Kp, s1, st));
                    xs = 1;
                }
                return launch_gemm_nt_mfma(t, env->as<float>(env->fac_m2c), env->as<float>(env->vbuf), env->E, env->A, Kp, Kp, Kp, *splits,
                                           st, xs, (size_t)env->E * Kp);
            }
        }
        *splits = 1;
        AO_TRY(launch_gemm_nt<T>(env->as<T>(env->signal), env->as<T>(env->fac_m), t, env->E, Kp, env->nSig, env->nSig, env->nSig, Kp, st));
        return launch_gemm_nt<T>(t, env->as<T>(env->fac_m2c), env->as<T>(env->vbuf), env->E, env->A, Kp, Kp, Kp, env->A, st);
    }
    return gemm_dispatch<T>(env, env->as<T>(env->signal), env->as<T>(env->recon), env->as<T>(env->vbuf), env->E, env->A, env->nSig,
                            splits, st);
}

template <typename T>
int run_recon(AoEnv* env, const T* d_action, T* d_obs, T* d_reward, T* d_strehl, int telemetry_index, int integrate,
              double gain, hipStream_t st) {
    int splits = 1;
    {
        AO_PROF(env, GEMM_RECON, st);
        AO_TRY(recon_product<T>(env, &splits, st));
    }
    {
        FinishArgs<T> fa = finish_args<T>(env, d_action, d_obs, d_reward, d_strehl, telemetry_index, integrate, gain, splits);
        AO_PROF(env, RECON_FINISH, st);
        AO_TRY(launch_recon_finish<T>(fa, env->E, st));
    }
    if (integrate) AO_TRY(refresh_dense_dm<T>(env, st));
    return 0;
}

// ---- the whole step as one kernel (step_kernel.hip) -------------------------------------------------------------
template <typename T>
bool fused_step_ok(const AoEnv*) { return false; }
template <>
bool fused_step_ok<float>(const AoEnv* env) {
    return env->use_fused_step && env->use_fast_wfs && env->use_mfma && env->use_fused_tail && env->c.wfs_type == AOENV_WFS_SH && env->c.dm_separable && env->n_modes > 0 &&
           env->c.max_group == 1 && env->L > 0 && env->uniform && env->c.cam_res == env->R && env->debug_ablate == 0 &&
           step_fused_supported(env->R, env->nSub, env->nVal, env->nAct, env->n_modes) != 0;
}

template <typename T>
void fill_phase_args(AoEnv* env, PhaseArgs& pa, PhaseBuffers<T>& pb, int update_atm, int store_atm, int store_phase) {
    pa = PhaseArgs{};
    pa.n_layer = env->L;
    pa.S = env->S;
    pa.foot = (env->N / 2 - env->R / 2) + 1;
    for (int l = 0; l < env->L; ++l) {                            // the R x R footprint of an on-axis source in layer l's grid
        pa.S_l[l] = env->Sl[l];                                    // (OOPAO/Atmosphere.py:226-232: centre N_l // 2)
        pa.foot_l[l] = (env->Nl[l] / 2 - env->R / 2) + 1;
    }
    pa.update_atm = (update_atm && env->L > 0 && !env->atm_user_defined) ? 1 : 0;
    pa.store_atm = store_atm;
    pa.store_phase = store_phase;
    for (int l = 0; l < env->L; ++l) {
        pa.screen[l] = env->screen_ptr(0, l);
        pa.minmax[l] = env->minmax_ptr(l);
        pa.minmax_dirty[l] = env->minmax_dirty[l] ? 1 : 0;
        LayerTaps& t = pa.taps[l];
        t.oy = env->org[l][0];
        t.ox = env->org[l][1];
        taps_from_buff(env->clk[l].buff, t);
        t.weight = env->layer_weight[l];
    }
    pa.env_taps = env->per_env_wind ? env->env_taps : nullptr;
    pa.n_env = env->E;
    pb = PhaseBuffers<T>{};
    pb.opd_atm = env->as<T>(env->opd_atm);
    pb.coefs = env->as<T>(env->coefs);
    pb.coefs_img = env->use_coefs_img && env->coefs_img ? env->as<T>(env->coefs_img) : nullptr;
    pb.s1a = env->use_coefs_img && env->c.dm_separable ? env->dm_rows : nullptr;
    pb.dm_opd = env->c.dm_separable ? nullptr : env->as<T>(env->dm_opd);
    pb.gx = env->as<T>(env->gx);
    pb.gy = env->as<T>(env->gy);
    pb.gxt = env->as<T>(env->gxt);
    pb.gxa = env->gxa;
    pb.gya = env->gya;
    pb.ga_stride = env->ga_stride;
    pb.act_idx = env->act_idx;
    pb.pupil = env->pupil;
    pb.phase = env->as<T>(env->phase);
    pb.part = env->part;
    pb.wfs_max = env->as<T>(env->wfs_max);
}

template <typename T>
int run_fused_step(AoEnv*, int, const void*, void*, void*, void*, double, hipStream_t) { return fail("fused step: float32 only"); }
template <>
int run_fused_step<float>(AoEnv* env, int i, const void* d_action, void* d_obs, void* d_reward, void* d_strehl, double gain,
                          hipStream_t st) {
    if (env->amp_pupil_dirty) {
        const size_t R2 = (size_t)env->R * env->R;
        if (env->h_pupil.size() != R2 || env->h_amp.size() != R2) return fail("pupil / WFS amplitude have not been uploaded");
        std::vector<float> t(R2);
        for (size_t q = 0; q < R2; ++q) t[q] = env->h_pupil[q] ? (float)env->h_amp[q] : -1.f;
        AO_HIP(hipMemcpyAsync(env->amp_pupil, t.data(), R2 * sizeof(float), hipMemcpyHostToDevice, st));
        AO_HIP(hipStreamSynchronize(st));
        env->amp_pupil_dirty = false;
    }
    StepArgs a{};
    PhaseArgs pa;
    PhaseBuffers<float> pb;
    fill_phase_args<float>(env, pa, pb, 1, env->store_opd_atm ? 1 : 0, 1);
    a.k = make_phase_kargs<float>(pa, pb, env->R, env->nAct, env->A, env->c.atm_wavelength, env->c.src_wavelength);
    a.sc = sh_const<float>(env);
    a.fa = finish_args<float>(env, static_cast<const float*>(d_action), static_cast<float*>(d_obs),
                              static_cast<float*>(d_reward), static_cast<float*>(d_strehl), i, 1, gain, 1);
    a.fa.n_tiles = 1;
    a.frame = env->as<float>(env->frame);
    a.signal = env->as<float>(env->signal);
    a.wfs_max = env->as<float>(env->wfs_max);
    a.fac_m = env->as<float>(env->fac_m);
    a.fac_m2c_t = env->as<float>(env->fac_m2c_t);
    a.slot_of = env->slot_of;
    a.amp_pupil = env->amp_pupil;
    a.gxa = env->gxa;
    a.gya = env->gya;
    if (env->det.active) env->det.frame_counter += 1;              // every measurement is a new frame of the noise streams
    a.det = env->det;
    a.pa = env->alias();
    for (int l = 0; l < env->L; ++l) {
        a.ring_x[l] = env->ring_pending[l] ? static_cast<const float*>(env->ring_src[l]) : nullptr;
        a.ring_splits[l] = env->ring_pending[l];
    }
    a.outer_idx = env->outer_idx;
    a.n_outer = env->nout;
    for (int l = 0; l < env->L; ++l) {
        const bool g = env->gather_next[l] && env->ring_pending[l] && env->ahead[l].valid;
        a.next_zx[l] = g ? static_cast<float*>(env->zx_pipe_ptr(env->ahead[l].buf, l)) : nullptr;
        a.next_sx[l] = env->ahead[l].sx;
        a.next_sy[l] = env->ahead[l].sy;
        if (env->gather_next[l] && !g) env->ahead[l].valid = false;
        env->gather_next[l] = false;
    }
    a.inner_idx = env->inner_idx;
    a.n_inner = env->nin;
    a.zx_ld = env->K;
    a.n_modes = env->n_modes;
    a.n_subap = env->nSub;
    a.n_valid = env->nVal;
    a.n_env = env->E;
    {
        AO_PROF(env, ENV_STEP, st);
        AO_TRY(launch_env_step(a, st));
    }
    for (int l = 0; l < env->L; ++l) env->minmax_dirty[l] = false;   // the kernel recomputed and stored them
    for (int l = 0; l < env->L; ++l) env->ring_pending[l] = 0;        // ... and wrote the deferred rings
    return 0;
}

template <typename T>
int step_t(AoEnv* env, int i, const void* d_action, void* d_obs, void* d_frame, void* d_reward, void* d_strehl,
           double gain, hipStream_t st) {
    const bool fused_step = fused_step_ok<T>(env);
    AO_TRY(advance_atmosphere<T>(env, fused_step, st));
    env->atm_user_defined = false;
    if (fused_step) {
        AO_TRY(run_fused_step<T>(env, i, d_action, d_obs, d_reward, d_strehl, gain, st));
        if (d_frame)
            AO_HIP(hipMemcpyAsync(d_frame, env->frame, (size_t)env->E * env->c.cam_res * env->c.cam_res * sizeof(T),
                                  hipMemcpyDeviceToDevice, st));
        return 0;
    }
    AO_TRY(run_phase<T>(env, 1, env->store_opd_atm ? 1 : 0, st));
    // one workgroup per env re-reads the factors from L2: wins while launch latency dominates (measured: 19.5 us vs
    // 32.5 us at 256 envs, 96 us vs 75 us at 2048), the batched MFMA GEMM path takes over for large shards
    const bool fused = env->c.wfs_type == AOENV_WFS_SH && env->use_fused_tail && env->n_modes > 0 && env->E <= 1024;
    if (fused) {
        const ShConst<T> sc = sh_const<T>(env);
        AO_TRY(run_spots_and_camera<T>(env, sc, st));
        FinishArgs<T> fa = finish_args<T>(env, static_cast<const T*>(d_action), static_cast<T*>(d_obs),
                                          static_cast<T*>(d_reward), static_cast<T*>(d_strehl), i, 1, gain, 1);
        int rc;
        {
            AO_PROF(env, SH_TAIL, st);
            rc = launch_sh_tail<T>(env->as<T>(env->frame), env->as<T>(env->wfs_max), sc, env->as<T>(env->signal),
                                   env->as<T>(env->fac_m), env->as<T>(env->fac_m2c_t), env->n_modes, fa, env->E, env->R,
                                   env->nSub, env->nVal, env->c.max_group, st);
        }
        if (rc > 0) return rc;
        if (rc == 0) {
            AO_TRY(refresh_dense_dm<T>(env, st));
        } else {                                                   // does not fit in LDS: the separate kernels
            {
                AO_PROF(env, SH_CENTROID, st);
                AO_TRY(launch_sh_centroid<T>(env->as<T>(env->frame), env->as<T>(env->wfs_max), sc, env->as<T>(env->signal),
                                             env->E, env->R, env->nSub, env->nVal, env->c.max_group, st));
            }
            AO_TRY(run_recon<T>(env, static_cast<const T*>(d_action), static_cast<T*>(d_obs), static_cast<T*>(d_reward),
                                static_cast<T*>(d_strehl), i, 1, gain, st));
        }
    } else {
        AO_TRY(run_wfs<T>(env, st));
        AO_TRY(run_recon<T>(env, static_cast<const T*>(d_action), static_cast<T*>(d_obs), static_cast<T*>(d_reward),
                            static_cast<T*>(d_strehl), i, 1, gain, st));
    }
    if (d_frame)
        AO_HIP(hipMemcpyAsync(d_frame, env->frame, (size_t)env->E * env->c.cam_res * env->c.cam_res * sizeof(T),
                              hipMemcpyDeviceToDevice, st));
    return 0;
}

struct BufInfo {
    void* ptr;
    size_t bytes;
};

int buf_info(AoEnv* env, int which, BufInfo* b) {
    const size_t z = env->esz, E = env->E, R2 = (size_t)env->R * env->R;
    switch (which) {
        case AOENV_B_SCREEN: *b = {nullptr, env->scr_off[env->L] * z}; return 0;   // gathered per layer: [E][S_l^2] blocks, layer after layer
        case AOENV_B_OPD_ATM: *b = {env->opd_atm, E * R2 * z}; return 0;
        case AOENV_B_COEFS: *b = {env->coefs, E * env->A * z}; return 0;
        case AOENV_B_PHASE: *b = {env->phase, E * R2 * z}; return 0;
        case AOENV_B_FRAME: *b = {env->frame, E * (size_t)env->c.cam_res * env->c.cam_res * z}; return 0;
        case AOENV_B_SIGNAL: *b = {env->signal, E * env->nSig * z}; return 0;
        case AOENV_B_TOTAL: *b = {env->total, (size_t)env->c.n_loop * E * z}; return 0;
        case AOENV_B_RESIDUAL: *b = {env->residual, (size_t)env->c.n_loop * E * z}; return 0;
        case AOENV_B_WFS_MAX: *b = {env->wfs_max, E * z}; return 0;
        case AOENV_B_XI: *b = {env->last_zx ? const_cast<void*>(env->last_zx) : env->zx, E * env->Kl[env->last_zx_layer] * z}; return 0;
        case AOENV_B_MT_STATE: *b = {nullptr, (size_t)env->L * E * (kMtN + 1) * 4}; return 0;     // packed on the host
        case AOENV_B_COUNTERS: *b = {nullptr, 16}; return 0;
        case AOENV_B_DM_PREV: *b = {env->dm_prev, E * env->A * z}; return 0;
        default: return fail("unknown buffer id %d", which);
    }
}

}  // namespace

// Every entry point runs on the env's device and hands the calling thread's current device back on return: PyTorch shares
// this runtime, and an env on device k must not redirect the caller's later allocations and launches to k.
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = prev == device || hipSetDevice(device) == hipSuccess;
        if (prev == device) prev = -1;                               // nothing to restore
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define AO_CHECK_ENV(env)                                                             \
    if (!(env)) return fail("null AoEnv");                                            \
    DeviceGuard ao_device_guard((env)->device);                                       \
    if (!ao_device_guard.ok) return fail("hipSetDevice(%d) failed", (env)->device)
// the ring tables of ONE layer: [A | B] (float64 [n_outer_l][n_inner_l + n_outer_l]) or the flat indices of its Z / X pixels
static int upload_layer_table(AoEnv* env, int kind, int l, const void* h, size_t bytes) {
    if (l < 0 || l >= env->L) return fail("layer %d outside [0, %d)", l, env->L);
    const int S = env->Sl[l], N = env->Nl[l];
    auto need = [&](size_t n) { return bytes == n ? 0 : fail("ring table %d of layer %d: got %zu bytes, expected %zu", kind, l, bytes, n); };
    if (kind == AOENV_C_AB) {
        AO_TRY(need((size_t)env->noutl[l] * env->Kl[l] * 8));
        AO_TRY(sync_lookaheads(env));                              // a ring computed ahead used the old operators
        AO_TRY(upload_real(env, env->ab_l[l], static_cast<const double*>(h), (size_t)env->noutl[l] * env->Kl[l]));
        env->have_ab[l] = true;
        return 0;
    }
    if (kind != AOENV_C_INNER_IDX && kind != AOENV_C_OUTER_IDX) return fail("table %d is not a per-layer table", kind);
    const int cnt = kind == AOENV_C_INNER_IDX ? env->ninl[l] : env->noutl[l];
    AO_TRY(need((size_t)cnt * 4));
    const int32_t* ix = static_cast<const int32_t*>(h);
    for (int i = 0; i < cnt; ++i)
        if (ix[i] < 0 || ix[i] >= S * S) return fail("ring index %d out of the %dx%d screen", ix[i], S, S);
    if (kind == AOENV_C_INNER_IDX)          // the gather reads idx - sy*S - sx with |s| <= 1
        for (int i = 0; i < cnt; ++i) {
            const int r = ix[i] / S, c = ix[i] % S;
            if (r < 1 || r > N || c < 1 || c > N) return fail("inner ring index %d is not interior", ix[i]);
        }
    AO_HIP(hipMemcpy(kind == AOENV_C_INNER_IDX ? env->inner_idx_l[l] : env->outer_idx_l[l], h, (size_t)cnt * 4, hipMemcpyHostToDevice));
    (kind == AOENV_C_INNER_IDX ? env->have_in[l] : env->have_out[l]) = true;
    return 0;
}

template <typename T>
int atm_update_t(AoEnv* env, hipStream_t st) {
    AO_TRY(advance_atmosphere<T>(env, false, st));
    env->atm_user_defined = false;
    return run_phase<T>(env, 1, 1, st);
}

#define AO_DISPATCH(env, fn, ...) ((env)->c.dtype == AOENV_F32 ? fn<float>(__VA_ARGS__) : fn<double>(__VA_ARGS__))

extern "C" {

const char* aoenv_last_error(void) { return g_err.c_str(); }
int aoenv_abi_version(void) { return AOENV_ABI_VERSION; }

int aoenv_create(const AoCfg* cfg, int device, AoEnv** out) {
    if (!cfg || !out) return fail("aoenv_create: null argument");
    if (cfg->abi_version != AOENV_ABI_VERSION) return fail("ABI version %d != %d", cfg->abi_version, AOENV_ABI_VERSION);
    if (cfg->dtype != AOENV_F32 && cfg->dtype != AOENV_F64) return fail("bad dtype %d", cfg->dtype);
    if (cfg->n_env < 1 || cfg->resolution < 2) return fail("bad n_env / resolution");
    if (cfg->n_layer < 0 || cfg->n_layer > kMaxLayer) return fail("n_layer %d out of range [0, %d]", cfg->n_layer, kMaxLayer);
    if (cfg->n_subap < 1 || cfg->resolution % cfg->n_subap) return fail("resolution %% n_subap != 0");
    if (cfg->n_layer > 0) {
        if (cfg->layer_res < cfg->resolution + 4) return fail("layer_res %d < R + 4", cfg->layer_res);
        if (cfg->n_inner != 8 * cfg->layer_res - 16 || cfg->n_outer != 4 * cfg->layer_res + 4)
            return fail("n_inner / n_outer do not match layer_res");
        for (int l = 0; l < cfg->n_layer; ++l)
            if (cfg->layer_res_l[l] != 0 && cfg->layer_res_l[l] < cfg->resolution + 4)
                return fail("layer_res_l[%d] = %d < R + 4", l, cfg->layer_res_l[l]);
    }
    if (cfg->n_signal != 2 * cfg->n_valid_subap) return fail("n_signal != 2 n_valid_subap");
    if (cfg->wfs_type != AOENV_WFS_SH && cfg->wfs_type != AOENV_WFS_PYRAMID) return fail("unknown wfs_type %d", cfg->wfs_type);
    if (cfg->wfs_type == AOENV_WFS_PYRAMID) {
        if (cfg->pyr_n_res < cfg->resolution || cfg->pyr_n_res % 2 || cfg->cam_res < 1 || cfg->pyr_n_res % cfg->cam_res)
            return fail("pyramid: nRes %d must be even, >= R and a multiple of the camera size %d", cfg->pyr_n_res, cfg->cam_res);
        if (cfg->pyr_n_theta < 1) return fail("pyramid: n_theta must be >= 1");
        if (cfg->pyr_q_lo < 0 || cfg->pyr_q_hi + cfg->n_subap > cfg->cam_res) return fail("pyramid: quadrants outside the camera");
    }
    if (cfg->max_group < 1) return fail("max_group must be >= 1");
    int ndev = 0;
    AO_HIP(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail("device %d not in [0, %d)", device, ndev);
    DeviceGuard ao_device_guard(device);
    if (!ao_device_guard.ok) return fail("hipSetDevice(%d) failed", device);

    AoEnv* e = new (std::nothrow) AoEnv();
    if (!e) return fail("out of host memory");
    e->c = *cfg;
    e->device = device;
    e->esz = cfg->dtype == AOENV_F32 ? 4 : 8;
    e->R = cfg->resolution; e->N = cfg->layer_res; e->S = cfg->layer_res + 2; e->A = cfg->n_valid_act;
    e->nAct = cfg->n_act; e->E = cfg->n_env; e->L = cfg->n_layer; e->nin = cfg->n_inner; e->nout = cfg->n_outer;
    e->K = cfg->n_inner + cfg->n_outer; e->nSig = cfg->n_signal; e->nSub = cfg->n_subap; e->nVal = cfg->n_valid_subap;
    e->p = e->R / e->nSub; e->n = 2 * e->p;
    for (int l = 0; l < e->L; ++l) {
        e->Nl[l] = cfg->layer_res_l[l] ? cfg->layer_res_l[l] : cfg->layer_res;
        e->Sl[l] = e->Nl[l] + 2;
        e->ninl[l] = 8 * e->Nl[l] - 16;
        e->noutl[l] = 4 * e->Nl[l] + 4;
        e->Kl[l] = e->ninl[l] + e->noutl[l];
        e->scr_off[l + 1] = e->scr_off[l] + (size_t)e->E * e->Sl[l] * e->Sl[l];
        e->Kmax = std::max(e->Kmax, e->Kl[l]);
        e->noutmax = std::max(e->noutmax, e->noutl[l]);
        e->Smax = std::max(e->Smax, e->Sl[l]);
        if (e->Nl[l] != e->Nl[0]) e->uniform = false;
    }
    if (e->L > 0) {                                                // "layer 0" scalars: the grid of every layer when uniform
        e->N = e->Nl[0]; e->S = e->Sl[0]; e->nin = e->ninl[0]; e->nout = e->noutl[0]; e->K = e->Kl[0];
    }
    const size_t z = e->esz, E = e->E, R2 = (size_t)e->R * e->R;
    int rc = 0;
    auto A_ = [&](void** p, size_t bytes) { if (!rc) rc = dmalloc(e, p, bytes); };
    if (e->L > 0) {
        A_(&e->screen[0], e->scr_off[e->L] * z);
        A_(&e->minmax, (size_t)e->L * E * 2 * z);
        A_((void**)&e->mt_state, (size_t)2 * e->L * E * kMtN * 4);
        A_((void**)&e->mt_pos, (size_t)2 * e->L * E * 4);
        if (cfg->dtype == AOENV_F32) {                              // ring pipeline (fused float32 path)
            A_(&e->zx_pipe, (size_t)2 * e->L * E * e->Kmax * z);
        }
        A_(&e->zx, E * e->Kmax * z);
        A_(&e->xbuf, (size_t)e->L * kMaxSplits * E * e->noutmax * z);
        for (int l = 0; l < e->L; ++l) {                           // layers on the same grid share one set of tables
            int same = -1;
            for (int j = 0; j < l; ++j)
                if (e->Nl[j] == e->Nl[l]) { same = j; break; }
            if (same >= 0 && e->uniform) {                         // (non-uniform shards: operators per layer -- r0 / L0 scale them alike, but
                e->ab_l[l] = e->ab_l[same];                        //  the reference computes them per layer too when fov != 0)
                e->inner_idx_l[l] = e->inner_idx_l[same];
                e->outer_idx_l[l] = e->outer_idx_l[same];
                continue;
            }
            A_(&e->ab_l[l], (size_t)e->noutl[l] * e->Kl[l] * z);
            A_((void**)&e->inner_idx_l[l], (size_t)e->ninl[l] * 4);
            A_((void**)&e->outer_idx_l[l], (size_t)e->noutl[l] * 4);
        }
        e->ab = e->ab_l[0];
        e->inner_idx = e->inner_idx_l[0];
        e->outer_idx = e->outer_idx_l[0];
    }
    A_(&e->gx, (size_t)e->R * e->nAct * z);
    A_(&e->gy, (size_t)e->R * e->nAct * z);
    A_(&e->gxt, (size_t)((e->nAct + 3) & ~3) * (size_t)(cdiv(e->R, 128) * 128) * z);
    if (!cfg->dm_separable) {
        A_(&e->modes, R2 * e->A * z);
        A_(&e->dm_opd, E * R2 * z);
    }
    A_((void**)&e->act_idx, (size_t)e->A * 4);
    A_((void**)&e->pupil, R2);
    A_(&e->opd_atm, E * R2 * z);
    A_(&e->coefs, E * e->A * z);
    A_(&e->dm_prev, E * e->A * z);                                 // self.dm_prev = self.dm.coefs.copy() = 0 (OOPAOEnv.py:313-314)
    A_(&e->phase, E * R2 * z);
    A_(&e->scal, E * 4 * z);
    e->n_tiles = phase_tiles(e->R, e->nAct, e->esz);
    A_((void**)&e->part, E * (size_t)e->n_tiles * 4 * sizeof(double));
    A_(&e->total, (size_t)cfg->n_loop * E * z);
    A_(&e->residual, (size_t)cfg->n_loop * E * z);
    A_(&e->wfs_max, E * z);
    A_(&e->amp, R2 * z);
    A_((void**)&e->subap_idx, (size_t)e->nVal * 4);
    A_((void**)&e->valid2d, (size_t)e->nSub * e->nSub);
    A_((void**)&e->slot_of, (size_t)e->nSub * e->nSub * sizeof(short));
    A_((void**)&e->amp_pupil, R2 * sizeof(float));
    e->ga_stride = std::max(8, ((cdiv(e->nAct, 4) + 3) / 4) * 4);
    A_((void**)&e->gxa, (size_t)(cdiv(e->R, 128) * 128) * 4 * e->ga_stride * sizeof(float));
    A_((void**)&e->gya, (size_t)(cdiv(e->R, 128) * 128) * 4 * e->ga_stride * sizeof(float));
    if (e->A > 1024 && !rc) { rc = alloc_dm_rows(e); e->use_coefs_img = true; }
    A_(&e->sh_ref, (size_t)2 * e->nVal * z);
    A_(&e->tw, (size_t)e->n * 2 * z);
    A_(&e->phs, (size_t)e->p * 2 * z);
    A_(&e->frame, E * (size_t)cfg->cam_res * cfg->cam_res * z);
    A_(&e->signal, E * e->nSig * z);
    A_(&e->recon, (size_t)e->A * e->nSig * z);
    A_(&e->fac_m, (size_t)kMaxModes * e->nSig * z);
    A_(&e->fac_m2c_t, (size_t)kMaxModes * e->A * z);
    if (cfg->wfs_type == AOENV_WFS_PYRAMID) {
        const size_t N = cfg->pyr_n_res;
        e->pyr_chunk = cfg->pyr_n_theta < 4 ? cfg->pyr_n_theta : 4;
        A_(&e->pyr_mask, N * N * 2 * z);
        A_(&e->pyr_tt, (size_t)cfg->pyr_n_theta * R2 * z);
        A_(&e->pyr_tw, N * 2 * z);
        A_(&e->pyr_t1, E * e->pyr_chunk * (size_t)e->R * N * 2 * z);
        A_(&e->pyr_t2, E * e->pyr_chunk * N * N * 2 * z);
    }
    A_(&e->vbuf, (size_t)kMaxSplits * E * e->A * z);
    {
        const PoissonAliasHost& ph = poisson_alias_host();
        int budget = (int)ph.tab.size();
        if (cfg->wfs_type == AOENV_WFS_SH && e->p == fast6::P && e->R <= 128 && e->nAct <= 32) budget = std::min(budget, step_alias_capacity(e->nAct));
        ph.prefix(budget, &e->alias_words, &e->alias_lmax);
        if (e->alias_lmax < palias::kCoarseStep && !rc) rc = fail("aoenv_create: no room for the photon-noise tables");
        A_((void**)&e->alias_tab, ph.tab.size() * 4);
        if (!rc && hipMemcpy(e->alias_tab, ph.tab.data(), ph.tab.size() * 4, hipMemcpyHostToDevice) != hipSuccess)
            rc = fail("aoenv_create: upload of the photon-noise tables failed");
    }
    if (rc) { aoenv_destroy(e); return rc; }
    for (int l = 0; l < e->L; ++l) {
        e->mt_cur[l] = e->mt_state + (size_t)l * E * kMtN;
        e->mt_alt[l] = e->mt_state + (size_t)(e->L + l) * E * kMtN;
        e->pos_cur[l] = e->mt_pos + (size_t)l * E;
        e->pos_alt[l] = e->mt_pos + (size_t)(e->L + l) * E;
    }
    // DFT twiddles w^k = exp(-2 pi i k / n) and the centring phasor exp(-i pi (n+1)/n x) at x = a + lo
    // (OOPAO/ShackHartmann.py:208-209), in float64 then converted
    if (cfg->wfs_type == AOENV_WFS_SH) {
        const int n = e->n, p = e->p, lo = n / 2 - p / 2;
        std::vector<double> tw(2 * n), ph(2 * p);
        const double pi = 3.14159265358979323846;
        for (int k = 0; k < n; ++k) { tw[2 * k] = std::cos(2 * pi * k / n); tw[2 * k + 1] = -std::sin(2 * pi * k / n); }
        for (int a = 0; a < p; ++a) {
            const double ang = pi * (n + 1) / n * (a + lo);
            ph[2 * a] = std::cos(ang); ph[2 * a + 1] = -std::sin(ang);
        }
        rc = upload_real(e, e->tw, tw.data(), tw.size());
        if (!rc) rc = upload_real(e, e->phs, ph.data(), ph.size());
        if (rc) { aoenv_destroy(e); return rc; }
    }
    if (cfg->wfs_type == AOENV_WFS_PYRAMID) {
        const int N = cfg->pyr_n_res;
        rc = make_fft_plan(N, &e->pyr_plan);
        std::vector<double> tw(2 * (size_t)N);
        const double pi = 3.14159265358979323846;
        for (int k = 0; k < N; ++k) { tw[2 * k] = std::cos(2 * pi * k / N); tw[2 * k + 1] = -std::sin(2 * pi * k / N); }
        if (!rc) rc = upload_real(e, e->pyr_tw, tw.data(), tw.size());
        if (rc) { aoenv_destroy(e); return rc; }
    }
    *out = e;
    return 0;
}

int aoenv_upload_layer(AoEnv* env, int kind, int layer, const void* h, size_t bytes) {
    AO_CHECK_ENV(env);
    if (!h) return fail("aoenv_upload_layer: null data");
    if (env->L == 0) return fail("no atmosphere in this shard");
    return upload_layer_table(env, kind, layer, h, bytes);
}

int aoenv_destroy(AoEnv* env) {
    if (!env) return 0;
    DeviceGuard ao_device_guard(env->device);
    for (auto& e : env->prof_ev) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (void* p : env->allocs) (void)hipFree(p);
    delete env;
    return 0;
}

int aoenv_upload(AoEnv* env, int kind, const void* h, size_t bytes) {
    AO_CHECK_ENV(env);
    if (!h) return fail("aoenv_upload: null data");
    const size_t R2 = (size_t)env->R * env->R;
    auto need = [&](size_t n) { return bytes == n ? 0 : fail("aoenv_upload(kind=%d): got %zu bytes, expected %zu", kind, bytes, n); };
    const double* d = static_cast<const double*>(h);
    switch (kind) {
        case AOENV_C_PUPIL: {
            AO_TRY(need(R2));
            AO_HIP(hipMemcpy(env->pupil, h, R2, hipMemcpyHostToDevice));
            const uint8_t* pu = static_cast<const uint8_t*>(h);
            env->h_pupil.assign(pu, pu + R2);
            env->amp_pupil_dirty = true;
            env->n_pupil = 0;
            for (size_t i = 0; i < R2; ++i) env->n_pupil += pu[i] != 0;
            break;
        }
        case AOENV_C_AB:
        case AOENV_C_INNER_IDX:
        case AOENV_C_OUTER_IDX:
            // one set of ring tables for every layer: shards whose layers share one grid (fov = 0, or no layer above the ground)
            if (env->L == 0) return fail("no atmosphere in this shard");
            if (!env->uniform) return fail("the layers of this shard have grids of their own: upload the ring tables per layer (aoenv_upload_layer)");
            AO_TRY(upload_layer_table(env, kind, 0, h, bytes));
            for (int l = 1; l < env->L; ++l) {
                if (kind == AOENV_C_AB) env->have_ab[l] = true;
                else if (kind == AOENV_C_INNER_IDX) env->have_in[l] = true;
                else env->have_out[l] = true;
            }
            break;
        case AOENV_C_LAYER_WEIGHT:
            AO_TRY(need((size_t)env->L * 8));
            for (int l = 0; l < env->L; ++l) env->layer_weight[l] = d[l];
            break;
        case AOENV_C_DM_GX:
        case AOENV_C_DM_GY:
            AO_TRY(need((size_t)env->R * env->nAct * 8));
            AO_TRY(upload_real(env, kind == AOENV_C_DM_GX ? env->gx : env->gy, d, (size_t)env->R * env->nAct));
            {                                                      // MFMA operand layout (float32 kernels): zero padded
                const int st_ = env->ga_stride;
                std::vector<float> t((size_t)(cdiv(env->R, 128) * 128) * 4 * st_, 0.f);
                for (int x = 0; x < env->R; ++x)
                    for (int k = 0; k < env->nAct; ++k) t[ga_index(x, k, st_)] = (float)d[(size_t)x * env->nAct + k];
                AO_HIP(hipMemcpy(kind == AOENV_C_DM_GX ? env->gxa : env->gya, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
            }
            if (kind == AOENV_C_DM_GX) {
                const int nap = (env->nAct + 3) & ~3, rp = cdiv(env->R, 128) * 128;
                std::vector<double> t((size_t)nap * rp, 0.0);
                for (int x = 0; x < env->R; ++x)
                    for (int ix = 0; ix < env->nAct; ++ix) t[(size_t)ix * rp + x] = d[(size_t)x * env->nAct + ix];
                AO_TRY(upload_real(env, env->gxt, t.data(), t.size()));
            }
            break;
        case AOENV_C_DM_MODES:
            if (env->c.dm_separable) return fail("dense modes uploaded to a separable-DM shard");
            AO_TRY(need(R2 * env->A * 8));
            AO_TRY(upload_real(env, env->modes, d, R2 * env->A));
            break;
        case AOENV_C_ACT_IDX: {
            AO_TRY(need((size_t)env->A * 4));
            const int32_t* ix = static_cast<const int32_t*>(h);
            for (int i = 0; i < env->A; ++i)
                if (ix[i] < 0 || ix[i] >= env->nAct * env->nAct) return fail("actuator index %d out of range", ix[i]);
            AO_HIP(hipMemcpy(env->act_idx, h, (size_t)env->A * 4, hipMemcpyHostToDevice));
            break;
        }
        case AOENV_C_WFS_AMP:
            AO_TRY(need(R2 * 8));
            AO_TRY(upload_real(env, env->amp, d, R2));
            env->h_amp.assign(d, d + R2);
            env->amp_pupil_dirty = true;
            break;
        case AOENV_C_SH_SUBAP_IDX: {
            AO_TRY(need((size_t)env->nVal * 4));
            const int32_t* ix = static_cast<const int32_t*>(h);
            for (int i = 0; i < env->nVal; ++i)
                if (ix[i] < 0 || ix[i] >= env->nSub * env->nSub) return fail("lenslet index %d out of range", ix[i]);
            AO_HIP(hipMemcpy(env->subap_idx, h, (size_t)env->nVal * 4, hipMemcpyHostToDevice));
            std::vector<uint8_t> v2((size_t)env->nSub * env->nSub, 0);
            for (int i = 0; i < env->nVal; ++i) v2[ix[i]] = 1;
            AO_HIP(hipMemcpy(env->valid2d, v2.data(), v2.size(), hipMemcpyHostToDevice));
            std::vector<short> so((size_t)env->nSub * env->nSub, (short)-1);
            if (env->nVal <= 32767)
                for (int i = 0; i < env->nVal; ++i) so[ix[i]] = (short)i;
            AO_HIP(hipMemcpy(env->slot_of, so.data(), so.size() * sizeof(short), hipMemcpyHostToDevice));
            break;
        }
        case AOENV_C_SH_REF:
            AO_TRY(need((size_t)2 * env->nVal * 8));
            AO_TRY(upload_real(env, env->sh_ref, d, (size_t)2 * env->nVal));
            break;
        case AOENV_C_WFS_UNITS:
            AO_TRY(need(8));
            if (!(d[0] != 0)) return fail("slopes units must be non-zero");
            env->units = d[0];
            break;
        case AOENV_C_RECON:
            AO_TRY(need((size_t)env->A * env->nSig * 8));
            AO_TRY(upload_real(env, env->recon, d, (size_t)env->A * env->nSig));
            env->n_modes = 0;                                      // factors must be re-uploaded for a new R
            break;
        case AOENV_C_RECON_FACTORS: {
            // [K*nSig] M then [A*K] M2C, K from the size
            const size_t per = (size_t)env->nSig + env->A;
            if (bytes % (8 * per)) return fail("aoenv_upload(RECON_FACTORS): %zu bytes is not K*(nSig+A) doubles", bytes);
            const int K = (int)(bytes / (8 * per));
            if (K < 1 || K > kMaxModes) return fail("reconstructor rank %d outside [1, %d]", K, kMaxModes);
            AO_TRY(upload_real(env, env->fac_m, d, (size_t)K * env->nSig));
            const double* m2c = d + (size_t)K * env->nSig;
            std::vector<double> t((size_t)K * env->A);
            for (int a = 0; a < env->A; ++a)
                for (int k = 0; k < K; ++k) t[(size_t)k * env->A + a] = m2c[(size_t)a * K + k];
            AO_TRY(upload_real(env, env->fac_m2c_t, t.data(), t.size()));
            // the same factors for the batched (non-fused) reconstruction: M with zero rows up to Kp, M2C as [A][Kp]
            const int Kp = (K + 3) & ~3;
            if ((size_t)Kp > env->fac_cap) {
                AO_TRY(dmalloc(env, &env->fac_m2c, (size_t)env->A * Kp * env->esz));
                AO_TRY(dmalloc(env, &env->tbuf, (size_t)kMaxSplits * env->E * Kp * env->esz));
                env->fac_cap = (size_t)Kp;
            }
            if (Kp > K && Kp <= kMaxModes)
                AO_HIP(hipMemset(static_cast<char*>(env->fac_m) + (size_t)K * env->nSig * env->esz, 0, (size_t)(Kp - K) * env->nSig * env->esz));
            std::vector<double> mp((size_t)env->A * Kp, 0.0);
            for (int a = 0; a < env->A; ++a)
                for (int k = 0; k < K; ++k) mp[(size_t)a * Kp + k] = m2c[(size_t)a * K + k];
            AO_TRY(upload_real(env, env->fac_m2c, mp.data(), mp.size()));
            env->n_modes = K;
            break;
        }
        case AOENV_C_PYR_MASK: {
            if (env->c.wfs_type != AOENV_WFS_PYRAMID) return fail("not a pyramid shard");
            const size_t n = (size_t)env->c.pyr_n_res * env->c.pyr_n_res * 2;
            AO_TRY(need(n * 8));
            AO_TRY(upload_real(env, env->pyr_mask, d, n));
            break;
        }
        case AOENV_C_PYR_TT: {
            if (env->c.wfs_type != AOENV_WFS_PYRAMID) return fail("not a pyramid shard");
            const size_t n = (size_t)env->c.pyr_n_theta * R2;
            AO_TRY(need(n * 8));
            AO_TRY(upload_real(env, env->pyr_tt, d, n));
            break;
        }
        default: return fail("unknown constant id %d", kind);
    }
    env->have[kind] = true;
    return 0;
}

// ---- per-env clocks -----------------------------------------------------------------------------------------------------
static int alloc_env_clocks(AoEnv* env) {
    if (env->env_taps) return 0;
    const size_t n = (size_t)env->L * env->E;
    for (int b = 0; b < 2; ++b) {
        void* p_ = nullptr;
        AO_HIP(hipMalloc(&p_, n * sizeof(EnvClock)));
        env->allocs.push_back(p_);
        env->env_clk[b] = static_cast<EnvClock*>(p_);
    }
    void* t = nullptr;
    AO_HIP(hipMalloc(&t, n * sizeof(LayerTaps)));
    env->allocs.push_back(t);
    env->env_taps = static_cast<LayerTaps*>(t);
    return 0;
}

// host copy of the clocks -> device (current buffer) + the taps they imply (no ring pending)
static int push_env_clocks(AoEnv* env, const std::vector<EnvClock>& clk) {
    const size_t n = (size_t)env->L * env->E;
    std::vector<LayerTaps> taps(n);
    for (int l = 0; l < env->L; ++l)
        for (int e = 0; e < env->E; ++e) {
            const EnvClock& c = clk[(size_t)l * env->E + e];
            LayerTaps& t = taps[(size_t)l * env->E + e];
            t = LayerTaps{};
            t.oy = c.org[0];
            t.ox = c.org[1];
            taps_from_buff(c.buff, t);
            t.weight = env->layer_weight[l];
        }
    AO_HIP(hipMemcpy(env->env_clk[env->clk_cur], clk.data(), n * sizeof(EnvClock), hipMemcpyHostToDevice));
    AO_HIP(hipMemcpy(env->env_taps, taps.data(), n * sizeof(LayerTaps), hipMemcpyHostToDevice));
    return 0;
}

static int pull_env_clocks(AoEnv* env, std::vector<EnvClock>& clk) {
    clk.resize((size_t)env->L * env->E);
    AO_HIP(hipMemcpy(clk.data(), env->env_clk[env->clk_cur], clk.size() * sizeof(EnvClock), hipMemcpyDeviceToHost));
    return 0;
}

int aoenv_set_wind_env(AoEnv* env, const double* h_ratio, int reset_buff, void* stream) {
    AO_CHECK_ENV(env);
    if (!h_ratio) return fail("null ratio");
    const size_t n = (size_t)env->L * env->E;
    for (size_t i = 0; i < 2 * n; ++i)
        if (!(std::fabs(h_ratio[i]) < 1.0)) return fail("per-env wind: |ratio| = %g px/frame, must be < 1 (an env extrudes at most one ring per step)", std::fabs(h_ratio[i]));
    hipStream_t st = static_cast<hipStream_t>(stream);
    AO_TRY(drop_lookaheads(env, st));
    AO_TRY(AO_DISPATCH(env, flush_rings, env, st));                // a deferred ring of the clocks as they were
    AO_HIP(hipStreamSynchronize(st));
    AO_TRY(alloc_env_clocks(env));
    std::vector<EnvClock> clk;
    if (env->per_env_wind) {
        AO_TRY(pull_env_clocks(env, clk));
    } else {                                                       // from the shared clock: every env starts where the shard is
        clk.assign(n, EnvClock{});
        for (int l = 0; l < env->L; ++l)
            for (int e = 0; e < env->E; ++e) {
                EnvClock& c = clk[(size_t)l * env->E + e];
                c.buff[0] = env->clk[l].buff[0];
                c.buff[1] = env->clk[l].buff[1];
                c.org[0] = env->org[l][0];
                c.org[1] = env->org[l][1];
            }
    }
    for (size_t i = 0; i < n; ++i) {
        clk[i].ratio[0] = h_ratio[2 * i];
        clk[i].ratio[1] = h_ratio[2 * i + 1];
        if (reset_buff) clk[i].buff[0] = clk[i].buff[1] = 0;
    }
    env->per_env_wind = true;
    return push_env_clocks(env, clk);
}

int aoenv_get_clock_env(AoEnv* env, double* h_clock) {
    AO_CHECK_ENV(env);
    if (!h_clock) return fail("null argument");
    if (!env->per_env_wind) return fail("the shard runs the shared clock (aoenv_set_wind): use aoenv_get_buff");
    AO_HIP(hipDeviceSynchronize());
    std::vector<EnvClock> clk;
    AO_TRY(pull_env_clocks(env, clk));
    for (size_t i = 0; i < clk.size(); ++i) {
        h_clock[4 * i] = clk[i].ratio[0];
        h_clock[4 * i + 1] = clk[i].ratio[1];
        h_clock[4 * i + 2] = clk[i].buff[0];
        h_clock[4 * i + 3] = clk[i].buff[1];
    }
    return 0;
}

int aoenv_set_clock_env(AoEnv* env, const double* h_clock) {
    AO_CHECK_ENV(env);
    if (!h_clock) return fail("null argument");
    if (!env->per_env_wind) return fail("the shard runs the shared clock (aoenv_set_wind): use aoenv_set_buff");
    AO_HIP(hipDeviceSynchronize());
    std::vector<EnvClock> clk;
    AO_TRY(pull_env_clocks(env, clk));                             // (keeps the origins)
    for (size_t i = 0; i < clk.size(); ++i) {
        for (int d = 0; d < 2; ++d) {
            if (!(std::fabs(h_clock[4 * i + d]) < 1.0)) return fail("per-env wind: |ratio| must be < 1 px/frame");
            if (!(std::fabs(h_clock[4 * i + 2 + d]) < 1.0)) return fail("|buff| must be < 1");
            clk[i].ratio[d] = h_clock[4 * i + d];
            clk[i].buff[d] = h_clock[4 * i + 2 + d];
        }
    }
    for (int l = 0; l < env->L; ++l) env->ring_pending[l] = 0;
    return push_env_clocks(env, clk);
}

// every env's clock back to the start of an episode: accumulators and origins zero (new screens / uploaded screens)
static int reset_env_clocks(AoEnv* env, bool reset_buff) {
    if (!env->per_env_wind) return 0;
    std::vector<EnvClock> clk;
    AO_TRY(pull_env_clocks(env, clk));
    for (auto& c : clk) {
        c.org[0] = c.org[1] = 0;
        if (reset_buff) c.buff[0] = c.buff[1] = 0;
    }
    return push_env_clocks(env, clk);
}

int aoenv_set_wind(AoEnv* env, const double* h_ratio, int reset_buff) {
    AO_CHECK_ENV(env);
    if (!h_ratio) return fail("null ratio");
    if (env->per_env_wind) {                                       // the shard keeps its per-env clocks: the same wind for every env
        std::vector<double> r((size_t)env->L * env->E * 2);
        for (int l = 0; l < env->L; ++l)
            for (int e = 0; e < env->E; ++e) {
                r[2 * ((size_t)l * env->E + e)] = h_ratio[2 * l];
                r[2 * ((size_t)l * env->E + e) + 1] = h_ratio[2 * l + 1];
            }
        // (this entry point has no stream argument: the caller may be stepping on a non-blocking stream, whose pending ring /
        //  clock work must be through before the clocks are pulled, changed and pushed back on the null stream)
        AO_HIP(hipDeviceSynchronize());
        return aoenv_set_wind_env(env, r.data(), reset_buff, nullptr);
    }
    for (int l = 0; l < env->L; ++l) {
        env->clk[l].ratio[0] = h_ratio[2 * l];
        env->clk[l].ratio[1] = h_ratio[2 * l + 1];
        if (reset_buff) env->clk[l].buff[0] = env->clk[l].buff[1] = 0;
    }
    return 0;
}

static int require_step_constants(AoEnv* env, bool atmosphere) {
    static const int base[] = {AOENV_C_PUPIL, AOENV_C_ACT_IDX, AOENV_C_WFS_AMP, AOENV_C_SH_SUBAP_IDX};
    for (int k : base)
        if (!env->have[k]) return fail("constant table %d has not been uploaded", k);
    if (env->c.dm_separable ? !(env->have[AOENV_C_DM_GX] && env->have[AOENV_C_DM_GY]) : !env->have[AOENV_C_DM_MODES])
        return fail("DM influence functions have not been uploaded");
    if (atmosphere && env->L > 0) {
        if (!env->have[AOENV_C_LAYER_WEIGHT]) return fail("atmosphere table %d has not been uploaded", (int)AOENV_C_LAYER_WEIGHT);
        for (int l = 0; l < env->L; ++l)
            if (!env->have_ab[l] || !env->have_in[l] || !env->have_out[l]) return fail("the ring tables of layer %d have not been uploaded", l);
    }
    return 0;
}

// ring RandomState seeding, first ring X = A.Z + B.xi, accumulators, atm.OPD: the part of generateNewPhaseScreen
// after the new interior is in mapShift (OOPAO/Atmosphere.py:579-592)
static int finish_new_screens(AoEnv* env, const uint32_t* h_ring_seeds, hipStream_t st) {
    const int E = env->E, L = env->L;
    for (int l = 0; l < L; ++l) env->ring_pending[l] = 0;          // a deferred ring of the old screens is moot
    AO_TRY(sync_lookaheads(env));
    std::vector<uint32_t> keys((size_t)E * kMtN);
    std::vector<int> pos((size_t)E, kMtN);
    for (int l = 0; l < L; ++l) {
        for (int e = 0; e < E; ++e) mt_seed(h_ring_seeds[(size_t)e * L + l], &keys[(size_t)e * kMtN]);
        AO_HIP(hipMemcpy(env->mt_cur[l], keys.data(), keys.size() * 4, hipMemcpyHostToDevice));
        AO_HIP(hipMemcpy(env->pos_cur[l], pos.data(), pos.size() * 4, hipMemcpyHostToDevice));
    }
    AO_HIP(hipStreamSynchronize(st));
    AO_TRY(reset_env_clocks(env, true));                           // per-env clocks: accumulators and origins of every env to zero
    const bool pe = env->per_env_wind;
    env->per_env_wind = false;                                     // the first ring is one extrusion of the whole shard, origin 0
    for (int l = 0; l < L; ++l) {
        env->clk[l].buff[0] = env->clk[l].buff[1] = 0;            // notDoneOnce (OOPAO/Atmosphere.py:586, 359-364)
        const int rc = AO_DISPATCH(env, extrude, env, l, 0, 0, false, st);   // (the callers reset the torus origin with the new interior)
        if (rc) { env->per_env_wind = pe; return rc; }
    }
    env->per_env_wind = pe;
    env->atm_user_defined = false;
    AO_TRY(AO_DISPATCH(env, run_phase, env, 1, 1, st));            // fill_phase_support + set_OPD + atm*tel
    return 0;
}

int aoenv_new_screens(AoEnv* env, const double* h_screens, const uint32_t* h_ring_seeds, void* stream) {
    AO_CHECK_ENV(env);
    if (env->L == 0) return fail("no atmosphere in this shard");
    AO_TRY(require_step_constants(env, true));
    if (!h_ring_seeds) return fail("null ring seeds");
    hipStream_t st = static_cast<hipStream_t>(stream);
    AO_HIP(hipStreamSynchronize(st));
    const int E = env->E, L = env->L;
    if (!h_screens) {
        if (env->per_env_wind) return fail("per-env clocks: the interior cannot be kept (every env has moved its own origin): hand the screens over");
        for (int l = 0; l < L; ++l)
            if (env->org[l][0] || env->org[l][1]) return fail("keeping the interior is only possible before the first shift");
    }
    for (int l = 0; l < L; ++l) env->org[l][0] = env->org[l][1] = 0;
    if (h_screens) {
        // mapShift[~outerMask] = phase  (OOPAO/Atmosphere.py:585); the ring is drawn below
        std::vector<char> host;
        size_t layer_base = 0;                                     // (layers with grids of their own: layer-major blocks)
        for (int l = 0; l < L; ++l) {
            const int N = env->Nl[l], S = env->Sl[l];
            host.assign((size_t)E * S * S * env->esz, 0);
            for (int e = 0; e < E; ++e) {
                const double* src = env->uniform ? h_screens + ((size_t)e * L + l) * N * N : h_screens + layer_base + (size_t)e * N * N;
                for (int r = 0; r < N; ++r)
                    for (int c = 0; c < N; ++c) {
                        const size_t o = (size_t)e * S * S + (size_t)(r + 1) * S + (c + 1);
                        if (env->esz == 4) reinterpret_cast<float*>(host.data())[o] = (float)src[(size_t)r * N + c];
                        else reinterpret_cast<double*>(host.data())[o] = src[(size_t)r * N + c];
                    }
            }
            AO_HIP(hipMemcpy(env->screen_ptr(0, l), host.data(), host.size(), hipMemcpyHostToDevice));
            layer_base += (size_t)E * N * N;
        }
    }
    return finish_new_screens(env, h_ring_seeds, st);
}

namespace {
struct TmpFree {                                   // scratch of one reset: released on every exit path
    std::vector<void*> p;
    ~TmpFree() { for (void* q : p) (void)hipFree(q); }
    int get(void** out, size_t bytes) {
        AO_HIP(hipMalloc(out, bytes));
        p.push_back(*out);
        return 0;
    }
};

// von Karman spectrum of OOPAO/phaseStats.py:206-207 (l0 = 1e-10 m: the inner-scale roll-off is 1 in float64)
double vk_psd(double f, double r0, double L0) {
    const double fm = 5.92 / 1e-10 / (2 * 3.14159265358979323846), f0 = 1.0 / L0;
    return 0.023 * std::pow(r0, -5.0 / 3) * std::exp(-((f / fm) * (f / fm))) / std::pow(f * f + f0 * f0, 11.0 / 6);
}
}  // namespace

int aoenv_new_screens_device(AoEnv* env, const uint32_t* h_screen_seeds, const uint32_t* h_ring_seeds, double r0,
                             double L0, double pixel_size, void* stream) {
    AO_CHECK_ENV(env);
    if (env->L == 0) return fail("no atmosphere in this shard");
    AO_TRY(require_step_constants(env, true));
    if (!h_screen_seeds || !h_ring_seeds) return fail("null seeds");
    if (!(r0 > 0) || !(L0 > 0) || !(pixel_size > 0)) return fail("r0, L0 and the pixel size must be positive");
    hipStream_t st = static_cast<hipStream_t>(stream);
    AO_HIP(hipStreamSynchronize(st));
    const int E = env->E, L = env->L;
    for (int l = 0; l < L; ++l) env->org[l][0] = env->org[l][1] = 0;
    const double pi = 3.14159265358979323846;
    TmpFree tmp;
    int N_built = -1, EC = 1;
    ScreenArgs sa{};
    double *d_amp = nullptr, *d_sub = nullptr, *d_tw = nullptr, *d_nrm = nullptr, *d_hi = nullptr;
    void* d_scr = nullptr;
    uint32_t* d_mt = nullptr;
    int* d_pos = nullptr;
    std::vector<uint32_t> keys;
    std::vector<int> pos;
    for (int l = 0; l < L; ++l) {
        const int N = env->Nl[l], S = env->Sl[l];
        const size_t N2 = (size_t)N * N;
        if (N != N_built) {                                        // tables of this grid size (every layer's when fov = 0)
            // frequency-grid amplitude sqrt(PSD) del_f (phaseStats.py:209-222) and the 3 x 4 sub-harmonic terms (:277-309)
            std::vector<double> amp(N2), sub(36), tw(2 * (size_t)N);
            const double del_f = 1.0 / (N * pixel_size);
            for (int y = 0; y < N; ++y)
                for (int x = 0; x < N; ++x) {
                    const double fx = (x - N / 2.0) * del_f, fy = (y - N / 2.0) * del_f;
                    amp[(size_t)y * N + x] = std::sqrt(vk_psd(std::sqrt(fx * fx + fy * fy), r0, L0)) * del_f;
                }
            amp[(size_t)(N / 2) * N + N / 2] = 0;
            const double D = N * pixel_size;
            for (int p = 1; p <= 3; ++p) {
                const double df = 1.0 / (std::pow(3.0, p) * D);
                for (int i = 0; i < 2; ++i)
                    for (int j = 0; j < 2; ++j) {
                        const double fx = (j - 1) * df, fy = (i - 1) * df;
                        double* t = &sub[3 * (4 * (p - 1) + 2 * i + j)];
                        t[0] = (i == 1 && j == 1) ? 0.0 : std::sqrt(vk_psd(std::sqrt(fx * fx + fy * fy), r0, L0)) * df;
                        t[1] = fx;
                        t[2] = fy;
                    }
            }
            for (int k = 0; k < N; ++k) { tw[2 * k] = std::cos(2 * pi * k / N); tw[2 * k + 1] = -std::sin(2 * pi * k / N); }
            sa = ScreenArgs{};
            AO_TRY(make_fft_plan(N, &sa.plan));
            sa.N = N;
            sa.delta = pixel_size;
            const size_t per_env = 40 * N2;                        // normals + complex scratch + real screen, float64
            EC = (int)std::min<size_t>((size_t)E, std::max<size_t>(1, ((size_t)1 << 30) / per_env));
            AO_TRY(tmp.get((void**)&d_amp, N2 * 8));
            AO_TRY(tmp.get((void**)&d_sub, 36 * 8));
            AO_TRY(tmp.get((void**)&d_tw, 2 * (size_t)N * 8));
            AO_TRY(tmp.get((void**)&d_nrm, (size_t)EC * 2 * N2 * 8));
            AO_TRY(tmp.get(&d_scr, (size_t)EC * N2 * 16));
            AO_TRY(tmp.get((void**)&d_hi, (size_t)EC * N2 * 8));
            AO_TRY(tmp.get((void**)&d_mt, (size_t)EC * kMtN * 4));
            AO_TRY(tmp.get((void**)&d_pos, (size_t)EC * 4));
            AO_HIP(hipMemcpy(d_amp, amp.data(), N2 * 8, hipMemcpyHostToDevice));
            AO_HIP(hipMemcpy(d_sub, sub.data(), 36 * 8, hipMemcpyHostToDevice));
            AO_HIP(hipMemcpy(d_tw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
            sa.amp = d_amp; sa.sub = d_sub; sa.tw = d_tw; sa.nrm = d_nrm; sa.hi = d_hi;
            sa.scratch = reinterpret_cast<cx<double>*>(d_scr);
            keys.assign((size_t)EC * kMtN, 0u);
            pos.assign((size_t)EC, kMtN);
            N_built = N;
        }
        for (int e0 = 0; e0 < E; e0 += EC) {
            const int ne = std::min(EC, E - e0);
            // the layer's own RandomState(seed + layer) draws normal(size=(N, N)) twice (real, imaginary parts)
            for (int e = 0; e < ne; ++e) mt_seed(h_screen_seeds[(size_t)(e0 + e) * L + l], &keys[(size_t)e * kMtN]);
            AO_HIP(hipMemcpyAsync(d_mt, keys.data(), (size_t)ne * kMtN * 4, hipMemcpyHostToDevice, st));
            AO_HIP(hipMemcpyAsync(d_pos, pos.data(), (size_t)ne * 4, hipMemcpyHostToDevice, st));
            AO_TRY(launch_mt_normal<double>(d_mt, d_pos, d_nrm, ne, (int)(2 * N2), 0, (int)(2 * N2), st));
            sa.n_env = ne;
            char* map = static_cast<char*>(env->screen_ptr(0, l)) + (size_t)e0 * S * S * env->esz;
            if (env->esz == 4) AO_TRY(launch_screen<float>(sa, reinterpret_cast<float*>(map), S, st));
            else AO_TRY(launch_screen<double>(sa, reinterpret_cast<double*>(map), S, st));
            AO_HIP(hipStreamSynchronize(st));                      // keys / pos are reused by the next chunk
        }
    }
    return finish_new_screens(env, h_ring_seeds, st);
}

int aoenv_set_atm_opd(AoEnv* env, const double* h_opd, void* stream) {
    AO_CHECK_ENV(env);
    hipStream_t st = static_cast<hipStream_t>(stream);
    AO_HIP(hipStreamSynchronize(st));
    const size_t n = (size_t)env->E * env->R * env->R;
    env->atm_user_defined = true;
    if (!h_opd) { AO_HIP(hipMemset(env->opd_atm, 0, n * env->esz)); return 0; }
    return upload_real(env, env->opd_atm, h_opd, n);
}

int aoenv_set_coefs(AoEnv* env, const double* h_coefs, void* stream) {
    AO_CHECK_ENV(env);
    hipStream_t st = static_cast<hipStream_t>(stream);
    AO_HIP(hipStreamSynchronize(st));
    const size_t n = (size_t)env->E * env->A;
    if (!h_coefs) AO_HIP(hipMemset(env->coefs, 0, n * env->esz));
    else AO_TRY(upload_real(env, env->coefs, h_coefs, n));
    return AO_DISPATCH(env, refresh_dense_dm, env, st);
}

int aoenv_measure(AoEnv* env, void* stream) {
    AO_CHECK_ENV(env);
    AO_TRY(require_step_constants(env, false));
    hipStream_t st = static_cast<hipStream_t>(stream);
    AO_TRY(AO_DISPATCH(env, run_phase, env, 1, 1, st));            // atmosphere re-derived from the screens at the current buff
    return AO_DISPATCH(env, run_wfs, env, st);
}

int aoenv_atm_update(AoEnv* env, void* stream) {
    AO_CHECK_ENV(env);
    AO_TRY(require_step_constants(env, true));
    hipStream_t st = static_cast<hipStream_t>(stream);
    return AO_DISPATCH(env, atm_update_t, env, st);
}

int aoenv_reset_soft(AoEnv* env, void* d_obs, void* stream) {
    AO_CHECK_ENV(env);
    if (!d_obs) return fail("null obs");
    if (!env->have[AOENV_C_RECON]) return fail("the reconstructor has not been uploaded");
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (env->c.dtype == AOENV_F32)
        return run_recon<float>(env, nullptr, static_cast<float*>(d_obs), nullptr, nullptr, -1, 0, 0.0, st);
    return run_recon<double>(env, nullptr, static_cast<double*>(d_obs), nullptr, nullptr, -1, 0, 0.0, st);
}

int aoenv_step(AoEnv* env, int i, const void* d_action, void* d_obs, void* d_frame, void* d_reward, void* d_strehl,
               void* stream) {
    AO_CHECK_ENV(env);
    if (!d_action || !d_obs) return fail("aoenv_step: null action / obs");
    if (i < 0 || i >= env->c.n_loop) return fail("frame index %d outside [0, n_loop=%d)", i, env->c.n_loop);
    AO_TRY(require_step_constants(env, true));
    if (!env->have[AOENV_C_RECON]) return fail("the reconstructor has not been uploaded");
    hipStream_t st = static_cast<hipStream_t>(stream);
    return AO_DISPATCH(env, step_t, env, i, d_action, d_obs, d_frame, d_reward, d_strehl, 0.0, st);
}

int aoenv_run_integrator(AoEnv* env, int i0, int n_steps, double gain, void* d_obs, void* d_frame, void* d_reward,
                         void* d_strehl, void* stream) {
    AO_CHECK_ENV(env);
    if (!d_obs) return fail("aoenv_run_integrator: null obs");
    if (gain == 0) return fail("aoenv_run_integrator: gain must be non-zero");
    if (i0 < 0 || n_steps < 0 || i0 + n_steps > env->c.n_loop) return fail("frames [%d, %d) outside [0, n_loop=%d)", i0, i0 + n_steps, env->c.n_loop);
    AO_TRY(require_step_constants(env, true));
    if (!env->have[AOENV_C_RECON]) return fail("the reconstructor has not been uploaded");
    hipStream_t st = static_cast<hipStream_t>(stream);
    for (int k = 0; k < n_steps; ++k)
        AO_TRY(AO_DISPATCH(env, step_t, env, i0 + k, d_obs /*unused*/, d_obs, k == n_steps - 1 ? d_frame : nullptr, d_reward,
                           d_strehl, gain, st));
    return 0;
}

extern "C++" {
template <typename T>
static int compute_psf_t(AoEnv* env, int zp, void* d_psf, hipStream_t st) {
    // the reference runs the transform at oversampling 2 for every even image size and sum-bins |.|^2 2 x 2 (Telescope.py:303-305)
    const int R = env->R, N = 2 * zp * R;
    const size_t R2 = (size_t)R * R;
    // scratch of one call: amplitude pupil * sqrt(src.fluxMap) (= the SH field amplitude; the Pyramid's is per modulation
    // point), twiddles, first-pass output
    TmpFree tmp;
    T *d_amp, *d_tw;
    void* d_t1;
    AO_TRY(tmp.get((void**)&d_amp, R2 * sizeof(T)));
    AO_TRY(tmp.get((void**)&d_tw, (size_t)2 * N * sizeof(T)));
    AO_TRY(tmp.get(&d_t1, (size_t)env->E * R * N * 2 * sizeof(T)));
    if (env->h_amp.size() != R2) return fail("the WFS amplitude has not been uploaded");
    const double scale = env->c.wfs_type == AOENV_WFS_PYRAMID ? std::sqrt((double)env->c.pyr_n_theta) : 1.0;
    std::vector<T> amp(R2), tw(2 * (size_t)N);
    for (size_t q = 0; q < R2; ++q) amp[q] = (T)(env->h_amp[q] * scale);
    const double pi = 3.14159265358979323846;
    for (int k = 0; k < N; ++k) { tw[2 * k] = (T)std::cos(2 * pi * k / N); tw[2 * k + 1] = (T)(-std::sin(2 * pi * k / N)); }
    AO_HIP(hipMemcpyAsync(d_amp, amp.data(), R2 * sizeof(T), hipMemcpyHostToDevice, st));
    AO_HIP(hipMemcpyAsync(d_tw, tw.data(), tw.size() * sizeof(T), hipMemcpyHostToDevice, st));
    PyrArgs<T> pa{};
    AO_TRY(make_fft_plan(N, &pa.plan));
    pa.phase = env->as<T>(env->phase);
    pa.amp = d_amp;
    pa.tt = nullptr;
    pa.tw = d_tw;
    pa.t1 = reinterpret_cast<cx<T>*>(d_t1);
    pa.R = R;
    pa.N = N;
    pa.off = N / 2 - R / 2;                                        // pad_width = (N - R) / 2
    pa.phasor_mult = 1;                                            // exp(-i pi / N (x + y)): even image sizes (Telescope.py:316)
    pa.n_env = env->E;
    AO_TRY(launch_psf<T>(pa, static_cast<T*>(d_psf), st));
    AO_HIP(hipStreamSynchronize(st));                              // the scratch is released on return
    return 0;
}
}  // extern "C++"

int aoenv_compute_psf(AoEnv* env, int zero_padding, void* d_psf, void* stream) {
    AO_CHECK_ENV(env);
    if (!d_psf) return fail("null psf");
    if (zero_padding < 1 || zero_padding > 8) return fail("zeroPaddingFactor %d outside [1, 8]", zero_padding);
    if ((zero_padding * env->R) % 2) return fail("odd image sizes are not built");
    if (2 * zero_padding * env->R > 8192) return fail("PSF transform length %d too long", 2 * zero_padding * env->R);
    hipStream_t st = static_cast<hipStream_t>(stream);
    return env->c.dtype == AOENV_F32 ? compute_psf_t<float>(env, zero_padding, d_psf, st)
                                     : compute_psf_t<double>(env, zero_padding, d_psf, st);
}

int aoenv_set_detector(AoEnv* env, const AoDetector* cfg, void* stream) {
    AO_CHECK_ENV(env);
    hipStream_t st = static_cast<hipStream_t>(stream);
    DetectorCfg d{};
    if (cfg) {
        if (cfg->bits < 0 || cfg->bits > 24) return fail("detector: bits %d outside [0, 24]", cfg->bits);
        if (cfg->bits > 0 && !(cfg->fwc > 0)) return fail("detector: the ADC needs a full-well capacity (FWC = None with bits set is not built)");
        if (!(cfg->qe > 0) || !(cfg->gain > 0) || cfg->dark_electrons < 0 || cfg->readout_noise < 0 || cfg->fwc < 0)
            return fail("detector: QE and gain must be positive, dark current / read-out noise / FWC non-negative");
        d.active = 1;
        d.photon_noise = cfg->photon_noise != 0;
        d.bits = cfg->bits;
        d.emccd = cfg->emccd != 0;
        d.qe = (float)cfg->qe;
        d.dark_e = (float)cfg->dark_electrons;
        d.fwc = (float)cfg->fwc;
        d.gain = (float)cfg->gain;
        d.readout_noise = (float)cfg->readout_noise;
        d.seed_lo = (uint32_t)(cfg->seed & 0xffffffffu);
        d.seed_hi = (uint32_t)(cfg->seed >> 32);
        d.env_offset = (uint32_t)cfg->env_index_offset;
        // the noise streams keep counting frames across camera changes: only a new seed starts them again
        d.frame_counter = (env->det_seeded && env->det.seed_lo == d.seed_lo && env->det.seed_hi == d.seed_hi) ? env->det.frame_counter : 0;
        // identity settings are the ideal camera: keep the fast paths
        if (!d.photon_noise && d.bits == 0 && d.qe == 1.f && d.dark_e == 0.f && d.fwc == 0.f && d.gain == 1.f && d.readout_noise == 0.f)
            d.active = 0;
    }
    if (env->det.active)                                           // no stale noise outside the valid lenslets (the new camera may not write there)
        AO_HIP(hipMemsetAsync(env->frame, 0, (size_t)env->E * env->c.cam_res * env->c.cam_res * env->esz, st));
    if (!cfg) {                                                    // ideal detector: the stream position and its seed are kept
        d.seed_lo = env->det.seed_lo; d.seed_hi = env->det.seed_hi; d.frame_counter = env->det.frame_counter;
        d.env_offset = env->det.env_offset;
    } else {
        env->det_seeded = true;
    }
    env->det = d;
    return 0;
}

int aoenv_set_return_accumulator(AoEnv* env, void* d_return) {
    AO_CHECK_ENV(env);
    env->ret_acc = d_return;
    return 0;
}

int aoenv_buffer(AoEnv* env, int which, void** d_ptr, size_t* bytes) {
    AO_CHECK_ENV(env);
    BufInfo b{};
    AO_TRY(buf_info(env, which, &b));
    if (which == AOENV_B_SCREEN || which == AOENV_B_MT_STATE || which == AOENV_B_COUNTERS)
        return fail("buffer %d has no flat device image (tori / host-side state): use aoenv_download / aoenv_upload_state", which);
    if (d_ptr) *d_ptr = b.ptr;
    if (bytes) *bytes = b.bytes;
    return 0;
}

int aoenv_download(AoEnv* env, int which, void* h_dst, size_t bytes, void* stream) {
    AO_CHECK_ENV(env);
    BufInfo b{};
    AO_TRY(buf_info(env, which, &b));
    if (bytes != b.bytes) return fail("aoenv_download(%d): got %zu bytes, expected %zu", which, bytes, b.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    AO_HIP(hipStreamSynchronize(st));
    if (which == AOENV_B_SCREEN) {
        AO_TRY(AO_DISPATCH(env, flush_rings, env, st));
        AO_HIP(hipStreamSynchronize(st));
        // the device keeps every screen as a torus: hand back the logical layer.mapShift
        const size_t z = env->esz;
        std::vector<char> tmp;
        std::vector<EnvClock> clk_h;
        if (env->per_env_wind) AO_TRY(pull_env_clocks(env, clk_h));
        for (int l = 0; l < env->L; ++l) {
            const int S = env->Sl[l];
            const size_t per = (size_t)env->E * S * S * z;
            tmp.resize(per);
            AO_HIP(hipMemcpy(tmp.data(), env->screen_ptr(0, l), per, hipMemcpyDeviceToHost));
            char* dst = static_cast<char*>(h_dst) + env->scr_off[l] * z;
            for (int e = 0; e < env->E; ++e) {
                const int oy = env->per_env_wind ? clk_h[(size_t)l * env->E + e].org[0] : env->org[l][0];
                const int ox = env->per_env_wind ? clk_h[(size_t)l * env->E + e].org[1] : env->org[l][1];
                for (int r = 0; r < S; ++r) {
                    const char* srow = tmp.data() + ((size_t)e * S * S + (size_t)((r + oy) % S) * S) * z;
                    char* drow = dst + ((size_t)e * S * S + (size_t)r * S) * z;
                    std::memcpy(drow, srow + (size_t)ox * z, (size_t)(S - ox) * z);
                    std::memcpy(drow + (size_t)(S - ox) * z, srow, (size_t)ox * z);
                }
            }
        }
        return 0;
    }
    if (which == AOENV_B_MT_STATE) {
        const size_t n = (size_t)env->L * env->E;
        std::vector<uint32_t> st_(n * kMtN);
        std::vector<int> pos(n);
        for (int l = 0; l < env->L; ++l) {                         // the committed copy of every layer (a look-ahead writes the other)
            AO_HIP(hipMemcpy(&st_[(size_t)l * env->E * kMtN], env->mt_cur[l], (size_t)env->E * kMtN * 4, hipMemcpyDeviceToHost));
            AO_HIP(hipMemcpy(&pos[(size_t)l * env->E], env->pos_cur[l], (size_t)env->E * 4, hipMemcpyDeviceToHost));
        }
        uint32_t* out = static_cast<uint32_t*>(h_dst);
        for (size_t i = 0; i < n; ++i) {
            std::memcpy(out + i * (kMtN + 1), &st_[i * kMtN], kMtN * 4);
            out[i * (kMtN + 1) + kMtN] = (uint32_t)pos[i];
        }
        return 0;
    }
    if (which == AOENV_B_COUNTERS) {
        uint32_t* out = static_cast<uint32_t*>(h_dst);
        out[0] = env->det.frame_counter; out[1] = out[2] = out[3] = 0;
        return 0;
    }
    if (which == AOENV_B_OPD_ATM && env->L > 0 && !env->atm_user_defined && !env->store_opd_atm) {
        // not written by the step kernels unless AOENV_OPT_STORE_ATM_OPD: re-derive it from the screens now
        // (same kernel, same sampling constants; the residual phase it rewrites is identical)
        AO_TRY(AO_DISPATCH(env, run_phase, env, 1, 1, st, 0));
        AO_HIP(hipStreamSynchronize(st));
    }
    AO_HIP(hipMemcpy(h_dst, b.ptr, bytes, hipMemcpyDeviceToHost));
    return 0;
}

int aoenv_upload_state(AoEnv* env, int which, const void* h_src, size_t bytes, void* stream) {
    AO_CHECK_ENV(env);
    BufInfo b{};
    AO_TRY(buf_info(env, which, &b));
    if (bytes != b.bytes) return fail("aoenv_upload_state(%d): got %zu bytes, expected %zu", which, bytes, b.bytes);
    hipStream_t st = static_cast<hipStream_t>(stream);
    AO_HIP(hipStreamSynchronize(st));
    if (which == AOENV_B_SCREEN) {
        // logical layer.mapShift of every env: the tori restart at origin 0; the clip range is re-derived by its next consumer
        AO_TRY(sync_lookaheads(env));
        for (int l = 0; l < env->L; ++l) {
            const size_t per = (size_t)env->E * env->Sl[l] * env->Sl[l] * env->esz;
            AO_HIP(hipMemcpy(env->screen_ptr(0, l), static_cast<const char*>(h_src) + env->scr_off[l] * env->esz, per, hipMemcpyHostToDevice));
            env->org[l][0] = env->org[l][1] = 0;
            env->ring_pending[l] = 0;
            env->minmax_dirty[l] = true;
        }
        env->atm_user_defined = false;
        return reset_env_clocks(env, false);                       // per-env clocks: origins to zero, accumulators kept
    }
    if (which == AOENV_B_MT_STATE) {
        const size_t n = (size_t)env->L * env->E;
        std::vector<uint32_t> st_(n * kMtN);
        std::vector<int> pos(n);
        const uint32_t* in = static_cast<const uint32_t*>(h_src);
        for (size_t i = 0; i < n; ++i) {
            std::memcpy(&st_[i * kMtN], in + i * (kMtN + 1), kMtN * 4);
            pos[i] = (int)in[i * (kMtN + 1) + kMtN];
            if (pos[i] < 0 || pos[i] > kMtN || pos[i] % 4) return fail("MT19937 position %d is not a multiple of 4 in [0, 624]", pos[i]);
        }
        AO_TRY(sync_lookaheads(env));
        for (int l = 0; l < env->L; ++l) {
            AO_HIP(hipMemcpy(env->mt_cur[l], &st_[(size_t)l * env->E * kMtN], (size_t)env->E * kMtN * 4, hipMemcpyHostToDevice));
            AO_HIP(hipMemcpy(env->pos_cur[l], &pos[(size_t)l * env->E], (size_t)env->E * 4, hipMemcpyHostToDevice));
        }
        return 0;
    }
    if (which == AOENV_B_COUNTERS) {
        env->det.frame_counter = static_cast<const uint32_t*>(h_src)[0];
        return 0;
    }
    AO_HIP(hipMemcpy(b.ptr, h_src, bytes, hipMemcpyHostToDevice));
    if (which == AOENV_B_COEFS) return AO_DISPATCH(env, refresh_dense_dm, env, st);
    return 0;
}

int aoenv_get_buff(AoEnv* env, double* h_buff) {
    if (!env || !h_buff) return fail("null argument");
    if (env->per_env_wind) return fail("the shard runs per-env clocks (aoenv_set_wind_env): use aoenv_get_clock_env");
    for (int l = 0; l < env->L; ++l) { h_buff[2 * l] = env->clk[l].buff[0]; h_buff[2 * l + 1] = env->clk[l].buff[1]; }
    return 0;
}

int aoenv_set_buff(AoEnv* env, const double* h_buff) {
    if (!env || !h_buff) return fail("null argument");
    if (env->per_env_wind) return fail("the shard runs per-env clocks (aoenv_set_wind_env): use aoenv_set_clock_env");
    for (int l = 0; l < env->L; ++l) {
        if (std::fabs(h_buff[2 * l]) >= 1 || std::fabs(h_buff[2 * l + 1]) >= 1) return fail("|buff| must be < 1");
        env->clk[l].buff[0] = h_buff[2 * l]; env->clk[l].buff[1] = h_buff[2 * l + 1];
    }
    return 0;
}

int aoenv_fused_step_active(AoEnv* env) {
    if (!env) return 0;
    return env->c.dtype == AOENV_F32 && fused_step_ok<float>(env) ? 1 : 0;
}

int aoenv_set_option(AoEnv* env, int option, int value) {
    AO_CHECK_ENV(env);
    switch (option) {
        case AOENV_OPT_FAST_WFS: env->use_fast_wfs = value != 0; return 0;
        case AOENV_OPT_MFMA_GEMM: env->use_mfma = value != 0; return 0;
        case AOENV_OPT_FAST_TRIG: env->use_fast_trig = value != 0; return 0;
        case AOENV_OPT_STORE_ATM_OPD: env->store_opd_atm = value != 0; return 0;
        case AOENV_OPT_FUSED_TAIL: env->use_fused_tail = value != 0; return 0;
        case AOENV_OPT_FUSED_STEP: env->use_fused_step = value != 0; return 0;
        case AOENV_OPT_DEFER_RING: env->defer_ring = value != 0; return 0;
        case AOENV_OPT_FACTORED_RECON: env->use_factored_recon = value != 0; return 0;
        case AOENV_OPT_RING_LOOKAHEAD:
            if (!value) AO_TRY(sync_lookaheads(env));
            env->use_lookahead = value != 0;
            return 0;
        case AOENV_OPT_COEFS_IMAGE:
            if (value) AO_TRY(alloc_dm_rows(env));
            env->use_coefs_img = value != 0;
            return 0;
        case 99: env->debug_ablate = value; return 0;
        default: return fail("unknown option %d", option);
    }
}

int aoenv_profile(AoEnv* env, int enable) {
    AO_CHECK_ENV(env);
    for (auto& e : env->prof_ev) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    env->prof_ev.clear();
    env->prof_on = enable != 0;
    return 0;
}

int aoenv_profile_read(AoEnv* env, double* h_ms, int32_t* h_count, void* stream) {
    AO_CHECK_ENV(env);
    if (!h_ms || !h_count) return fail("null argument");
    AO_HIP(hipStreamSynchronize(static_cast<hipStream_t>(stream)));
    for (int k = 0; k < AOENV_K_COUNT; ++k) { h_ms[k] = 0; h_count[k] = 0; }
    for (auto& e : env->prof_ev) {
        float ms = 0;
        AO_HIP(hipEventElapsedTime(&ms, e.a, e.b));
        h_ms[e.stage] += ms;
        h_count[e.stage] += 1;
    }
    return 0;
}

int aoenv_test_normal(int device, uint32_t seed, int n, int n_calls, double* h_out) {
    if (n < 2 || n % 2 || n_calls < 1 || !h_out) return fail("aoenv_test_normal: bad arguments");
    DeviceGuard ao_device_guard(device);
    if (!ao_device_guard.ok) return fail("hipSetDevice(%d) failed", device);
    uint32_t* st = nullptr; int* pos = nullptr; double* zx = nullptr;
    std::vector<uint32_t> key(kMtN);
    mt_seed(seed, key.data());
    int p0 = kMtN;
    AO_HIP(hipMalloc((void**)&st, kMtN * 4));
    AO_HIP(hipMalloc((void**)&pos, 4));
    AO_HIP(hipMalloc((void**)&zx, (size_t)n * 8));
    AO_HIP(hipMemcpy(st, key.data(), kMtN * 4, hipMemcpyHostToDevice));
    AO_HIP(hipMemcpy(pos, &p0, 4, hipMemcpyHostToDevice));
    int rc = 0;
    for (int c = 0; c < n_calls && !rc; ++c) {
        rc = launch_mt_normal<double>(st, pos, zx, 1, n, 0, n, nullptr);
        if (!rc && hipMemcpy(h_out + (size_t)c * n, zx, (size_t)n * 8, hipMemcpyDeviceToHost) != hipSuccess) rc = fail("copy back failed");
    }
    (void)hipFree(st); (void)hipFree(pos); (void)hipFree(zx);
    return rc;
}

}  // extern "C"
