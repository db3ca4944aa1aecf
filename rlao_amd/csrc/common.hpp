// Shared declarations of libaoenv (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include <string>
#include <vector>

#include "aoenv.h"

namespace ao {

constexpr int kWave = 64;          // CDNA wavefront
constexpr int kMaxLayer = 8;
constexpr int kMtN = 624;          // MT19937 state words
constexpr int kMaxSplits = 16;     // split-K slabs of the float32 MFMA GEMM
constexpr int kMaxModes = 512;     // rank of the factored reconstructor (fused tail)

int fail(const char* fmt, ...);
#define AO_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return ::ao::fail("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
    } while (0)
#define AO_TRY(call)            \
    do {                        \
        int r_ = (call);        \
        if (r_ != 0) return r_; \
    } while (0)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- per-layer sampling constants of one step (host -> phase kernel, by value) -------------------
// Sub-pixel translation of layer.mapShift by layer.buff (OOPAO/Atmosphere.py:406-407): the sampling
// point of output pixel (r, c) is (r - buff_y, c - buff_x); for a pure translation the integer offset
// and the 4 Catmull-Rom tap weights are the same for every pixel, so they are computed once on the host
// in float64 from the float64 accumulator (as the reference's float64 warp sees it).
struct LayerTaps {
    int oy, ox;            // origin of the torus the screen is stored as: logical (r, c) at ((r + oy) % S, (c + ox) % S)
    int dy, dx;            // floor(-buff_y), floor(-buff_x)  in {-1, 0}
    double wy[4], wx[4];   // Catmull-Rom weights of the taps at floor-1 .. floor+2
    double weight;         // sqrt(fractionalR0)
    int ring;              // per-env clocks only: 1 = this env's layer crossed a pixel this step (its ring is to be scattered)
    int pad_;
};

// ---- the atmosphere clock of one (env, layer): OOPAO/Atmosphere.py:350-407 ------------------------------------------------
// Shared by the host (one clock per layer for the whole shard, aoenv_set_wind) and the device (one clock per env and layer,
// aoenv_set_wind_env: k_ring_prepare_env): the same source, compiled without floating-point contraction, so that an env
// stepped by its own clock is bit-identical to a shard stepped by the host clock with that wind.
struct EnvClock {
    double ratio[2];       // pixels per frame (x, y), |ratio| < 1 for per-env clocks          Atmosphere.py:362-363
    double buff[2];        // sub-pixel accumulator                                             Atmosphere.py:392-404
    int org[2];            // torus origin (oy, ox)
    int pad_[2];
};

__host__ __device__ inline double clock_sgn(double v) { return (double)((v > 0) - (v < 0)); }

__host__ __device__ inline void catmull_rom(double x, double w[4]) {
#pragma clang fp contract(off)
    w[0] = 0.5 * (-x * x * x + 2 * x * x - x);
    w[1] = 0.5 * (3 * x * x * x - 5 * x * x + 2);
    w[2] = 0.5 * (-3 * x * x * x + 4 * x * x + x);
    w[3] = 0.5 * (x * x * x - x * x);
}

// the sub-pixel part of updateLayer: buff += frac(|ratio|) sign(ratio); a pixel is crossed where |buff| >= 1: *bx, *by in
// {-1, 0, 1} (the caller extrudes the ring and moves the origin); buff keeps its fractional part.
__host__ __device__ inline bool clock_subpixel(const double ratio[2], double buff[2], int* bx, int* by) {
#pragma clang fp contract(off)
    for (int d = 0; d < 2; ++d) buff[d] += fmod(fabs(ratio[d]), 1.0) * clock_sgn(ratio[d]);
    *bx = fabs(buff[0]) < 1 ? 0 : (int)clock_sgn(buff[0]);
    *by = fabs(buff[1]) < 1 ? 0 : (int)clock_sgn(buff[1]);
    for (int d = 0; d < 2; ++d) buff[d] = fmod(fabs(buff[d]), 1.0) * clock_sgn(buff[d]);
    return *bx != 0 || *by != 0;
}

// tap offsets and weights of the sub-pixel warp for the accumulator `buff` (OOPAO/Atmosphere.py:406-407)
__host__ __device__ inline void taps_from_buff(const double buff[2], LayerTaps& t) {
#pragma clang fp contract(off)
    const double fy = -buff[1], fx = -buff[0];
    const double ky = floor(fy), kx = floor(fx);
    t.dy = (int)ky;
    t.dx = (int)kx;
    catmull_rom(fy - ky, t.wy);
    catmull_rom(fx - kx, t.wx);
}

struct PhaseArgs {
    const void* screen[kMaxLayer];   // current mapShift of each layer, [n_env][(N+2)^2]
    const void* minmax[kMaxLayer];   // [n_env][2] min / max of each mapShift (warp output clip)
    LayerTaps taps[kMaxLayer];
    int n_layer;
    int S;                           // N + 2 of layer 0 (the fused step kernel: every layer on that grid)
    int foot;                        // offset of the R x R pupil footprint inside the (N+2)^2 screen, layer 0
    int S_l[kMaxLayer];              // ... of every layer: with fov != 0 a layer at altitude h has a grid of its own
    int foot_l[kMaxLayer];           //     (OOPAO/Atmosphere.py:216-232)
    int update_atm;                  // 1: atmosphere OPD from the screens; 0: from the opd_atm buffer (user-defined OPD)
    int store_atm;                   // 1: also write atm.OPD_no_pupil to the opd_atm buffer (state inspection)
    int store_phase;                 // 0: leave the residual phase, the telemetry sums and wfs_max untouched
    int minmax_dirty[kMaxLayer];     // fused step kernel: 1 = recompute the layer's min / max from the map and store it
    const LayerTaps* env_taps;       // per-env clocks (aoenv_set_wind_env): [n_layer][n_env] taps of THIS step, else null
    int n_env;                       // row length of env_taps
};

// the taps of (layer l, env e): the env's own when the shard runs per-env clocks, else the layer's
__device__ inline const LayerTaps& layer_taps(const PhaseArgs& pa, int l, int e) {
    return pa.env_taps ? pa.env_taps[(size_t)l * pa.n_env + e] : pa.taps[l];
}

}  // namespace ao

// ---- device entry points implemented in the kernel translation units -----------------------------
namespace ao {

struct Env;  // host object, env.hpp

template <typename T>
int launch_ring_prepare(const T* map, T* zx, const int* inner_idx, const uint32_t* mt_state, const int* mt_pos,
                        uint32_t* mt_state_out, int* mt_pos_out, int n_env, int S, int n_inner, int n_outer, int K, int sx, int sy,
                        int oy, int ox, hipStream_t st);
template <typename T>
int launch_mt_normal(uint32_t* mt_state, int* mt_pos, T* zx, int n_env, int K, int n_inner, int n_outer,
                     hipStream_t st);
template <typename T>
int launch_scatter_minmax(T* new_map, const T* X, const int* outer_idx, T* minmax, int n_env, int S, int n_outer,
                          int splits, int oy, int ox, int with_minmax, hipStream_t st, const LayerTaps* env_taps = nullptr);
template <typename T>
int launch_ring_prepare_env(const T* map, T* zx, const int* inner_idx, uint32_t* mt_state, int* mt_pos, const EnvClock* clk_in,
                            EnvClock* clk_out, LayerTaps* taps, double weight, int n_env, int S, int n_inner, int n_outer, int K,
                            hipStream_t st);
template <typename T>
int launch_minmax(const T* maps, T* minmax, int n_env, int S, hipStream_t st);
int gemm_splits(int M, int N, int K);
// the draw of a layer's next innovations, run beside the ring GEMM (k_ring_gemm_draw_ahead)
struct MtAhead {
    const uint32_t* mt_in;   // [n_env][624] committed stream
    const int* pos_in;
    uint32_t* mt_out;        // the stream after the draw (another buffer)
    int* pos_out;
    float* zx_out;           // [n_env][K]: the draw goes to columns n_inner ..
    int K, n_inner, n_outer, n_env;
};
int launch_ring_gemm_draw_ahead(const float* X, const float* W, float* Cpart, int M, int N, int K, int splits, const MtAhead& m,
                                hipStream_t st);
int launch_sum_slabs(float* x, size_t n, int slabs, hipStream_t st);     // slab 0 += slabs 1 .. (in that order), n floats per slab
int launch_gemm_nt_mfma(const float* X, const float* W, float* Cpart, int M, int N, int K, int ldx, int ldw, int splits,
                        hipStream_t st, int xsplits = 1, size_t xslab = 0);
template <typename T>
int launch_gemm_nt(const T* X, const T* W, T* C, int M, int N, int K, int ldx, int ldw, int ldc, hipStream_t st);

// MFMA 16x16x4 operand tables.  Lane (lc = lane & 15, lq = lane >> 4) of a wave working on the 16-row tile t needs
// g[16 t + lc][lq + 4 s] for every k step s: the table stores a lane's four consecutive steps as one float4 and the 64
// lanes' float4s of one (tile, step quad) contiguously, so every operand load is one fully coalesced 1 KB wave access
// (a row-major table makes the same load touch 64 cache lines, which was half of the ELT-size phase kernel).
__host__ __device__ inline size_t ga_index(int x, int k, int ga_stride) {
    const int q = k & 3, s = k >> 2;
    return ((((size_t)(x >> 4) * (ga_stride >> 2) + (s >> 2)) * 4 + q) * 16 + (x & 15)) * 4 + (s & 3);
}
template <typename T>
struct PhaseBuffers {
    T* opd_atm;            // [E][R*R]
    const T* coefs;        // [E][A]
    const T* coefs_img;    // [E][nAct^2] the same as actuator images (zero elsewhere), or nullptr: the kernel scatters coefs itself
    const T* dm_opd;       // [E][R*R] (dense DM path) or nullptr
    const T* gx;           // [R][nAct]
    const T* gy;           // [R][nAct]
    const T* gxt;          // [nActPad4][Rpad128] zero-padded transpose of gx (MFMA B operand), may be null
    const float* gxa;      // gx as float32 MFMA operands, ga_index() layout: [Rpad128 / 16][ga_stride / 4][64 lanes][4]
    const float* gya;      // the same for gy
    int ga_stride;         // k steps per lane (n_act / 4 rounded up to a multiple of 4)
    const float* s1a;      // [E] x the same layout: Gy C (k_dm_rows), or nullptr: the kernel forms its rows itself
    const int* act_idx;    // [A]
    const uint8_t* pupil;  // [R*R]
    T* phase;              // [E][R*R]
    double* part;          // [E][tiles][4] per-tile Sum / Sum^2 of the atmosphere and residual OPD over the pupil
    T* wfs_max;            // [E] zeroed here for the WFS kernels
};
int phase_tiles(int R, int n_act, size_t esz);
// coefs [E][A] -> actuator images [E][nAct^2] (large DMs: every tile workgroup of the phase kernel would otherwise repeat the scatter)
template <typename T>
int launch_coefs_image(const T* coefs, const int* act_idx, T* img, int n_env, int n_act, int n_valid_act, hipStream_t st);
// float32, separable DM: Gy C of every env in MFMA operand layout, once per step (see k_dm_rows)
int launch_dm_rows(const float* coefs, const int* act_idx, const float* gya, float* s1a, int n_env, int R, int n_act, int n_valid_act,
                   int ga_stride, hipStream_t st);
template <typename T>
struct KArgs {                 // kernel argument block of the phase kernels
    PhaseArgs pa;
    PhaseBuffers<T> pb;
    int R, n_act, n_valid_act, tx, rp, ablate;
    T atm_scale;   // lambda_atm / 2 pi
    T src_scale;   // 2 pi / lambda_src
};
template <typename T>
KArgs<T> make_phase_kargs(const PhaseArgs& pa, const PhaseBuffers<T>& pb, int R, int n_act, int n_valid_act,
                          double atm_wavelength, double src_wavelength);
template <typename T>
int launch_phase(const PhaseArgs& pa, const PhaseBuffers<T>& pb, int n_env, int R, int n_act, int n_valid_act,
                 double atm_wavelength, double src_wavelength, int use_mfma, hipStream_t st);

template <typename T>
struct ShConst {
    const T* amp;            // [R*R]
    const int* subap_idx;    // [nValid]
    const uint8_t* valid2d;  // [n_subap^2] 1 = valid lenslet (fast P = 6 path), may be null
    const T* ref;            // [2*nValid]
    const T* tw;             // [n][2]   exp(-2 pi i k / n)
    const T* ph;             // [p][2]   phasor at padded coordinate a + lo
    T units;                 // slopes_units
    T threshold;
    int fast_trig;           // 1: hardware sin/cos after Cody-Waite reduction (float32 shards only)
};
template <typename T>
int launch_sh_spots(const T* phase, const ShConst<T>& sc, T* frame, T* wfs_max, int n_env, int R, int n_subap,
                    int n_valid, hipStream_t st);
template <typename T>
int launch_sh_centroid(const T* frame, const T* wfs_max, const ShConst<T>& sc, T* signal, int n_env, int R,
                       int n_subap, int n_valid, int max_group, hipStream_t st);

// Step epilogue (one workgroup per env): reconstruction image, reward, integrator, telemetry scalars.
template <typename T>
struct FinishArgs {
    const T* v;              // [splits][E][A] split-K slabs of R.s
    int splits;
    const int* act_idx;      // [A]
    const T* action;         // [E][nAct^2] or null (gain_from_obs != 0)
    T* coefs;                // [E][A]   dm.coefs
    T* dm_prev;              // [E][A]   env.dm_prev: the integrator's state (MAIN/OOPAOEnv/OOPAOEnv.py:314, 508-509), which
                             //          `dm.coefs = ...` from outside does not touch
    T* obs;                  // [E][nAct^2]
    T* reward;               // [E] or null
    T* ret;                  // [E] or null: episode return accumulator, += reward every step
    T* strehl;               // [E] or null
    T* scal;                 // [E][4] total_nm, residual_nm, strehl
    T* total;                // [n_loop][E]
    T* residual;             // [n_loop][E]
    const double* part;      // [E][tiles][4]
    int n_tiles, n_pupil, telemetry_index;
    int n_act, n_valid_act, do_integrate;
    T leak, gain_from_obs;
    double src_scale;        // 2 pi / lambda_src
};
template <typename T>
int launch_recon_finish(const FinishArgs<T>& fa, int n_env, hipStream_t st);
// fused SH tail: centroid + low-rank R.s + epilogue in one launch; -1 if it does not fit in LDS
template <typename T>
int launch_sh_tail(const T* frame, const T* wfs_max, const ShConst<T>& sc, T* signal, const T* fac_m,
                   const T* fac_m2c_t, int n_modes, const FinishArgs<T>& fa, int n_env, int R, int n_subap, int n_valid,
                   int max_group, hipStream_t st);
// WFS camera model (detector.hpp)
struct DetectorCfg {           // by value into kernels
    int active;                // 0: ideal detector (identity)
    int photon_noise, bits, emccd;
    float qe, dark_e, fwc, gain, readout_noise;     // fwc <= 0: no full-well capacity
    uint32_t seed_lo, seed_hi;
    uint32_t frame_counter;    // incremented by the host for every measurement
    uint32_t env_offset;       // global index of env 0 of this shard
};

}  // namespace ao
#include "poisson_alias.hpp"
namespace ao {

// fused per-env step kernel (step_kernel.hip): float32, 6 px per lenslet, separable DM, factored reconstructor
struct StepArgs {
    KArgs<float> k;
    ShConst<float> sc;
    FinishArgs<float> fa;        // n_tiles must be 1: the kernel leaves its telemetry sums in part[e][0][:]
    float* frame;                // [E][R*R]
    float* signal;               // [E][2 nValid]
    float* wfs_max;              // [E]
    const float* fac_m;          // [K][nSig]
    const float* fac_m2c_t;      // [K][A]
    const short* slot_of;        // [nSub^2] lenslet -> index among the valid ones, -1 = not valid
    const float* amp_pupil;      // [R*R] WFS field amplitude inside the pupil, -1 outside (pupil and amp in one load)
    const float* gxa;            // [128][4][8] gx[x][q + 4 s] at [x][q][s], zero padded: MFMA operands as two 16-byte loads
    const float* gya;            // [128][4][8] gy[y][q + 4 s]
    DetectorCfg det;             // WFS camera (active = 0: ideal)
    PoissonAlias pa;             // its photon-noise tables (poisson_alias.hpp)
    // ring extrusion whose scatter was deferred to this kernel (add_row part 3, OOPAO/Atmosphere.py:309-310):
    const float* ring_x[kMaxLayer];   // [splits][E][n_outer] split-K slabs of X = A Z + B xi of the layer, null = nothing pending
    int ring_splits[kMaxLayer];
    const int* outer_idx;        // [n_outer] logical flat index of every ring pixel
    int n_outer;
    // ... and the Z of the layer's NEXT crossing, gathered here once the ring is in place (the screen does not change until then):
    float* next_zx[kMaxLayer];   // [E][zx_ld] operand of the next ring GEMM (columns 0 .. n_inner), null = not asked for
    int next_sx[kMaxLayer], next_sy[kMaxLayer];   // direction of that crossing
    const int* inner_idx;        // [n_inner]
    int n_inner, zx_ld;
    int n_modes, n_subap, n_valid, n_env;
};
int step_fused_supported(int R, int n_subap, int n_valid, int n_act, int n_modes);
int step_alias_capacity(int n_act);
int launch_env_step(const StepArgs& a, hipStream_t st);

template <typename T>
int launch_convert_from_f64(const double* src, T* dst, size_t n, hipStream_t st);

// ---- Pyramid WFS (pyr_kernels.hip) ---------------------------------------------------------------------
template <typename T> struct cx;
struct FftPlan {
    int n;            // transform length
    int np;           // LDS stride between the sequences of a workgroup: n made odd, so that accesses that run ACROSS the
                      // sequences (the column gathers / scatters of the 2-D transforms) fall in different banks
    int n_fac;        // number of stages
    int fac[12];      // radix of each stage, product = n
    unsigned magic_ns[12];   // floor(2^32 / ns) + 1 of each stage (ns = product of the earlier radices): x / ns for x < 2^16
    unsigned magic_m[12];    // the same for m = n / radix
};
int make_fft_plan(int n, FftPlan* pl);
template <typename T>
struct PyrArgs {
    const T* phase;        // [E][R*R]
    const T* amp;          // [R*R]   sqrt(flux / nTheta) * reflectivity
    const T* tt;           // [nTheta][R*R] modulation tip/tilt (float32-rounded in the reference) or null
    const T* mask;         // [N*N][2] exp(i m), complex64-rounded
    const T* tw;           // [N][2]
    cx<T>* t1;             // [E][chunk][R][N]
    cx<T>* t2;             // [E][chunk][N][N]
    T* frame;              // [E][cam*cam]
    FftPlan plan;
    int R, N, cam, off, centering, theta0, n_theta_chunk, n_env, seq_per_block;
    unsigned magic_seq;    // floor(2^32 / seq_per_block) + 1: i / seq_per_block for i < 2^16 without a division (set by the launchers)
    int phasor_mult;       // the pupil field is multiplied by exp(-i pi m (x + y) / N) on the padded grid: m = N + 1 (Pyramid with a
                           // centred mask, Pyramid.py:294), 1 (science PSF, Telescope.py:316), 0 (none)
    int generic_fft;       // diagnostic (aoenv_set_option 99): bit 512 the Stockham passes also where pyr528_kernels.hip applies, bit 1024
                           // its column blocks dealt round-robin over the XCDs
};
// nRes = 528 / 288 in float32: the passes on the register-resident 24 x 22 / 16 x 18 transforms (pyr528_kernels.hip)
int pyramid528_supported(const PyrArgs<float>& a);
int launch_pyramid528(const PyrArgs<float>& base, int n_theta, int chunk, hipStream_t st);
// science-path PSF (Telescope.computePSF): |FFT2 of the zero-padded pupil field|^2 / N^2, fftshifted, into psf [E][N][N]
template <typename T>
int launch_psf(const PyrArgs<T>& base, T* psf, hipStream_t st);
template <typename T>
struct PyrSlopeArgs {
    const T* frame;          // [E][cam*cam]
    const int* valid_idx;    // [nValid] r * nSub + c inside a quadrant
    const T* ref;            // [2*nValid] reference slopes at the valid pixels (x block, y block)
    T* signal;               // [E][2*nValid]
    int cam, n_sub, n_valid, q_lo, q_hi;   // quadrant origins (grabQuadrant)
    int norm_valid_mean;     // 0: frame.mean() ; 1: mean of I1+I2+I3+I4 over the valid pixels
    T units;
};
template <typename T>
int launch_pyramid(const PyrArgs<T>& base, int n_theta, int chunk, hipStream_t st);      // camera frame only
template <typename T>
int launch_pyramid_slopes(const PyrSlopeArgs<T>& sl, int n_env, hipStream_t st);

// ---- episode reset: von Karman screens on the device (screen_kernels.hip) ---------------------------------
struct ScreenArgs {
    const double* nrm;       // [E][2 N^2] normals of this layer: real parts then imaginary parts
    const double* amp;       // [N^2]      sqrt(PSD_phi) * del_f on the frequency grid (PSD[N/2][N/2] = 0)
    const double* sub;       // [12][3]    sub-harmonic terms (amplitude, fx, fy), order p = 1..3, (i, j) in {0,1}^2
    const double* tw;        // [N][2]
    cx<double>* scratch;     // [E][N][N]
    double* hi;              // [E][N][N]  high-frequency screen
    FftPlan plan;
    int N, n_env, seq_per_block;
    double delta;            // layer pixel size [m]
};
// writes layer.phase (rad @ 500 nm) of n_env envs into the interior of their (N+2)^2 mapShift
template <typename T>
int launch_screen(const ScreenArgs& base, T* map, int S, hipStream_t st);

}  // namespace ao
