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
    int dy, dx;            // floor(-buff_y), floor(-buff_x)  in {-1, 0}
    double wy[4], wx[4];   // Catmull-Rom weights of the taps at floor-1 .. floor+2
    double weight;         // sqrt(fractionalR0)
};

struct PhaseArgs {
    const void* screen[kMaxLayer];   // current mapShift of each layer, [n_env][(N+2)^2]
    const void* minmax[kMaxLayer];   // [n_env][2] min / max of each mapShift (warp output clip)
    LayerTaps taps[kMaxLayer];
    int n_layer;
    int S;                           // N + 2
    int foot;                        // offset of the R x R pupil footprint inside the (N+2)^2 screen
    int update_atm;                  // 1: recompute opd_atm from the screens; 0: keep the buffer
    int telemetry_index;             // >= 0: write total[i], residual[i]
};

}  // namespace ao

// ---- device entry points implemented in the kernel translation units -----------------------------
namespace ao {

struct Env;  // host object, env.hpp

template <typename T>
int launch_shift_gather(const T* old_map, T* new_map, T* zx, const int* inner_idx, int n_env, int S, int n_inner,
                        int K, int sx, int sy, int do_copy, hipStream_t st);
template <typename T>
int launch_mt_normal(uint32_t* mt_state, int* mt_pos, T* zx, int n_env, int K, int n_inner, int n_outer,
                     hipStream_t st);
template <typename T>
int launch_scatter_minmax(T* new_map, const T* X, const int* outer_idx, T* minmax, int n_env, int S, int n_outer,
                          hipStream_t st);
template <typename T>
int launch_gemm_nt(const T* X, const T* W, T* C, int M, int N, int K, int ldx, int ldw, int ldc, hipStream_t st);

template <typename T>
struct PhaseBuffers {
    T* opd_atm;            // [E][R*R]
    const T* coefs;        // [E][A]
    const T* dm_opd;       // [E][R*R] (dense DM path) or nullptr
    const T* gx;           // [R][nAct]
    const T* gy;           // [R][nAct]
    const int* act_idx;    // [A]
    const uint8_t* pupil;  // [R*R]
    T* phase;              // [E][R*R]
    T* scal;               // [E][4]: total_nm, residual_nm, strehl, (unused)
    T* total;              // [n_loop][E]
    T* residual;           // [n_loop][E]
    T* wfs_max;            // [E] zeroed here for the WFS kernels
};
template <typename T>
int launch_phase(const PhaseArgs& pa, const PhaseBuffers<T>& pb, int n_env, int R, int n_act, int n_valid_act,
                 int n_pupil, double atm_wavelength, double src_wavelength, hipStream_t st);

template <typename T>
struct ShConst {
    const T* amp;            // [R*R]
    const int* subap_idx;    // [nValid]
    const T* ref;            // [2*nValid]
    const T* tw;             // [n][2]   exp(-2 pi i k / n)
    const T* ph;             // [p][2]   phasor at padded coordinate a + lo
    T units;                 // slopes_units
    T threshold;
};
template <typename T>
int launch_sh_spots(const T* phase, const ShConst<T>& sc, T* frame, T* wfs_max, int n_env, int R, int n_subap,
                    int n_valid, hipStream_t st);
template <typename T>
int launch_sh_centroid(const T* frame, const T* wfs_max, const ShConst<T>& sc, T* signal, int n_env, int R,
                       int n_subap, int n_valid, int max_group, hipStream_t st);

template <typename T>
int launch_recon_finish(const T* v, const int* act_idx, const T* action, T* coefs, T* obs, T* reward, int n_env,
                        int n_act, int n_valid_act, double leak, int do_integrate, double gain_from_obs,
                        hipStream_t st);
template <typename T>
int launch_copy_scal(const T* scal, T* d_strehl, int n_env, hipStream_t st);
template <typename T>
int launch_convert_from_f64(const double* src, T* dst, size_t n, hipStream_t st);

}  // namespace ao
