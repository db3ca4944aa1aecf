// Atmosphere kernels: integer screen shift + conditioning-ring gather, MT19937/legacy-Gaussian
// innovations, outer-ring scatter.   Reference: OOPAO/Atmosphere.py:301-311 (add_row).
#include "common.hpp"
#include "ring_device.hpp"

namespace ao {

template <typename T>
__global__ void __launch_bounds__(256) k_mt_normal(uint32_t* __restrict__ mt_state, int* __restrict__ mt_pos,
                                                   T* __restrict__ zx, int K, int n_inner, int n_outer) {
    mt_normal_body<T>(mt_state, mt_pos, mt_state, mt_pos, zx, K, n_inner, n_outer, blockIdx.x);
}

// One launch for the two independent halves of the ring operand [Z | xi] of env blockIdx.y:
// blockIdx.x == 0 gathers Z, blockIdx.x == 1 draws the n_outer innovations xi of the layer's MT19937 stream.
template <typename T>
__global__ void __launch_bounds__(256) k_ring_prepare(const T* __restrict__ map, T* __restrict__ zx,
                                                      const int* __restrict__ inner_idx, const uint32_t* mt_state,
                                                      const int* mt_pos, uint32_t* mt_state_out, int* mt_pos_out, int S,
                                                      int n_inner, int n_outer, int K, int sx, int sy, int oy, int ox) {
    const int e = blockIdx.y;
    if (blockIdx.x == 0) gather_ring<T>(map, zx, inner_idx, S, n_inner, K, sx, sy, oy, ox, e, threadIdx.x, 256);
    else mt_normal_body<T>(mt_state, mt_pos, mt_state_out, mt_pos_out, zx, K, n_inner, n_outer, e);
}

template <typename T>
int launch_ring_prepare(const T* map, T* zx, const int* inner_idx, const uint32_t* mt_state, const int* mt_pos,
                        uint32_t* mt_state_out, int* mt_pos_out, int n_env, int S, int n_inner, int n_outer, int K, int sx, int sy,
                        int oy, int ox, hipStream_t st) {
    if (n_outer % 2) return fail("mt_normal: n_outer=%d must be even", n_outer);
    hipLaunchKernelGGL(k_ring_prepare<T>, dim3(2, n_env), dim3(256), 0, st, map, zx, inner_idx, mt_state, mt_pos, mt_state_out,
                       mt_pos_out, S, n_inner, n_outer, K, sx, sy, oy, ox);
    AO_HIP(hipGetLastError());
    return 0;
}

// Per-env clocks (aoenv_set_wind_env): the same launch also ADVANCES the clock of (env, layer) -- every env has its own wind, so
// which envs cross a pixel this step is decided here, on the device, with the host clock's arithmetic (clock_subpixel, common.hpp).
// Both workgroups of an env read clk_in and come to the same verdict; blockIdx.x == 0 writes the advanced clock to clk_out (the host
// swaps the two) and this step's taps; an env that crosses gathers its Z through its OLD origin and draws its innovations in place.
template <typename T>
__global__ void __launch_bounds__(256) k_ring_prepare_env(const T* __restrict__ map, T* __restrict__ zx,
                                                          const int* __restrict__ inner_idx, uint32_t* mt_state, int* mt_pos,
                                                          const EnvClock* __restrict__ clk_in, EnvClock* __restrict__ clk_out,
                                                          LayerTaps* __restrict__ taps, double weight, int S, int n_inner,
                                                          int n_outer, int K) {
    const int e = blockIdx.y;
    EnvClock c = clk_in[e];
    const int oy = c.org[0], ox = c.org[1];
    int bx, by;
    const bool cross = clock_subpixel(c.ratio, c.buff, &bx, &by);
    if (blockIdx.x == 0) {
        if (cross) gather_ring<T>(map, zx, inner_idx, S, n_inner, K, bx, by, oy, ox, e, threadIdx.x, 256);
        if (threadIdx.x == 0) {
            c.org[0] = ((oy - by) % S + S) % S;
            c.org[1] = ((ox - bx) % S + S) % S;
            clk_out[e] = c;
            LayerTaps t;
            t.oy = c.org[0];
            t.ox = c.org[1];
            taps_from_buff(c.buff, t);
            t.weight = weight;
            t.ring = cross ? 1 : 0;
            t.pad_ = 0;
            taps[e] = t;
        }
    } else if (cross) {
        mt_normal_body<T>(mt_state, mt_pos, mt_state, mt_pos, zx, K, n_inner, n_outer, e);
    }
}

template <typename T>
int launch_ring_prepare_env(const T* map, T* zx, const int* inner_idx, uint32_t* mt_state, int* mt_pos, const EnvClock* clk_in,
                            EnvClock* clk_out, LayerTaps* taps, double weight, int n_env, int S, int n_inner, int n_outer, int K,
                            hipStream_t st) {
    if (n_outer % 2) return fail("mt_normal: n_outer=%d must be even", n_outer);
    hipLaunchKernelGGL(k_ring_prepare_env<T>, dim3(2, n_env), dim3(256), 0, st, map, zx, inner_idx, mt_state, mt_pos, clk_in, clk_out,
                       taps, weight, S, n_inner, n_outer, K);
    AO_HIP(hipGetLastError());
    return 0;
}

template <typename T>
int launch_mt_normal(uint32_t* mt_state, int* mt_pos, T* zx, int n_env, int K, int n_inner, int n_outer,
                     hipStream_t st) {
    if (n_outer % 2) return fail("mt_normal: n_outer=%d must be even", n_outer);
    hipLaunchKernelGGL(k_mt_normal<T>, dim3(n_env), dim3(256), 0, st, mt_state, mt_pos, zx, K, n_inner, n_outer);
    AO_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// add_row, part 3: map_full[outerMask] = X through the NEW origin (+ min / max of the whole new map, which the
// sub-pixel warp clips its output to: skimage clip=True; with_minmax = 0 leaves that to the consumer -- the fused
// step kernel recomputes it from the map it reads anyway).  One workgroup per env.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__device__ inline void block_minmax(const T* __restrict__ map, T* __restrict__ minmax, int S, int e) {
    __shared__ T red_lo[16], red_hi[16];
    T lo = (T)3.0e38, hi = (T)-3.0e38;
    // the torus is a permutation: scan physically, 8 independent loads per lane in flight (ELT screens: 236 k pixels per env)
    for (int i0 = threadIdx.x; i0 < S * S; i0 += 8 * (int)blockDim.x) {
        T v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = i0 + q * (int)blockDim.x;
            v[q] = map[i < S * S ? i : i0];
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            lo = v[q] < lo ? v[q] : lo;
            hi = v[q] > hi ? v[q] : hi;
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const T ol = __shfl_down(lo, off), oh = __shfl_down(hi, off);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    const int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        red_lo[w] = lo;
        red_hi[w] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)blockDim.x / kWave; ++i) {
            lo = red_lo[i] < lo ? red_lo[i] : lo;
            hi = red_hi[i] > hi ? red_hi[i] : hi;
        }
        minmax[2 * e] = lo;
        minmax[2 * e + 1] = hi;
    }
}

template <typename T>
__global__ void __launch_bounds__(1024) k_scatter_minmax(T* __restrict__ new_map, const T* __restrict__ X,
                                                         const int* __restrict__ outer_idx, T* __restrict__ minmax,
                                                         int S, int n_outer, int splits, size_t slab, int oy, int ox,
                                                         int with_minmax, const LayerTaps* __restrict__ env_taps) {
    const int e = blockIdx.x;
    if (env_taps) {                                                // per-env clocks: only the envs that crossed, through their origin
        if (!env_taps[e].ring) return;
        oy = env_taps[e].oy;
        ox = env_taps[e].ox;
    }
    T* map = new_map + (size_t)e * S * S;
    const T* x = X + (size_t)e * n_outer;
    for (int k = threadIdx.x; k < n_outer; k += blockDim.x) {
        T v = x[k];
        for (int z = 1; z < splits; ++z) v += x[(size_t)z * slab + k];       // split-K slabs, fixed order
        const int idx = outer_idx[k];
        map[torus(idx / S, idx % S, oy, ox, S)] = v;
    }
    if (!with_minmax) return;
    __syncthreads();                                               // the ring is in place (this workgroup wrote it)
    block_minmax<T>(map, minmax, S, e);
}

template <typename T>
int launch_scatter_minmax(T* new_map, const T* X, const int* outer_idx, T* minmax, int n_env, int S, int n_outer,
                          int splits, int oy, int ox, int with_minmax, hipStream_t st, const LayerTaps* env_taps) {
    hipLaunchKernelGGL(k_scatter_minmax<T>, dim3(n_env), dim3(with_minmax ? 1024 : 512), 0, st, new_map, X, outer_idx, minmax, S,
                       n_outer, splits, (size_t)n_env * n_outer, oy, ox, with_minmax, env_taps);
    AO_HIP(hipGetLastError());
    return 0;
}

// min / max of every env's map on its own (a consumer other than the fused step kernel needs it after a lean scatter)
template <typename T>
__global__ void __launch_bounds__(1024) k_minmax(const T* __restrict__ maps, T* __restrict__ minmax, int S) {
    block_minmax<T>(maps + (size_t)blockIdx.x * S * S, minmax, S, blockIdx.x);
}

template <typename T>
int launch_minmax(const T* maps, T* minmax, int n_env, int S, hipStream_t st) {
    hipLaunchKernelGGL(k_minmax<T>, dim3(n_env), dim3(1024), 0, st, maps, minmax, S);
    AO_HIP(hipGetLastError());
    return 0;
}

#define INST(T)                                                                                                    \
    template int launch_ring_prepare<T>(const T*, T*, const int*, const uint32_t*, const int*, uint32_t*, int*, int, int, int, \
                                        int, int, int, int, int, int, hipStream_t);                                \
    template int launch_mt_normal<T>(uint32_t*, int*, T*, int, int, int, int, hipStream_t);                        \
    template int launch_scatter_minmax<T>(T*, const T*, const int*, T*, int, int, int, int, int, int, int,         \
                                          hipStream_t, const LayerTaps*);                                          \
    template int launch_ring_prepare_env<T>(const T*, T*, const int*, uint32_t*, int*, const EnvClock*, EnvClock*, LayerTaps*, \
                                            double, int, int, int, int, int, hipStream_t);                                                          \
    template int launch_minmax<T>(const T*, T*, int, int, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace ao
