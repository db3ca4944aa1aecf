// Atmosphere kernels: integer screen shift + conditioning-ring gather, MT19937/legacy-Gaussian
// innovations, outer-ring scatter.   Reference: OOPAO/Atmosphere.py:301-311 (add_row).
#include "common.hpp"

namespace ao {

// ---------------------------------------------------------------------------------------------------
// add_row, part 1: onePixelShiftedPhaseScreen = warp(map_full, translate(sx, sy))[1:-1, 1:-1]  and
// Z = shifted[innerMask].  A one-pixel translation through the cubic interpolator returns the source
// pixel exactly: new[r][c] = old[r - sy][c - sx] on the N x N interior, and no interior pixel reads
// outside the (N+2)^2 map.  The screen is therefore kept as a TORUS: logical pixel (r, c) lives at
// physical ((r + oy) mod S, (c + ox) mod S), and the shift only moves the origin (oy, ox) -> (oy - sy,
// ox - sx); the old border row / column that wraps around becomes the new border, which the ring
// extrusion overwrites anyway.  No pixel is copied (the copy was as much HBM traffic as the step itself).
// Here only Z = old_logical[r_k - sy][c_k - sx] is gathered, through the OLD origin.
// ---------------------------------------------------------------------------------------------------
__device__ inline int torus(int r, int c, int oy, int ox, int S) {
    int pr = r + oy, pc = c + ox;
    pr = pr >= S ? pr - S : pr;
    pc = pc >= S ? pc - S : pc;
    return pr * S + pc;
}

template <typename T>
__device__ inline void gather_ring(const T* __restrict__ map, T* __restrict__ zx, const int* __restrict__ inner_idx, int S,
                                   int n_inner, int K, int sx, int sy, int oy, int ox, int e, int t0, int nt) {
    const T* src = map + (size_t)e * S * S;
    for (int k = t0; k < n_inner; k += nt) {
        const int idx = inner_idx[k];
        const int r = idx / S - sy, c = idx % S - sx;            // interior pixel: 0 <= r, c < S
        zx[(size_t)e * K + k] = src[torus(r, c, oy, ox, S)];
    }
}

// ---------------------------------------------------------------------------------------------------
// layer.randomState.normal(size=n_outer)  (OOPAO/Atmosphere.py:308): NumPy's legacy generator =
// MT19937 + polar Box-Muller with a cached second deviate.  One workgroup per stream (env, layer).
//   * 53-bit double = ((w0 >> 5) * 2^26 + (w1 >> 6)) / 2^53 ; one polar attempt consumes 4 words
//   * attempt accepted iff 0 < r2 < 1; the call returns f*x2 first, then the cached f*x1
//   * n_outer = 4N+4 is even, so every call leaves the cache empty and the word position a multiple
//     of 4; 624 = 4*156, so an attempt never straddles a state regeneration ("twist").
// Parallel form: all (624-pos)/4 attempts of the current state block are evaluated at once, an
// exclusive scan of the accept flags assigns output slots, and the first attempt that completes the
// request decides how many words are consumed -- the stream position stays bit-identical to NumPy's.
// ---------------------------------------------------------------------------------------------------
__device__ inline uint32_t mt_temper(uint32_t y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

__device__ inline uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t far) {
    uint32_t y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

// regenerate the 624-word state in LDS; three dependency-free phases + the wrap-around word
__device__ inline void mt_twist(uint32_t* s) {
    const int t = threadIdx.x;
    uint32_t v = 0;
    if (t < 227) v = mt_mix(s[t], s[t + 1], s[t + 397]);
    __syncthreads();
    if (t < 227) s[t] = v;
    __syncthreads();
    if (t < 227) v = mt_mix(s[t + 227], s[t + 228], s[t]);          // i = t + 227 in [227, 454)
    __syncthreads();
    if (t < 227) s[t + 227] = v;
    __syncthreads();
    if (t < 169) v = mt_mix(s[t + 454], s[t + 455], s[t + 227]);    // i in [454, 623)
    __syncthreads();
    if (t < 169) s[t + 454] = v;
    __syncthreads();
    if (t == 0) s[623] = mt_mix(s[623], s[0], s[396]);
    __syncthreads();
}

// mt_state_out / mt_pos_out: where the advanced stream is stored (== the inputs: in place; another buffer: the draw is
// speculative -- the ring look-ahead of env.hip -- and the caller commits it by swapping the buffers)
template <typename T>
__device__ inline void mt_normal_body(const uint32_t* mt_state, const int* mt_pos, uint32_t* mt_state_out, int* mt_pos_out,
                                      T* __restrict__ zx, int K, int n_inner, int n_outer, int e) {
    __shared__ uint32_t s[kMtN];
    __shared__ int scan[256];
    __shared__ int sh_pos, sh_got, sh_stop;
    const int t = threadIdx.x;
    const uint32_t* gs = mt_state + (size_t)e * kMtN;
    uint32_t* gso = mt_state_out + (size_t)e * kMtN;
    for (int i = t; i < kMtN; i += 256) s[i] = gs[i];
    if (t == 0) {
        sh_pos = mt_pos[e];
        sh_got = 0;
    }
    __syncthreads();
    const int need_pairs = n_outer / 2;
    T* out = zx + (size_t)e * K + n_inner;
    while (true) {
        if (sh_pos >= kMtN) {
            mt_twist(s);
            if (t == 0) sh_pos = 0;
            __syncthreads();
        }
        const int pos = sh_pos, got = sh_got;
        const int avail = (kMtN - pos) / 4;                      // <= 156 attempts in this block
        int acc = 0;
        double n0 = 0.0, n1 = 0.0;
        if (t < avail) {
            const uint32_t* w = s + pos + 4 * t;
            const double d1 = ((double)(mt_temper(w[0]) >> 5) * 67108864.0 + (double)(mt_temper(w[1]) >> 6)) /
                              9007199254740992.0;
            const double d2 = ((double)(mt_temper(w[2]) >> 5) * 67108864.0 + (double)(mt_temper(w[3]) >> 6)) /
                              9007199254740992.0;
            const double x1 = 2.0 * d1 - 1.0, x2 = 2.0 * d2 - 1.0;
            const double r2 = x1 * x1 + x2 * x2;
            if (r2 < 1.0 && r2 != 0.0) {
                acc = 1;
                const double f = sqrt(-2.0 * log(r2) / r2);
                n0 = f * x2;                                      // returned first
                n1 = f * x1;                                      // cached, returned next
            }
        }
        // inclusive scan of the accept flags over the workgroup: ballot + popcount inside a wave, 4 wave totals in LDS
        const unsigned long long bal = __ballot(acc);
        const int lane = t & (kWave - 1), wv = t / kWave;
        const int in_wave = __popcll(bal & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull)));
        if (lane == 0) scan[wv] = __popcll(bal);
        __syncthreads();
        int before = 0;
        for (int q = 0; q < wv; ++q) before += scan[q];
        const int total_acc = scan[0] + scan[1] + scan[2] + scan[3];
        const int incl = before + in_wave;
        const int remaining = need_pairs - got;
        if (t == 0) sh_stop = -1;
        __syncthreads();
        if (acc && incl <= remaining) {
            const int j = got + incl - 1;
            out[2 * j] = (T)n0;
            out[2 * j + 1] = (T)n1;
            if (incl == remaining) sh_stop = t;                  // the attempt that completes the request
        }
        __syncthreads();
        if (sh_stop >= 0) {
            if (t == 0) sh_pos = pos + 4 * (sh_stop + 1);
            break;
        }
        if (t == 0) {
            sh_pos = kMtN;                                       // block exhausted: consume all of it
            sh_got = got + total_acc;
        }
        __syncthreads();
    }
    __syncthreads();
    for (int i = t; i < kMtN; i += 256) gso[i] = s[i];
    if (t == 0) mt_pos_out[e] = sh_pos;
}

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
