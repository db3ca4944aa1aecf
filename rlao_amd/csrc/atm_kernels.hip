// Atmosphere kernels: integer screen shift + conditioning-ring gather, MT19937/legacy-Gaussian
// innovations, outer-ring scatter.   Reference: OOPAO/Atmosphere.py:301-311 (add_row).
#include "common.hpp"

namespace ao {

// ---------------------------------------------------------------------------------------------------
// add_row, part 1: onePixelShiftedPhaseScreen = warp(map_full, translate(sx, sy))[1:-1, 1:-1]  and
// Z = shifted[innerMask].  A one-pixel translation through the cubic interpolator returns the source
// pixel exactly, so the shift is a strided copy  new[r][c] = old[r - sy][c - sx]  of the N x N interior;
// no interior pixel reads outside the (N+2)^2 map.  The copy goes to the other half of a ping-pong pair
// (in-place would race).  Z is gathered straight from the OLD map (idx - sy*S - sx), so the two loops
// are independent.  grid = (chunks, n_env); lanes walk rows contiguously (coalesced 4/8-byte accesses).
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_shift_gather(const T* __restrict__ old_map, T* __restrict__ new_map,
                                                      T* __restrict__ zx, const int* __restrict__ inner_idx, int S,
                                                      int n_inner, int K, int sx, int sy, int do_copy) {
    const int e = blockIdx.y;
    const size_t base = (size_t)e * S * S;
    const T* src = old_map + base;
    const int shift = sy * S + sx;
    if (do_copy) {
        T* dst = new_map + base;
        const int N = S - 2;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N * N; i += gridDim.x * blockDim.x) {
            const int r = i / N + 1, c = i % N + 1;
            dst[r * S + c] = src[r * S + c - shift];
        }
    }
    if (blockIdx.x == 0) {
        for (int k = threadIdx.x; k < n_inner; k += blockDim.x) zx[(size_t)e * K + k] = src[inner_idx[k] - shift];
    }
}

template <typename T>
int launch_shift_gather(const T* old_map, T* new_map, T* zx, const int* inner_idx, int n_env, int S, int n_inner,
                        int K, int sx, int sy, int do_copy, hipStream_t st) {
    const int N = S - 2;
    int chunks = do_copy ? cdiv(N * N, 256 * 8) : 1;
    dim3 grid(chunks, n_env);
    hipLaunchKernelGGL(k_shift_gather<T>, grid, dim3(256), 0, st, old_map, new_map, zx, inner_idx, S, n_inner, K, sx,
                       sy, do_copy);
    AO_HIP(hipGetLastError());
    return 0;
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

template <typename T>
__global__ void __launch_bounds__(256) k_mt_normal(uint32_t* __restrict__ mt_state, int* __restrict__ mt_pos,
                                                   T* __restrict__ zx, int K, int n_inner, int n_outer) {
    __shared__ uint32_t s[kMtN];
    __shared__ int scan[256];
    __shared__ int sh_pos, sh_got, sh_stop;
    const int e = blockIdx.x;
    const int t = threadIdx.x;
    uint32_t* gs = mt_state + (size_t)e * kMtN;
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
        // inclusive scan of the accept flags over the workgroup
        scan[t] = acc;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            int v = (t >= off) ? scan[t - off] : 0;
            __syncthreads();
            scan[t] += v;
            __syncthreads();
        }
        const int incl = scan[t];
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
        const int total_acc = scan[255];
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
    for (int i = t; i < kMtN; i += 256) gs[i] = s[i];
    if (t == 0) mt_pos[e] = sh_pos;
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
// add_row, part 3: map_full[outerMask] = X   (+ min / max of the whole new map, which the sub-pixel
// warp clips its output to: skimage clip=True).  One workgroup per env.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024) k_scatter_minmax(T* __restrict__ new_map, const T* __restrict__ X,
                                                         const int* __restrict__ outer_idx, T* __restrict__ minmax,
                                                         int S, int n_outer, int splits, size_t slab) {
    __shared__ T red_lo[16], red_hi[16];
    const int e = blockIdx.x;
    T* map = new_map + (size_t)e * S * S;
    const T* x = X + (size_t)e * n_outer;
    T lo = (T)3.0e38, hi = (T)-3.0e38;
    for (int k = threadIdx.x; k < n_outer; k += blockDim.x) {
        T v = x[k];
        for (int z = 1; z < splits; ++z) v += x[(size_t)z * slab + k];       // split-K slabs, fixed order
        map[outer_idx[k]] = v;
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
    const int N = S - 2;
    for (int i = threadIdx.x; i < N * N; i += blockDim.x) {
        const T v = map[(i / N + 1) * S + (i % N) + 1];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
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
int launch_scatter_minmax(T* new_map, const T* X, const int* outer_idx, T* minmax, int n_env, int S, int n_outer,
                          int splits, hipStream_t st) {
    hipLaunchKernelGGL(k_scatter_minmax<T>, dim3(n_env), dim3(1024), 0, st, new_map, X, outer_idx, minmax, S,
                       n_outer, splits, (size_t)n_env * n_outer);
    AO_HIP(hipGetLastError());
    return 0;
}

#define INST(T)                                                                                                    \
    template int launch_shift_gather<T>(const T*, T*, T*, const int*, int, int, int, int, int, int, int,           \
                                        hipStream_t);                                                              \
    template int launch_mt_normal<T>(uint32_t*, int*, T*, int, int, int, int, hipStream_t);                        \
    template int launch_scatter_minmax<T>(T*, const T*, const int*, T*, int, int, int, int, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace ao
