// Device functions of the ring extrusion shared by several kernels: the torus addressing of the screens, the Z gather, and NumPy's
// legacy Gaussian stream (MT19937 + polar Box-Muller) -- used by k_ring_prepare* (atm_kernels.hip), by the ring GEMM launch that
// draws the NEXT crossing's innovations in workgroups of its own (gemm_kernels.hip) and by the fused step kernel, which gathers
// the next crossing's Z right after it has written a ring (step_kernel.hip).
#pragma once
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

}  // namespace ao
