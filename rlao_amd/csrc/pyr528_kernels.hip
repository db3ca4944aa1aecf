// Pyramid passes for nRes = 528 in float32 (the 40x40 Pyramid of BASELINE configs[2], OOPAO/Pyramid.py:469-607, 987-1006) on the
// register-resident 24 x 22 transform of fft528.hpp.  Same three passes, same T1 / T2 layouts and the same arithmetic on the
// pupil field as pyr_kernels.hip (which keeps every other length, float64 and the science PSF); what changes is how a
// 528-point transform is carried out: a lane holds a whole 24- or 22-point factor, a sequence crosses LDS once per transform
// (the Stockham version: three times), and global memory is read into and written from the registers that hold the factors.
//
// A workgroup is 192 lanes = 8 sequences x 24 lanes: "role A" is (sequence, n2 or m1 < 22) -- 176 lanes, the 24-point factor --
// and "role B" is (sequence, k1 < 24), the 22-point factor.
//   P1 rows     : the field of 8 pupil rows is built in LDS with the lanes along x; role A takes x = 22 n1 + n2 - off from there,
//                 24-point DFT, twiddle, exchange; role B 22-point DFT and stores X[k1 + 24 k2]: 192 contiguous bytes per sequence
//                 and store.
//   P2 columns  : (16 sequences, 384 lanes) role A loads 16 neighbouring columns of T1 (one 128-byte line per row), 24-point DFT, twiddle, exchange; role B
//                 22-point DFT, fftshift (k2 -> k2 + 11: a renaming) and mask, inverse 22-point DFT, twiddle, exchange; role A
//                 inverse 24-point DFT and stores T2.  Two exchanges for two transforms.
//   P3 rows^-1  : role B loads rows of T2, inverse 22-point DFT, twiddle, exchange; role A inverse 24-point DFT, |.|^2, summed over
//                 the rows of a camera row and the modulation points in LDS, binned to the camera row.
// LDS layouts are chosen per pass so that the exchange is conflict-free for the lane order that keeps global accesses
// contiguous (bank rules of ds_write_b64 / ds_read_b64, MI355X_MICROARCH.md; scripts/lds_banks_528.py counts them).
// Measured at 1024 envs (C3, profiles/r03_C3_kernel_stats.csv): 395 + 975 + 535 us against 837 + 2827 + 1583 us for the Stockham passes.
#include "common.hpp"
#include "fft.hpp"
#include "fft528.hpp"

namespace ao {

using f528::v2;
// diagnostic build (-DAO_PYR_STAMPS, scripts/diag_pyr_stamps.py): s_memtime per wave at the phases of the column pass
#ifdef AO_PYR_STAMPS
__device__ unsigned long long g_pstamps[32 * 40 * 6 * 8];
__device__ unsigned long long g_prt[32 * 40 * 3];               // s_memrealtime (100 MHz) at the start and the end of wave 0
#define AO_PSTAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.y == 0 && blockIdx.z >= 500 && blockIdx.z < 532) \
    g_pstamps[(((blockIdx.z - 500) * 40 + blockIdx.x) * 6 + (threadIdx.x >> 6)) * 8 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#define AO_PRT(i) do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z >= 500 && blockIdx.z < 532) \
    g_prt[((blockIdx.z - 500) * 40 + blockIdx.x) * 3 + (i)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define AO_PHW() do { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.z >= 500 && blockIdx.z < 532) \
    g_prt[((blockIdx.z - 500) * 40 + blockIdx.x) * 3 + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492); } while (0)
extern "C" int aoenv_debug_pstamps(unsigned long long* h_out) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_pstamps), sizeof(unsigned long long) * 32 * 40 * 6 * 8) == hipSuccess ? 0 : 1;
}
extern "C" int aoenv_debug_prt(unsigned long long* h_out) {
    return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(g_prt), sizeof(unsigned long long) * 32 * 40 * 3) == hipSuccess ? 0 : 1;
}
#else
#define AO_PSTAMP(i) do { } while (0)
#define AO_PRT(i) do { } while (0)
#define AO_PHW() do { } while (0)
#endif
#ifdef AO_PYR_STAMPS
}  // namespace ao
extern "C" int aoenv_debug_pyr_occupancy(int* out3);
namespace ao {
#endif
constexpr int kLanes528 = 192, kTws = 23;                         // twiddle table [k1][n2] with rows of 23 (odd: see P2's inverse reads)

// LDS bytes of the three passes, passed at launch.  The buffers are DYNAMIC shared memory on purpose: with a static size the compiler
// works out the LDS-limited occupancy (3 waves per SIMD if the workgroups spread evenly over the 4 SIMDs) and pads the kernel's
// register allocation up to what that occupancy allows (.amdhsa_next_free_vgpr 129 for 90-118 registers in use).  A workgroup of 3
// or 6 waves does not spread evenly -- the column pass puts 2, 2, 1, 1 waves on the SIMDs -- so a second workgroup needs a fourth
// slot on two of them, and with 136 registers per lane allocated a SIMD holds three: one workgroup per CU instead of two (seen in
// the per-CU intervals of scripts/diag_pyr_stamps.py; scripts/ubench/lds_occupancy.hip shows the LDS itself admits floor(160 / KB)).
constexpr size_t kLdsRows = (8 * 550 + 24 * kTws) * sizeof(v2), kLdsCols = (22 * 24 * 16 + 24 * kTws) * sizeof(v2),
                 kLdsRowsInv = (8 * 552 + 24 * kTws) * sizeof(v2);


// w_528^(k1 n2), k1 < 24, n2 < 22, from the N-entry table of the env (k1 n2 <= 483 < 528), in two steps: the loads are issued at the
// top of a kernel, ahead of the pass's own first loads, and the values are written to LDS once those have been issued too -- one
// memory latency per workgroup instead of two in a row (a workgroup lives for ~9 us, a load takes 1-2).
template <int LANES>
struct TwsRegs {
    static constexpr int K = (24 * kTws + LANES - 1) / LANES;
    v2 r[K];
};
template <int LANES>
__device__ inline void tws_issue(TwsRegs<LANES>& t, const float* __restrict__ tw, int tid) {
#pragma unroll
    for (int u = 0; u < TwsRegs<LANES>::K; ++u) {
        const int i = tid + u * LANES, k1 = i / kTws, n2 = i - kTws * k1;
        t.r[u] = reinterpret_cast<const v2*>(tw)[(i < 24 * kTws && n2 < 22) ? k1 * n2 : 0];
    }
}
template <int LANES>
__device__ inline void tws_commit(const TwsRegs<LANES>& t, v2* __restrict__ tws, int tid) {
#pragma unroll
    for (int u = 0; u < TwsRegs<LANES>::K; ++u)
        if (tid + u * LANES < 24 * kTws) tws[tid + u * LANES] = t.r[u];
}

// sin and cos of a float32 angle: three-term Cody-Waite reduction by pi/2 with fused multiply-adds (exact to float32 rounding
// for |x| < 3e4 rad: the quotient has 15 bits) and the fdlibm float kernels on |r| <= pi/4; larger angles take the library's
// routine.  ~30 instructions against ~150 for an inlined sincosf (whose large-argument path is never taken by a wave-front
// of a few hundred radians); both are within 1-2 ulp.
__device__ inline void sincos_cw(float x, float* sn, float* cs) {
    if (fabsf(x) > 30000.f) {
        sincosf(x, sn, cs);
        return;
    }
    const float k = rintf(x * 0.63661977236758134308f);
    float r = fmaf(-k, 1.57079637050628662109375f, x);
    r = fmaf(-k, -4.37113900018624283e-8f, r);
    r = fmaf(-k, -1.71512449079261335e-15f, r);
    const float z = r * r;
    const float ps = fmaf(fmaf(fmaf(2.7183114939898219064e-6f, z, -1.98393348360966317347e-4f), z, 8.3333293858894631756e-3f), z, -0.166666666416265235595f);
    const float pc = fmaf(fmaf(fmaf(2.43904487962774090654e-5f, z, -1.38867637746099294692e-3f), z, 4.16666233237390631894e-2f), z, -0.499999997251031003120f);
    const float s = fmaf(r * z, ps, r), c = fmaf(z, pc, 1.f);
    const int q = (int)k;
    const float ss = (q & 1) ? c : s, cc = (q & 1) ? s : c;
    *sn = (q & 2) ? -ss : ss;
    *cs = ((q + 1) & 2) ? -cc : cc;
}

// ---- P1: grid = (ceil(R / 8), chunk, E) -----------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kLanes528) k_pyr528_rows(const PyrArgs<float> a) {
    constexpr int N = f528::kN, SEQ = 550, S2 = 25;               // ex[c][n2][k1]: rows of 25 (odd) -> conflict-free writes
    extern __shared__ __align__(16) unsigned char lds_raw[];      // dynamic on purpose: see kLdsRows
    v2* ex = reinterpret_cast<v2*>(lds_raw);                      // [8 * SEQ]  first the field of the 8 rows: fld[c][x - 22 n1_lo]
    v2* tws = ex + 8 * SEQ;                                       // [24 * kTws]
    const int tid = threadIdx.x, R = a.R, off = a.off;
    TwsRegs<kLanes528> twr;
    tws_issue(twr, a.tw, tid);
    const int e = blockIdx.z, th = blockIdx.y, y0 = blockIdx.x * 8;
    const int n1_lo = off / 22, n1_hi = (off + R - 1) / 22;       // the 24-point inputs that can be inside the pupil, for any lane
    {
        // the field on the columns 22 n1_lo .. 22 (n1_hi + 1) of the padded grid, lanes along x (W = a.seq_per_block columns per row)
        const int W = a.seq_per_block, x_lo = 22 * n1_lo;
        const float* ph = a.phase + (size_t)e * R * R;
        const float* tt = a.tt ? a.tt + (size_t)(a.theta0 + th) * R * R : nullptr;
        const float pi_over_n = (float)(3.14159265358979323846 / N);
        // four points per lane and turn: their loads are independent of one another and issued together (one point at a time,
        // amplitude -> branch -> phase is a chain of two memory latencies per point: 20 us per workgroup)
        for (int i0 = tid; i0 < 8 * W; i0 += 4 * kLanes528) {
            float am[4], ang[4];
            int cc[4], xx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * kLanes528;
                cc[u] = fastdiv(i, a.magic_seq);
                xx[u] = i - cc[u] * W;
                const int x = x_lo + xx[u] - off, row = y0 + cc[u];
                const bool in = i < 8 * W && row < R && (unsigned)x < (unsigned)R;
                const int p = in ? row * R + x : 0;
                am[u] = in ? a.amp[p] : 0.f;
                ang[u] = ph[p];
                if (tt) ang[u] += tt[p];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v2 f = v2{0.f, 0.f};
                if (am[u] != 0.f) {
                    // centred mask: exp(-i pi (N+1)/N (x + y)) on the padded grid, angle reduced mod 2 pi in integers (k_pyr_rows)
                    const int xg = x_lo + xx[u], row = y0 + cc[u];
                    const float pang = a.phasor_mult ? pi_over_n * (float)((a.phasor_mult * (xg + row + off)) % (2 * N)) : 0.f;
                    float sn, co;
                    sincos_cw(ang[u] - pang, &sn, &co);
                    f = v2{am[u] * co, am[u] * sn};
                }
                if (i0 + u * kLanes528 < 8 * W) ex[cc[u] * SEQ + xx[u]] = f;
            }
        }
    }
    tws_commit(twr, tws, tid);
    __syncthreads();
    const int ca = tid / 22, n2 = tid - 22 * ca;
    v2 v[24];
    if (tid < 176) {
#pragma unroll
        for (int n1 = 0; n1 < 24; ++n1) {
            v[n1] = v2{0.f, 0.f};
            if (n1 >= n1_lo && n1 <= n1_hi) v[n1] = ex[ca * SEQ + 22 * (n1 - n1_lo) + n2];
        }
    }
    __syncthreads();                                              // the field has been read: ex becomes the exchange buffer
    if (tid < 176) {
        f528::dft24<false>(v);
#pragma unroll
        for (int k1 = 1; k1 < 24; ++k1) v[k1] = f528::cmul_tw<false>(v[k1], tws[k1 * kTws + n2]);
#pragma unroll
        for (int k1 = 0; k1 < 24; ++k1) ex[ca * SEQ + n2 * S2 + k1] = v[k1];
    }
    __syncthreads();
    {
        const int c = tid / 24, k1 = tid - 24 * c, row = y0 + c;
        v2 u[22];
#pragma unroll
        for (int n2b = 0; n2b < 22; ++n2b) u[n2b] = ex[c * SEQ + n2b * S2 + k1];
        f528::dft22<false>(u);
        if (row < R) {
            v2* t1 = reinterpret_cast<v2*>(a.t1) + (((size_t)e * a.n_theta_chunk + th) * R + row) * N + k1;
#pragma unroll
            for (int k2 = 0; k2 < 22; ++k2) t1[24 * k2] = u[k2];
        }
    }
}

// ---- P2: grid = (8 XCDs x 5 slots for 33 blocks of 16 columns, chunk, E); 384 lanes = 16 columns x 24 ---------------------------------------------------
// 16 columns are one 128-byte line of T1 / T2 per row: a wave's load or store touches 4 whole lines.  (With 8 columns per workgroup
// -- half lines, the other half read and written by a neighbouring workgroup -- the texture addresser was busy 60 % of the kernel
// against 38 % here, and the L2 served every line of T1 twice.)
// N1LO, N1CNT: the 24-point inputs n1 in [N1LO, N1LO + N1CNT) can lie inside the pupil rows (y = 22 n1 + n2 - off); the others are
// zero padding for every lane.
template <bool SHIFT, int N1LO, int N1CNT>
__global__ void __launch_bounds__(384, 4) k_pyr528_cols(const PyrArgs<float> a) {
    constexpr int N = f528::kN, CB = 16, SF = 24 * CB, SI = 22 * CB;   // ex[n2][k1][c], then ex[k1][m1][c]: 16 lanes = 16 c = 32 banks
    extern __shared__ __align__(16) unsigned char lds_raw[];
    v2* ex = reinterpret_cast<v2*>(lds_raw);                      // [22 * SF = 24 * SI = 8448]
    v2* tws = ex + 22 * SF;                                       // [24 * kTws]
    const int tid = threadIdx.x, c = tid & 15, j = tid >> 4, R = a.R, off = a.off;
    AO_PSTAMP(0);
    AO_PRT(0);
    AO_PHW();
    TwsRegs<384> twr;
    tws_issue(twr, a.tw, tid);
    // blockIdx.x % 8 is the XCD (workgroups go round-robin over the 8 XCDs): XCD x takes the blocks 4 x .. 4 x + 3 of every env and
    // the 33rd block of the envs with e % 8 = x, so that an XCD's L2 keeps the 5 blocks of the mask it needs (0.3 MB) while T1 / T2
    // stream through it.  (Blocks dealt round-robin: every L2 reads all 2.2 MB of the mask between 3 GB of streaming, and misses.)
    const int e = blockIdx.z, th = blockIdx.y, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int kx0;
    if (a.generic_fft & 1024) {                                   // diagnostic: blocks dealt round-robin
        if (blockIdx.x >= 33) return;
        kx0 = blockIdx.x * CB;
    } else {
        if (slot == 4 && (e & 7) != xcd) return;                 // (uniform for the workgroup)
        kx0 = (slot < 4 ? 4 * xcd + slot : 32) * CB;
    }
    // shifted column of frequency kx0 + c (N / 2 = 264 is 8 mod 16: with the shift a block is two half lines, and one block wraps)
    const int jx = SHIFT ? (kx0 + c + N / 2) % N : kx0 + c;
    const v2* t1 = reinterpret_cast<const v2*>(a.t1) + ((size_t)e * a.n_theta_chunk + th) * R * N;
    v2* t2 = reinterpret_cast<v2*>(a.t2) + ((size_t)e * a.n_theta_chunk + th) * N * N;
    const v2* mk = reinterpret_cast<const v2*>(a.mask);
    // the mask values of this lane's frequencies: shifted position i holds frequency (i + N/2) mod N, so frequency k1 + 24 k2 sits at
    // i = k1 + 24 ((k2 + 11) mod 22)   (N / 2 = 24 x 11; Pyramid.py:486-497)
    const unsigned oj = (unsigned)(j * N + jx);
    v2 v[24];
    {
        // every lane loads (rows clamped into the pupil, lanes j >= 22 the rows of j = 21): no branches around the loads, so that the
        // twiddle values requested before them can be waited for alone
        const int jj = j < 22 ? j : 21;
#pragma unroll
        for (int n1 = 0; n1 < 24; ++n1) {
            v[n1] = v2{0.f, 0.f};
            if (n1 >= N1LO && n1 < N1LO + N1CNT) {
                const int y = 22 * n1 + jj - off, yc = min(max(y, 0), R - 1);
                const v2 t = t1[(unsigned)(yc * N + kx0 + c)];
                v[n1] = y == yc ? t : v2{0.f, 0.f};
            }
        }
    }
    tws_commit(twr, tws, tid);
    __syncthreads();
    if (j < 22) {
        f528::dft24<false>(v);
#pragma unroll
        for (int k1 = 1; k1 < 24; ++k1) v[k1] = f528::cmul_tw<false>(v[k1], tws[k1 * kTws + j]);
#pragma unroll
        for (int k1 = 0; k1 < 24; ++k1) ex[j * SF + k1 * CB + c] = v[k1];
    }
    AO_PSTAMP(1);
    __syncthreads();
    AO_PSTAMP(2);
    v2 m[22];                                                     // (requested before the first barrier they were 50 us slower: 44 registers)
#pragma unroll
    for (int i2 = 0; i2 < 22; ++i2) m[i2] = mk[oj + (unsigned)(24 * N * i2)];
    v2 u[22];
#pragma unroll
    for (int n2 = 0; n2 < 22; ++n2) u[n2] = ex[n2 * SF + j * CB + c];
    f528::dft22<false>(u);
    v2 g[22];
#pragma unroll
    for (int k2 = 0; k2 < 22; ++k2) {
        const int i2 = SHIFT ? (k2 + 11) % 22 : k2;
        g[i2] = f528::cmul2(u[k2], m[i2]);
    }
    f528::dft22<true>(g);
#pragma unroll
    for (int m1 = 1; m1 < 22; ++m1) g[m1] = f528::cmul_tw<true>(g[m1], tws[j * kTws + m1]);
    AO_PSTAMP(3);
    __syncthreads();                                              // every lane has its forward values
    AO_PSTAMP(4);
#pragma unroll
    for (int m1 = 0; m1 < 22; ++m1) ex[j * SI + m1 * CB + c] = g[m1];
    __syncthreads();
    AO_PSTAMP(5);
    if (j < 22) {
#pragma unroll
        for (int k1 = 0; k1 < 24; ++k1) v[k1] = ex[k1 * SI + j * CB + c];
        f528::dft24<true>(v);
        AO_PSTAMP(6);
#pragma unroll
        for (int m2 = 0; m2 < 24; ++m2) t2[oj + (unsigned)(22 * N * m2)] = v[m2];
    }
    AO_PSTAMP(7);
    AO_PRT(1);
}

// ---- P3: grid = (cam / G, E): G camera rows = G nb rows of T2 per modulation point, in batches of 8 sequences ------------------------
// The sums of |.|^2 over the rows of a camera row stay in registers: lane t owns the columns x = t, t + 192, t + 384 of all G camera rows.
template <int G>
__global__ void __launch_bounds__(kLanes528, 4) k_pyr528_rows_inv(const PyrArgs<float> a, int accumulate) {
    constexpr int N = f528::kN, SEQ = 552, S1 = 23, NX = (N + kLanes528 - 1) / kLanes528;   // ex[c][k1][m1]
    extern __shared__ __align__(16) unsigned char lds_raw[];
    v2* ex = reinterpret_cast<v2*>(lds_raw);                      // [8 * SEQ]
    v2* tws = ex + 8 * SEQ;                                       // [24 * kTws]
    float* pw = reinterpret_cast<float*>(ex);                     // [8][N] |.|^2 of the batch (after the exchange has been read)
    const int tid = threadIdx.x, nb = N / a.cam, chunk = a.n_theta_chunk;
    TwsRegs<kLanes528> twr;
    tws_issue(twr, a.tw, tid);
    const int e = blockIdx.y, cr0 = blockIdx.x * G;
    const int per_g = chunk * nb, S = G * per_g;                  // sequence s = (g chunk + th) nb + q: row (cr0 + g) nb + q of point th
    const float scale = 1.f / ((float)N * (float)N * (float)N * (float)N);   // ifft2 normalisation 1/N^2 on the amplitude
    const int cb = tid / 24, k1 = tid - 24 * cb;
    const int ca = tid / 22, m1 = tid - 22 * ca;
    float acc[G][NX];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int t = 0; t < NX; ++t) acc[gi][t] = 0.f;
#pragma unroll 1
    for (int s0 = 0; s0 < S; s0 += 8) {
        v2 g[22];
        {
            const bool valid = s0 + cb < S;
            const int s = valid ? s0 + cb : S - 1;                // (a slot beyond the last sequence loads that one and zeroes it: no branch)
            const int gi = s / per_g, rem = s - gi * per_g, th = rem / nb, q = rem - th * nb;
            const v2* t2 = reinterpret_cast<const v2*>(a.t2) + (((size_t)e * chunk + th) * N + (size_t)(cr0 + gi) * nb + q) * N + k1;
#pragma unroll
            for (int k2 = 0; k2 < 22; ++k2) {
                const v2 t = t2[24 * k2];
                g[k2] = valid ? t : v2{0.f, 0.f};
            }
        }
        if (s0 == 0) {                                            // (uniform) the twiddle table, requested before the first rows
            tws_commit(twr, tws, tid);
            __syncthreads();
        }
        f528::dft22<true>(g);
#pragma unroll
        for (int q1 = 1; q1 < 22; ++q1) g[q1] = f528::cmul_tw<true>(g[q1], tws[k1 * kTws + q1]);
        __syncthreads();                                          // the previous batch's sums are taken (first batch: tables are loaded)
#pragma unroll
        for (int q1 = 0; q1 < 22; ++q1) ex[cb * SEQ + k1 * S1 + q1] = g[q1];
        __syncthreads();
        float p[24];
        if (tid < 176) {
            v2 v[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) v[k] = ex[ca * SEQ + k * S1 + m1];
            f528::dft24<true>(v);
#pragma unroll
            for (int m2 = 0; m2 < 24; ++m2) p[m2] = (v[m2].x * v[m2].x + v[m2].y * v[m2].y) * scale;
        }
        __syncthreads();                                          // pw aliases ex
        if (tid < 176) {
#pragma unroll
            for (int m2 = 0; m2 < 24; ++m2) pw[ca * N + m1 + 22 * m2] = p[m2];
        }
        __syncthreads();
        // sequences s0 .. s0 + 7 belong to camera rows g_first .. (uniform); slot cc goes to acc[g(cc)]
        const int n_in = min(8, S - s0);
#pragma unroll
        for (int t = 0; t < NX; ++t) {
            const int x = tid + t * kLanes528;
            if (x < N) {
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    const int lo = max(gi * per_g - s0, 0), hi = min((gi + 1) * per_g - s0, n_in);   // slots of camera row gi in this batch
                    float run = 0.f;
                    for (int cc = lo; cc < hi; ++cc) run += pw[cc * N + x];
                    acc[gi][t] += run;
                }
            }
        }
    }
    __syncthreads();                                              // the last batch's pw has been read
#pragma unroll
    for (int t = 0; t < NX; ++t) {
        const int x = tid + t * kLanes528;
        if (x < N)
#pragma unroll
            for (int gi = 0; gi < G; ++gi) pw[gi * N + x] = acc[gi][t];
    }
    __syncthreads();
    float* fr = a.frame + (size_t)e * a.cam * a.cam + (size_t)cr0 * a.cam;
    for (int i = tid; i < G * a.cam; i += kLanes528) {
        const int gi = i / a.cam, cc = i - gi * a.cam;
        float s = 0.f;
        for (int q = 0; q < nb; ++q) s += pw[gi * N + cc * nb + q];
        fr[i] = accumulate ? fr[i] + s : s;
    }
}

#ifdef AO_PYR_STAMPS
}  // namespace ao
extern "C" int aoenv_debug_pyr_occupancy(int* out3) {            // resident workgroups per CU of the three passes (runtime's answer)
    using namespace ao;
    int r = 0;
    r |= hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[0], k_pyr528_rows, kLanes528, kLdsRows) != hipSuccess;
    r |= hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[1], k_pyr528_cols<false, 6, 12>, 384, kLdsCols) != hipSuccess;
    r |= hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[2], k_pyr528_rows_inv<4>, kLanes528, kLdsRowsInv) != hipSuccess;
    return r;
}
namespace ao {
#endif

// 0: this geometry is not covered (the caller runs the Stockham passes of pyr_kernels.hip)
int pyramid528_supported(const PyrArgs<float>& a) {
    return a.N == f528::kN && a.R <= a.N && a.off >= 0 && a.off + a.R <= a.N && a.cam > 0 && a.N % a.cam == 0 && !(a.generic_fft & 512);
}

// (the column pass needs more than the 64 KiB a kernel may use without asking)
static int launch_cols(void (*kern)(const PyrArgs<float>), dim3 grid, hipStream_t st, const PyrArgs<float>& a) {
    AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsCols));
    hipLaunchKernelGGL(kern, grid, dim3(384), kLdsCols, st, a);
    return 0;
}

int launch_pyramid528(const PyrArgs<float>& base, int n_theta, int chunk, hipStream_t st) {
    PyrArgs<float> a = base;
    const int R = a.R;
    const int G = a.cam % 4 == 0 ? 4 : (a.cam % 2 == 0 ? 2 : 1);
    const bool c3 = a.off / 22 == 6 && (a.off + R - 1) / 22 == 17;           // R = 240 centred in 528: inputs n1 = 6 .. 17
    for (int t0 = 0; t0 < n_theta; t0 += chunk) {
        a.theta0 = t0;
        a.n_theta_chunk = (n_theta - t0) < chunk ? (n_theta - t0) : chunk;
        a.seq_per_block = 22 * ((a.off + R - 1) / 22 - a.off / 22 + 1);       // field columns per row that P1 evaluates
        a.magic_seq = fft_magic((unsigned)a.seq_per_block);
        hipLaunchKernelGGL(k_pyr528_rows, dim3(cdiv(R, 8), a.n_theta_chunk, a.n_env), dim3(kLanes528), kLdsRows, st, a);
        const dim3 g2(8 * 5, a.n_theta_chunk, a.n_env);               // 33 blocks of 16 columns: 4 per XCD + 1 (see the kernel)
        if (a.centering) {
            if (c3) AO_TRY(launch_cols(k_pyr528_cols<false, 6, 12>, g2, st, a));
            else AO_TRY(launch_cols(k_pyr528_cols<false, 0, 24>, g2, st, a));
        } else {
            if (c3) AO_TRY(launch_cols(k_pyr528_cols<true, 6, 12>, g2, st, a));
            else AO_TRY(launch_cols(k_pyr528_cols<true, 0, 24>, g2, st, a));
        }
        const dim3 g3(a.cam / G, a.n_env);
        const int accumulate = t0 > 0 ? 1 : 0;
        if (G == 4) hipLaunchKernelGGL(k_pyr528_rows_inv<4>, g3, dim3(kLanes528), kLdsRowsInv, st, a, accumulate);
        else if (G == 2) hipLaunchKernelGGL(k_pyr528_rows_inv<2>, g3, dim3(kLanes528), kLdsRowsInv, st, a, accumulate);
        else hipLaunchKernelGGL(k_pyr528_rows_inv<1>, g3, dim3(kLanes528), kLdsRowsInv, st, a, accumulate);
        AO_HIP(hipGetLastError());
    }
    return 0;
}

}  // namespace ao
