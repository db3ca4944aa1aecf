// Pyramid passes in float32 with register-resident transforms (OOPAO/Pyramid.py:469-607, 987-1006): nRes = 528 = 24 x 22, the 40x40
// Pyramid of BASELINE configs[2], and -- the same kernels, the factor pair is a template parameter -- nRes = 288 = 16 x 18, the 20x20
// Pyramid of the reference's Papyrus set-up (fft528.hpp holds the factors).  The numbers below are those of 528.
// Same three passes, same T1 / T2 layouts and the same arithmetic on the pupil field as pyr_kernels.hip (which keeps every other
// length, float64 and the science PSF); what changes is how a 528-point transform is carried out: a lane holds a whole 24- or
// 22-point factor, a sequence crosses LDS once per transform (the Stockham version: three times), and global memory is read into
// and written from the registers that hold the factors.
//
// "Role A" lanes are (sequence, n2 or m1 < 22) and hold the 24-point factor, "role B" lanes are (sequence, k1 < 24) and hold the
// 22-point factor; the row passes take 8 sequences per workgroup (192 lanes: 176 in role A), the column pass 16 (384 lanes).
//   P1 rows     : the field of 8 pupil rows is built in LDS with the lanes along x; role A takes x = 22 n1 + n2 - off from there,
//                 24-point DFT, twiddle, exchange; role B 22-point DFT and stores X[k1 + 24 k2]: 192 contiguous bytes per sequence
//                 and store.
//   P2 columns  : role A loads 16 neighbouring columns of T1 (one 128-byte line per row), 24-point DFT, twiddle, exchange; role B
//                 22-point DFT, fftshift (k2 -> k2 + 11: a renaming) and mask, inverse 22-point DFT, twiddle, exchange; role A
//                 inverse 24-point DFT and stores T2.  Two exchanges for two transforms.
//   P3 rows^-1  : role B loads rows of T2, inverse 22-point DFT, twiddle, exchange; role A inverse 24-point DFT, |.|^2, summed over
//                 the rows of a camera row and the modulation points (in registers), binned to the camera row.
// LDS layouts are chosen per pass so that the exchange is conflict-free for the lane order that keeps global accesses
// contiguous (bank rules of ds_write_b64 / ds_read_b64, MI355X_MICROARCH.md; scripts/lds_banks_528.py counts them).
// Measured at 1024 envs: 528 (C3, profiles/r03_C3_kernel_stats.csv) 391 + 972 + 521 us against 837 + 2827 + 1583 us for the Stockham
// passes; 288 (scripts/time_papyrus.py) 88 + 295 + 169 against 237 + 778 + 405 us.
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
// Geometry of a factor pair: N = N1 x N2; "role A" lanes (sequence, n2 or m1 < N2) hold the N1-point factor, "role B" lanes
// (sequence, k1 < N1) the N2-point factor.  N2 is even and N / 2 = N1 (N2 / 2): the fftshift is k2 -> k2 + N2 / 2.
template <int N1_, int N2_>
struct Fac {
    static constexpr int N1 = N1_, N2 = N2_, N = N1_ * N2_, H2 = N2_ / 2, MAXL = N1_ > N2_ ? N1_ : N2_;
    static constexpr int TWS = N2_ + 1;                           // twiddle table [k1][n2] with rows of N2 + 1 (odd: P2's inverse reads)
    static constexpr int S2 = N1_ + 1, SEQ1 = N2_ * S2;           // P1 exchange ex[c][n2][k1]: rows of N1 + 1 (odd) -> conflict-free writes
    static constexpr int S1 = N2_ + 1, SEQ3 = N1_ * S1;           // P3 exchange ex[c][k1][m1]
    static constexpr int CB = 16, LANES_C = (CB * MAXL + 63) / 64 * 64;   // column pass: 16 columns = one 128-byte line per row
    static_assert(N2_ % 2 == 0 && SEQ1 >= N && 2 * SEQ3 >= N, "layout");
    static constexpr int lanes(int seqs) { return (seqs * MAXL + 63) / 64 * 64; }
    static constexpr size_t lds_rows(int seqs) { return (size_t)(seqs * SEQ1 + N1 * TWS) * sizeof(v2); }
    static constexpr size_t lds_cols() { return (size_t)(N * CB + N1 * TWS) * sizeof(v2); }
    static constexpr size_t lds_rows_inv(int seqs) { return (size_t)(seqs * SEQ3 + N1 * TWS) * sizeof(v2); }
};
// 528 = 24 x 22: 8 sequences per workgroup of the row passes (192 lanes).  288 = 16 x 18: 14 (252 of 256 lanes) and 12 (P3: 24 rows
// of a workgroup's 4 camera rows in two batches).
using F528 = Fac<24, 22>;
using F288 = Fac<16, 18>;

// LDS is DYNAMIC shared memory on purpose: with a static size the compiler works out the LDS-limited occupancy (3 waves per SIMD if
// the workgroups spread evenly over the 4 SIMDs) and pads the kernel's register allocation up to what that occupancy allows
// (.amdhsa_next_free_vgpr 129 for 90-118 registers in use).  A workgroup of 3 or 6 waves does not spread evenly -- the column pass
// puts 2, 2, 1, 1 waves on the SIMDs -- so a second workgroup needs a fourth slot on two of them, and with 136 registers per lane
// allocated a SIMD holds three: one workgroup per CU instead of two (seen in the per-CU intervals of scripts/diag_pyr_stamps.py;
// scripts/ubench/lds_occupancy.hip shows the LDS itself admits floor(160 / KB)).

// w_N^(k1 n2), k1 < N1, n2 < N2, from the N-entry table of the env (k1 n2 < N), in two steps: the loads are issued at the top of a
// kernel, ahead of the pass's own first loads, and the values are written to LDS once those have been issued too -- one memory
// latency per workgroup instead of two in a row (a workgroup lives for ~9 us, a load takes 1-2).
template <class F, int LANES>
struct TwsRegs {
    static constexpr int K = (F::N1 * F::TWS + LANES - 1) / LANES;
    v2 r[K];
};
template <class F, int LANES>
__device__ inline void tws_issue(TwsRegs<F, LANES>& t, const float* __restrict__ tw, int tid) {
#pragma unroll
    for (int u = 0; u < TwsRegs<F, LANES>::K; ++u) {
        const int i = tid + u * LANES, k1 = i / F::TWS, n2 = i - F::TWS * k1;
        t.r[u] = reinterpret_cast<const v2*>(tw)[(i < F::N1 * F::TWS && n2 < F::N2) ? k1 * n2 : 0];
    }
}
template <class F, int LANES>
__device__ inline void tws_commit(const TwsRegs<F, LANES>& t, v2* __restrict__ tws, int tid) {
#pragma unroll
    for (int u = 0; u < TwsRegs<F, LANES>::K; ++u)
        if (tid + u * LANES < F::N1 * F::TWS) tws[tid + u * LANES] = t.r[u];
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

// ---- P1: grid = (ceil(R / SEQS), chunk, E) ----------------------------------------------------------------------------------------
template <class F, int SEQS>
__global__ void __launch_bounds__(F::lanes(SEQS)) k_pyr528_rows(const PyrArgs<float> a) {
    constexpr int N = F::N, N1 = F::N1, N2 = F::N2, SEQ = F::SEQ1, S2 = F::S2, LANES = F::lanes(SEQS);
    extern __shared__ __align__(16) unsigned char lds_raw[];      // dynamic on purpose (above)
    v2* ex = reinterpret_cast<v2*>(lds_raw);                      // [SEQS * SEQ]  first the field of the rows: fld[c][x - N2 n1_lo]
    v2* tws = ex + SEQS * SEQ;                                    // [N1 * TWS]
    const int tid = threadIdx.x, R = a.R, off = a.off;
    TwsRegs<F, LANES> twr;
    tws_issue(twr, a.tw, tid);
    const int e = blockIdx.z, th = blockIdx.y, y0 = blockIdx.x * SEQS;
    const int n1_lo = off / N2, n1_hi = (off + R - 1) / N2;       // the N1-point inputs that can be inside the pupil, for any lane
    {
        // the field on the columns N2 n1_lo .. N2 (n1_hi + 1) of the padded grid, lanes along x (W = a.seq_per_block columns per row)
        const int W = a.seq_per_block, x_lo = N2 * n1_lo;
        const float* ph = a.phase + (size_t)e * R * R;
        const float* tt = a.tt ? a.tt + (size_t)(a.theta0 + th) * R * R : nullptr;
        const float pi_over_n = (float)(3.14159265358979323846 / N);
        // four points per lane and turn: their loads are independent of one another and issued together (one point at a time,
        // amplitude -> branch -> phase is a chain of two memory latencies per point: 20 us per workgroup)
        for (int i0 = tid; i0 < SEQS * W; i0 += 4 * LANES) {
            float am[4], ang[4];
            int cc[4], xx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * LANES;
                cc[u] = fastdiv(i, a.magic_seq);
                xx[u] = i - cc[u] * W;
                const int x = x_lo + xx[u] - off, row = y0 + cc[u];
                const bool in = i < SEQS * W && row < R && (unsigned)x < (unsigned)R;
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
                if (i0 + u * LANES < SEQS * W) ex[cc[u] * SEQ + xx[u]] = f;
            }
        }
    }
    tws_commit(twr, tws, tid);
    __syncthreads();
    const int ca = tid / N2, n2 = tid - N2 * ca;
    v2 v[N1];
    if (tid < SEQS * N2) {
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            v[n1] = v2{0.f, 0.f};
            if (n1 >= n1_lo && n1 <= n1_hi) v[n1] = ex[ca * SEQ + N2 * (n1 - n1_lo) + n2];
        }
    }
    __syncthreads();                                              // the field has been read: ex becomes the exchange buffer
    if (tid < SEQS * N2) {
        f528::dft_len<N1, false>(v);
#pragma unroll
        for (int k1 = 1; k1 < N1; ++k1) v[k1] = f528::cmul_tw<false>(v[k1], tws[k1 * F::TWS + n2]);
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) ex[ca * SEQ + n2 * S2 + k1] = v[k1];
    }
    __syncthreads();
    if (tid < SEQS * N1) {
        const int c = tid / N1, k1 = tid - N1 * c, row = y0 + c;
        v2 u[N2];
#pragma unroll
        for (int n2b = 0; n2b < N2; ++n2b) u[n2b] = ex[c * SEQ + n2b * S2 + k1];
        f528::dft_len<N2, false>(u);
        if (row < R) {
            v2* t1 = reinterpret_cast<v2*>(a.t1) + (((size_t)e * a.n_theta_chunk + th) * R + row) * N + k1;
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) t1[N1 * k2] = u[k2];
        }
    }
}

// ---- P2: grid = (8 XCDs x slots for N / 16 blocks of 16 columns, chunk, E); 16 columns x max(N1, N2) lanes (528: 384) -------------
// 16 columns are one 128-byte line of T1 / T2 per row: a wave's load or store touches 4 whole lines.  (With 8 columns per workgroup
// -- half lines, the other half read and written by a neighbouring workgroup -- the texture addresser was busy 60 % of the kernel
// against 38 % here, and the L2 served every line of T1 twice.)
// N1LO, N1CNT: the N1-point inputs n1 in [N1LO, N1LO + N1CNT) can lie inside the pupil rows (y = N2 n1 + n2 - off); the others are
// zero padding for every lane.
template <class F, bool SHIFT, int N1LO, int N1CNT>
__global__ void __launch_bounds__(F::LANES_C, 4) k_pyr528_cols(const PyrArgs<float> a) {
    constexpr int N = F::N, N1 = F::N1, N2 = F::N2, CB = F::CB, SF = N1 * CB, SI = N2 * CB;   // ex[n2][k1][c], then ex[k1][m1][c]: 16 lanes = 16 c = 32 banks
    constexpr int NBLK = N / CB, PER_XCD = NBLK / 8, EXTRA = NBLK % 8;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    v2* ex = reinterpret_cast<v2*>(lds_raw);                      // [N2 * SF = N1 * SI = 16 N]
    v2* tws = ex + N2 * SF;                                       // [N1 * TWS]
    const int tid = threadIdx.x, c = tid & 15, j = tid >> 4, R = a.R, off = a.off;
    AO_PSTAMP(0);
    AO_PRT(0);
    AO_PHW();
    TwsRegs<F, F::LANES_C> twr;
    tws_issue(twr, a.tw, tid);
    // blockIdx.x % 8 is the XCD (workgroups go round-robin over the 8 XCDs): XCD x takes the blocks PER_XCD x .. of every env (528: 4)
    // and the EXTRA blocks that are left (528: the 33rd) in turn with the env, so that an XCD's L2 keeps the few blocks of the mask it
    // needs (0.3 MB) while T1 / T2 stream through it.  (Blocks dealt round-robin: every L2 reads all 2.2 MB of the mask between 3 GB of
    // streaming, and misses.)
    const int e = blockIdx.z, th = blockIdx.y, xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    int kx0;
    if (a.generic_fft & 1024) {                                   // diagnostic: blocks dealt round-robin
        if ((int)blockIdx.x >= NBLK) return;
        kx0 = blockIdx.x * CB;
    } else if (slot < PER_XCD) {
        kx0 = (PER_XCD * xcd + slot) * CB;
    } else {
        const int t = (xcd - e) & 7;                              // (uniform for the workgroup)
        if (t >= EXTRA) return;
        kx0 = (8 * PER_XCD + t) * CB;
    }
    // shifted column of frequency kx0 + c (528: N / 2 = 264 is 8 mod 16: with the shift a block is two half lines, and one block wraps)
    const int jx = SHIFT ? (kx0 + c + N / 2) % N : kx0 + c;
    const v2* t1 = reinterpret_cast<const v2*>(a.t1) + ((size_t)e * a.n_theta_chunk + th) * R * N;
    v2* t2 = reinterpret_cast<v2*>(a.t2) + ((size_t)e * a.n_theta_chunk + th) * N * N;
    const v2* mk = reinterpret_cast<const v2*>(a.mask);
    // the mask values of this lane's frequencies: shifted position i holds frequency (i + N/2) mod N, so frequency k1 + N1 k2 sits at
    // i = k1 + N1 ((k2 + N2 / 2) mod N2)   (N / 2 = N1 x N2 / 2; Pyramid.py:486-497)
    const unsigned oj = (unsigned)(j * N + jx);
    v2 v[N1];
    {
        // every lane loads (rows clamped into the pupil, lanes j >= N2 the rows of j = N2 - 1): no branches around the loads, so that
        // the twiddle values requested before them can be waited for alone
        const int jj = j < N2 ? j : N2 - 1;
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) {
            v[n1] = v2{0.f, 0.f};
            if (n1 >= N1LO && n1 < N1LO + N1CNT) {
                const int y = N2 * n1 + jj - off, yc = min(max(y, 0), R - 1);
                const v2 t = t1[(unsigned)(yc * N + kx0 + c)];
                v[n1] = y == yc ? t : v2{0.f, 0.f};
            }
        }
    }
    tws_commit(twr, tws, tid);
    __syncthreads();
    if (j < N2) {
        f528::dft_len<N1, false>(v);
#pragma unroll
        for (int k1 = 1; k1 < N1; ++k1) v[k1] = f528::cmul_tw<false>(v[k1], tws[k1 * F::TWS + j]);
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) ex[j * SF + k1 * CB + c] = v[k1];
    }
    AO_PSTAMP(1);
    __syncthreads();
    AO_PSTAMP(2);
    v2 g[N2];
    if (j < N1) {
        v2 m[N2];                                                 // (requested before the first barrier they were 50 us slower: 44 registers)
#pragma unroll
        for (int i2 = 0; i2 < N2; ++i2) m[i2] = mk[oj + (unsigned)(N1 * N * i2)];
        v2 u[N2];
#pragma unroll
        for (int n2 = 0; n2 < N2; ++n2) u[n2] = ex[n2 * SF + j * CB + c];
        f528::dft_len<N2, false>(u);
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
            const int i2 = SHIFT ? (k2 + F::H2) % N2 : k2;
            g[i2] = f528::cmul2(u[k2], m[i2]);
        }
        f528::dft_len<N2, true>(g);
#pragma unroll
        for (int m1 = 1; m1 < N2; ++m1) g[m1] = f528::cmul_tw<true>(g[m1], tws[j * F::TWS + m1]);
    }
    AO_PSTAMP(3);
    __syncthreads();                                              // every lane has its forward values
    AO_PSTAMP(4);
    if (j < N1) {
#pragma unroll
        for (int m1 = 0; m1 < N2; ++m1) ex[j * SI + m1 * CB + c] = g[m1];
    }
    __syncthreads();
    AO_PSTAMP(5);
    if (j < N2) {
#pragma unroll
        for (int k1 = 0; k1 < N1; ++k1) v[k1] = ex[k1 * SI + j * CB + c];
        f528::dft_len<N1, true>(v);
        AO_PSTAMP(6);
#pragma unroll
        for (int m2 = 0; m2 < N1; ++m2) t2[oj + (unsigned)(N2 * N * m2)] = v[m2];
    }
    AO_PSTAMP(7);
    AO_PRT(1);
}

// ---- P3: grid = (cam / G, E): G camera rows = G nb rows of T2 per modulation point, in batches of SEQS sequences -------------------
// The sums of |.|^2 over the rows of a camera row stay in registers: lane t owns the columns x = t, t + LANES, ... of all G camera rows.
template <class F, int SEQS, int G>
__global__ void __launch_bounds__(F::lanes(SEQS), 4) k_pyr528_rows_inv(const PyrArgs<float> a, int accumulate) {
    constexpr int N = F::N, N1 = F::N1, N2 = F::N2, SEQ = F::SEQ3, S1 = F::S1, LANES = F::lanes(SEQS), NX = (N + LANES - 1) / LANES;   // ex[c][k1][m1]
    static_assert(SEQS * SEQ * 2 >= SEQS * N && SEQS * SEQ * 2 >= 4 * N, "the |.|^2 rows alias the exchange buffer");
    extern __shared__ __align__(16) unsigned char lds_raw[];
    v2* ex = reinterpret_cast<v2*>(lds_raw);                      // [SEQS * SEQ]
    v2* tws = ex + SEQS * SEQ;                                    // [N1 * TWS]
    float* pw = reinterpret_cast<float*>(ex);                     // [SEQS][N] |.|^2 of the batch (after the exchange has been read)
    const int tid = threadIdx.x, nb = N / a.cam, chunk = a.n_theta_chunk;
    TwsRegs<F, LANES> twr;
    tws_issue(twr, a.tw, tid);
    const int e = blockIdx.y, cr0 = blockIdx.x * G;
    const int per_g = chunk * nb, S = G * per_g;                  // sequence s = (g chunk + th) nb + q: row (cr0 + g) nb + q of point th
    const float scale = 1.f / ((float)N * (float)N * (float)N * (float)N);   // ifft2 normalisation 1/N^2 on the amplitude
    const int cb = tid / N1, k1 = tid - N1 * cb;
    const int ca = tid / N2, m1 = tid - N2 * ca;
    const bool role_b = tid < SEQS * N1, role_a = tid < SEQS * N2;
    float acc[G][NX];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
        for (int t = 0; t < NX; ++t) acc[gi][t] = 0.f;
#pragma unroll 1
    for (int s0 = 0; s0 < S; s0 += SEQS) {
        v2 g[N2];
        {
            const bool valid = role_b && s0 + cb < S;
            const int s = valid ? s0 + cb : S - 1;                // (a slot beyond the last sequence loads that one and zeroes it: no branch)
            const int gi = s / per_g, rem = s - gi * per_g, th = rem / nb, q = rem - th * nb;
            const v2* t2 = reinterpret_cast<const v2*>(a.t2) + (((size_t)e * chunk + th) * N + (size_t)(cr0 + gi) * nb + q) * N + (role_b ? k1 : 0);
#pragma unroll
            for (int k2 = 0; k2 < N2; ++k2) {
                const v2 t = t2[N1 * k2];
                g[k2] = valid ? t : v2{0.f, 0.f};
            }
        }
        if (s0 == 0) {                                            // (uniform) the twiddle table, requested before the first rows
            tws_commit(twr, tws, tid);
            __syncthreads();
        }
        if (role_b) {
            f528::dft_len<N2, true>(g);
#pragma unroll
            for (int q1 = 1; q1 < N2; ++q1) g[q1] = f528::cmul_tw<true>(g[q1], tws[k1 * F::TWS + q1]);
        }
        __syncthreads();                                          // the previous batch's sums are taken (first batch: tables are loaded)
        if (role_b) {
#pragma unroll
            for (int q1 = 0; q1 < N2; ++q1) ex[cb * SEQ + k1 * S1 + q1] = g[q1];
        }
        __syncthreads();
        float p[N1];
        if (role_a) {
            v2 v[N1];
#pragma unroll
            for (int k = 0; k < N1; ++k) v[k] = ex[ca * SEQ + k * S1 + m1];
            f528::dft_len<N1, true>(v);
#pragma unroll
            for (int m2 = 0; m2 < N1; ++m2) p[m2] = (v[m2].x * v[m2].x + v[m2].y * v[m2].y) * scale;
        }
        __syncthreads();                                          // pw aliases ex
        if (role_a) {
#pragma unroll
            for (int m2 = 0; m2 < N1; ++m2) pw[ca * N + m1 + N2 * m2] = p[m2];
        }
        __syncthreads();
        // sequences s0 .. s0 + SEQS - 1 belong to camera rows g_first .. (uniform); slot cc goes to acc[g(cc)]
        const int n_in = min(SEQS, S - s0);
#pragma unroll
        for (int t = 0; t < NX; ++t) {
            const int x = tid + t * LANES;
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
        const int x = tid + t * LANES;
        if (x < N)
#pragma unroll
            for (int gi = 0; gi < G; ++gi) pw[gi * N + x] = acc[gi][t];
    }
    __syncthreads();
    float* fr = a.frame + (size_t)e * a.cam * a.cam + (size_t)cr0 * a.cam;
    for (int i = tid; i < G * a.cam; i += LANES) {
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
    r |= hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[0], k_pyr528_rows<F528, 8>, F528::lanes(8), F528::lds_rows(8)) != hipSuccess;
    r |= hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[1], k_pyr528_cols<F528, false, 6, 12>, F528::LANES_C, F528::lds_cols()) != hipSuccess;
    r |= hipOccupancyMaxActiveBlocksPerMultiprocessor(&out3[2], k_pyr528_rows_inv<F528, 8, 4>, F528::lanes(8), F528::lds_rows_inv(8)) != hipSuccess;
    return r;
}
namespace ao {
#endif

// 0: this geometry is not covered (the caller runs the Stockham passes of pyr_kernels.hip)
int pyramid528_supported(const PyrArgs<float>& a) {
    return (a.N == F528::N || a.N == F288::N) && a.R <= a.N && a.off >= 0 && a.off + a.R <= a.N && a.cam > 0 && a.N % a.cam == 0 &&
           !(a.generic_fft & 512);
}

// (dynamic LDS beyond the 64 KiB a kernel may use without asking)
static int launch_lds(void (*kern)(const PyrArgs<float>), dim3 grid, int lanes, size_t lds, hipStream_t st, const PyrArgs<float>& a) {
    if (lds > 64 * 1024) AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(lanes), lds, st, a);
    return 0;
}

// SEQS1 / SEQS3: sequences per workgroup of the row passes; LO, CNT: the specialised input range of the column pass for the
// reference geometry of this length (528: R = 240 centred, n1 = 6 .. 17; 288: R = 120 centred, n1 = 4 .. 11)
template <class F, int SEQS1, int SEQS3, int LO, int CNT>
static int launch_fac(PyrArgs<float> a, int n_theta, int chunk, hipStream_t st) {
    const int R = a.R;
    const int G = a.cam % 4 == 0 ? 4 : (a.cam % 2 == 0 ? 2 : 1);
    const int n1_lo = a.off / F::N2, n1_hi = (a.off + R - 1) / F::N2;
    const bool spec = n1_lo == LO && n1_hi == LO + CNT - 1;
    constexpr int slots = F::N / F::CB / 8 + ((F::N / F::CB) % 8 ? 1 : 0);
    for (int t0 = 0; t0 < n_theta; t0 += chunk) {
        a.theta0 = t0;
        a.n_theta_chunk = (n_theta - t0) < chunk ? (n_theta - t0) : chunk;
        a.seq_per_block = F::N2 * (n1_hi - n1_lo + 1);               // field columns per row that P1 evaluates
        a.magic_seq = fft_magic((unsigned)a.seq_per_block);
        hipLaunchKernelGGL((k_pyr528_rows<F, SEQS1>), dim3(cdiv(R, SEQS1), a.n_theta_chunk, a.n_env), dim3(F::lanes(SEQS1)), F::lds_rows(SEQS1), st, a);
        const dim3 g2((a.generic_fft & 1024) ? cdiv(F::N / F::CB, 8) * 8 : 8 * slots, a.n_theta_chunk, a.n_env);
        if (a.centering) {
            if (spec) AO_TRY(launch_lds(k_pyr528_cols<F, false, LO, CNT>, g2, F::LANES_C, F::lds_cols(), st, a));
            else AO_TRY(launch_lds(k_pyr528_cols<F, false, 0, F::N1>, g2, F::LANES_C, F::lds_cols(), st, a));
        } else {
            if (spec) AO_TRY(launch_lds(k_pyr528_cols<F, true, LO, CNT>, g2, F::LANES_C, F::lds_cols(), st, a));
            else AO_TRY(launch_lds(k_pyr528_cols<F, true, 0, F::N1>, g2, F::LANES_C, F::lds_cols(), st, a));
        }
        const dim3 g3(a.cam / G, a.n_env);
        const int accumulate = t0 > 0 ? 1 : 0;
        constexpr int L3 = F::lanes(SEQS3);
        constexpr size_t lds3 = F::lds_rows_inv(SEQS3);
        if (G == 4) hipLaunchKernelGGL((k_pyr528_rows_inv<F, SEQS3, 4>), g3, dim3(L3), lds3, st, a, accumulate);
        else if (G == 2) hipLaunchKernelGGL((k_pyr528_rows_inv<F, SEQS3, 2>), g3, dim3(L3), lds3, st, a, accumulate);
        else hipLaunchKernelGGL((k_pyr528_rows_inv<F, SEQS3, 1>), g3, dim3(L3), lds3, st, a, accumulate);
        AO_HIP(hipGetLastError());
    }
    return 0;
}

int launch_pyramid528(const PyrArgs<float>& base, int n_theta, int chunk, hipStream_t st) {
    if (base.N == F528::N) return launch_fac<F528, 8, 8, 6, 12>(base, n_theta, chunk, st);
    return launch_fac<F288, 14, 12, 4, 8>(base, n_theta, chunk, st);
}

}  // namespace ao
