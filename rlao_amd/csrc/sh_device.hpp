// Device-side pieces of the Shack-Hartmann path shared by sh_kernels.hip (stand-alone kernels) and
// step_kernel.hip (the fused per-env step kernel).
#pragma once
#include "common.hpp"

#ifndef AO_STAMP
#define AO_STAMP(i) do { } while (0)
#endif

namespace ao {

// Workgroup barrier for hand-offs through LDS only.  __syncthreads() is "s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier": it
// also drains every outstanding GLOBAL load and store of the wave, so each barrier of a kernel that streams to memory
// costs a full memory round trip (write acknowledgements included) and no prefetch survives it.  Here only the LDS
// counter is waited for; global loads issued before the barrier stay in flight (the compiler still waits for them before
// their first use) and stores retire in the background.  Not a fence for data exchanged through global memory.
__device__ inline void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename T> struct cplx { T re, im; };

template <typename T> __device__ inline void sincos_t(T x, T* s, T* c);
template <> __device__ inline void sincos_t<float>(float x, float* s, float* c) { sincosf(x, s, c); }
template <> __device__ inline void sincos_t<double>(double x, double* s, double* c) { sincos(x, s, c); }

// sin/cos through the hardware v_sin_f32 / v_cos_f32 (argument in revolutions) after a two-constant
// Cody-Waite reduction to [-pi, pi]: |phase| reaches ~100 rad in open loop, and float32 phase * (1/2pi)
// alone would lose ~1e-5 rad there.
__device__ inline void sincos_fast(float x, float* s, float* c) {
    const float n = rintf(x * 0.15915494309189535f);
    float r = fmaf(-n, 6.28318548202514648f, x);             // 2 pi rounded to float32 ...
    r = fmaf(-n, -1.74845553e-07f, r);                       // ... and the remainder of 2 pi
    const float t = r * 0.15915494309189535f;
    *s = __builtin_amdgcn_sinf(t);
    *c = __builtin_amdgcn_cosf(t);
}
__device__ inline void sincos_fast(double x, double* s, double* c) { sincos(x, s, c); }

// max over non-negative values (their bit patterns order like unsigned integers).  The running maximum is read first (a
// relaxed device-scope load served by L2) and the atomic is issued only by a value that would raise it: a few hundred
// waves per env raising the same word -- 64 envs' words share two cache lines -- otherwise queue up on one L2 channel
// (ELT-size Shack-Hartmann: 20 k atomics per frame were most of the spots kernel).  A stale read is only ever too small.
__device__ inline void atomic_max_nonneg(float* addr, float v) {
    unsigned int* a = reinterpret_cast<unsigned int*>(addr);
    const unsigned int bits = __float_as_uint(v);
    if (bits > __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a, bits);
}
__device__ inline void atomic_max_nonneg(double* addr, double v) {
    unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
    const unsigned long long bits = (unsigned long long)__double_as_longlong(v);
    if (bits > __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(a, bits);
}

__device__ inline double wave_sum_f64(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

namespace fast6 {
constexpr int P = 6, N = 12, LO = 3, HP = 3, SPW = 21, EST = P * P + 1;
// cos(k pi / 12), k = 0 .. 23
__device__ constexpr double kCos[24] = {
    1.0, 0.96592582628906829, 0.86602540378443865, 0.70710678118654752, 0.5, 0.25881904510252076,
    0.0, -0.25881904510252076, -0.5, -0.70710678118654752, -0.86602540378443865, -0.96592582628906829,
    -1.0, -0.96592582628906829, -0.86602540378443865, -0.70710678118654752, -0.5, -0.25881904510252076,
    0.0, 0.25881904510252076, 0.5, 0.70710678118654752, 0.86602540378443865, 0.96592582628906829};
// exp(-i pi m / 12)
__device__ constexpr double cre(int m) { return kCos[((m % 24) + 24) % 24]; }
__device__ constexpr double cim(int m) { return -kCos[((((m % 24) + 24) % 24) + 18) % 24]; }   // -sin(x) = -cos(x - pi/2)
// stage-2 twiddle ph_a w^{u(a+lo)} = exp(-i pi (a+lo)(13 + 2u)/12)
__device__ constexpr int k2(int u, int a) { return (a + LO) * (13 + 2 * u); }

// The 12 x 12 spectrum of one 6 x 6 lenslet field, binned 2 x 2: lane q of the lenslet's three lanes owns the
// spectral columns v in {2q, 2q+1, 2q+6, 2q+7}, i.e. camera columns Q = q (Ia[u]) and Q = q + 3 (Ib[u]), rows u = 0..5.
// Ej: the lenslet's E0[a][b] = amp e^{i phi} in LDS (36 complex values).  Ia / Ib must be zero on entry.
// LOW_REGS = n > 0: a compiler fence after every n rows of stage 1 keeps the 36 LDS loads from being hoisted together (72 live
// registers): needed where the kernel runs 4 waves per SIMD (128 VGPRs), at the price of less load latency hidden per wave.
// PRESCALED: the field already carries the 1/n of |FFT2(E)/n|^2 (ShackHartmann.py:539), no 1/n^2 on the intensities.
// Every product is an fma chained into its accumulator (4 per complex multiply-add, none for twiddle components that
// are exactly zero): ~1.1 k instructions per lane.
// ctab: the cosine table held across the wave, lane l = cos(pi (l % 24) / 12) (cos_table_lane()): the stage-1 twiddles
// depend on the lane (q) and are fetched with wave shuffles -- indexing the constant table is a global load, i.e. an
// HBM-class latency in front of every lenslet's arithmetic.
template <typename T>
__device__ inline T cos_table_lane() { return (T)kCos[(threadIdx.x & 63) % 24]; }
__device__ inline float lane_read(float v, int l) { return __shfl(v, l); }
__device__ inline double lane_read(double v, int l) { return __shfl(v, l); }

template <typename T, int LOW_REGS = 0, bool PRESCALED = false>
__device__ inline void lenslet_spots(const cplx<T>* __restrict__ Ej, int q, T ctab, T (&Ia)[6], T (&Ib)[6]) {
    // the two spectral columns v = 2q + c, c = 0, 1 (and their partners v + 6) one after the other:
    // keeps only 2 x 6 complex G values live (register pressure decides the occupancy here)
#pragma unroll 1
    for (int c = 0; c < 2; ++c) {
        // stage-1 twiddles ph_b w^{(b+lo) v} = exp(-i pi (b+lo)(13 + 2 v)/12) of this lane's column
        T t1r[P], t1i[P];
#pragma unroll
        for (int b = 0; b < P; ++b) {
            const int x = (b + LO) * (13 + 2 * (2 * q + c));         // <= 184
            const int m = x - 24 * ((x * 171) >> 12);                // x % 24 without a division
            const int m2 = m + 18 >= 24 ? m - 6 : m + 18;
            t1r[b] = lane_read(ctab, m);
            t1i[b] = -lane_read(ctab, m2);
        }
        T G0r[P], G0i[P], G1r[P], G1i[P];                      // columns v and v + 6
#pragma unroll
        for (int a = 0; a < P; ++a) {
            T evr = 0, evi = 0, odr = 0, odi = 0;              // b + lo even: b = 1, 3, 5 ; odd: b = 0, 2, 4
#pragma unroll
            for (int b = 0; b < P; ++b) {
                const cplx<T> x = Ej[a * P + b];
                if ((b + LO) % 2 == 0) {
                    evr = fma(x.re, t1r[b], evr); evr = fma(-x.im, t1i[b], evr);
                    evi = fma(x.re, t1i[b], evi); evi = fma(x.im, t1r[b], evi);
                } else {
                    odr = fma(x.re, t1r[b], odr); odr = fma(-x.im, t1i[b], odr);
                    odi = fma(x.re, t1i[b], odi); odi = fma(x.im, t1r[b], odi);
                }
            }
            G0r[a] = evr + odr; G0i[a] = evi + odi;
            G1r[a] = evr - odr; G1i[a] = evi - odi;
            if (LOW_REGS > 0 && (a + 1) % LOW_REGS == 0) asm volatile("" ::: "memory");
        }
        // stage 2 (compile-time twiddles) + binning; rows u and u + 6 share their products
#pragma unroll
        for (int u = 0; u < P; ++u) {
            T e0r = 0, e0i = 0, o0r = 0, o0i = 0, e1r = 0, e1i = 0, o1r = 0, o1i = 0;
#pragma unroll
            for (int a = 0; a < P; ++a) {
                const double kr = cre(k2(u, a)), ki = cim(k2(u, a));
                if ((a + LO) % 2 == 0) {
                    if (kr != 0) { e0r = fma(G0r[a], (T)kr, e0r); e0i = fma(G0i[a], (T)kr, e0i); e1r = fma(G1r[a], (T)kr, e1r); e1i = fma(G1i[a], (T)kr, e1i); }
                    if (ki != 0) { e0r = fma(-G0i[a], (T)ki, e0r); e0i = fma(G0r[a], (T)ki, e0i); e1r = fma(-G1i[a], (T)ki, e1r); e1i = fma(G1r[a], (T)ki, e1i); }
                } else {
                    if (kr != 0) { o0r = fma(G0r[a], (T)kr, o0r); o0i = fma(G0i[a], (T)kr, o0i); o1r = fma(G1r[a], (T)kr, o1r); o1i = fma(G1i[a], (T)kr, o1i); }
                    if (ki != 0) { o0r = fma(-G0i[a], (T)ki, o0r); o0i = fma(G0r[a], (T)ki, o0i); o1r = fma(-G1i[a], (T)ki, o1r); o1i = fma(G1r[a], (T)ki, o1i); }
                }
            }
            T fr_, fi_;
            if (PRESCALED) {
                fr_ = e0r + o0r; fi_ = e0i + o0i; Ia[u / 2] = fma(fr_, fr_, Ia[u / 2]); Ia[u / 2] = fma(fi_, fi_, Ia[u / 2]);              // (u, v)
                fr_ = e0r - o0r; fi_ = e0i - o0i; Ia[u / 2 + 3] = fma(fr_, fr_, Ia[u / 2 + 3]); Ia[u / 2 + 3] = fma(fi_, fi_, Ia[u / 2 + 3]);  // (u+6, v)
                fr_ = e1r + o1r; fi_ = e1i + o1i; Ib[u / 2] = fma(fr_, fr_, Ib[u / 2]); Ib[u / 2] = fma(fi_, fi_, Ib[u / 2]);              // (u, v+6)
                fr_ = e1r - o1r; fi_ = e1i - o1i; Ib[u / 2 + 3] = fma(fr_, fr_, Ib[u / 2 + 3]); Ib[u / 2 + 3] = fma(fi_, fi_, Ib[u / 2 + 3]);  // (u+6, v+6)
            } else {
                const T inv_n2 = (T)(1.0 / (N * N));
                fr_ = e0r + o0r; fi_ = e0i + o0i; Ia[u / 2] += (fr_ * fr_ + fi_ * fi_) * inv_n2;
                fr_ = e0r - o0r; fi_ = e0i - o0i; Ia[u / 2 + 3] += (fr_ * fr_ + fi_ * fi_) * inv_n2;
                fr_ = e1r + o1r; fi_ = e1i + o1i; Ib[u / 2] += (fr_ * fr_ + fi_ * fi_) * inv_n2;
                fr_ = e1r - o1r; fi_ = e1i - o1i; Ib[u / 2 + 3] += (fr_ * fr_ + fi_ * fi_) * inv_n2;
            }
        }
    }
}
}  // namespace fast6

// Scalar telemetry of one env from the pupil sums v = {Sum atm, Sum atm^2, Sum res, Sum res^2} of the OPD [m]:
// total / residual rms in nm and the Strehl ratio exp(-var(phase))   (MAIN/OOPAOEnv/OOPAOEnv.py:497, 522, 554-555)
template <typename T>
__device__ inline void telemetry_scalars(const FinishArgs<T>& f, int e, int n_env, const double (&v)[4]) {
    const double n = (double)f.n_pupil;
    double var_atm = v[1] / n - (v[0] / n) * (v[0] / n);
    double var_res = v[3] / n - (v[2] / n) * (v[2] / n);
    var_atm = var_atm > 0 ? var_atm : 0;
    var_res = var_res > 0 ? var_res : 0;
    const double total = sqrt(var_atm) * 1e9, resid = sqrt(var_res) * 1e9;
    const double sr = exp(-var_res * f.src_scale * f.src_scale);
    T* scp = f.scal + 4 * e;
    scp[0] = (T)total;
    scp[1] = (T)resid;
    scp[2] = (T)sr;
    if (f.strehl) f.strehl[e] = (T)sr;
    if (f.telemetry_index >= 0) {
        const size_t o = (size_t)f.telemetry_index * n_env + e;
        f.total[o] = (T)total;
        f.residual[o] = (T)resid;
    }
}

// Step tail shared by k_sh_tail and the fused per-env step kernel; the slopes of env e are in LDS (sl[2 nValid]),
// img_s [nAct^2 + n_modes] is LDS scratch (zeroed image + modal coefficients), red[16] LDS doubles.
//   t = M s ; o = -M2C t 1e6 ; integrator ; obs image ; reward ; telemetry      (MAIN/OOPAOEnv/OOPAOEnv.py:491-536)
// Every lane issues all its loads of a product before the first use (NB loads in flight): with one load per fma the
// products are chains of L2 / Infinity-Cache round trips.  TELEMETRY = false: the caller has done telemetry_scalars().
template <typename T, bool TELEMETRY = true>
__device__ inline void tail_from_slopes(const T* sl, T* img_s, double* red, const T* __restrict__ fac_m,
                                        const T* __restrict__ fac_m2c_t, int n_modes, const FinishArgs<T>& f, int e,
                                        int n_valid, int n_env) {
    const int tid = threadIdx.x;
    const int img = f.n_act * f.n_act;
    const int n_sig = 2 * n_valid, A = f.n_valid_act;
    T* tm = img_s + img;                                        // [K] modal coefficients
    typedef T vec4 __attribute__((ext_vector_type(4)));
    // ---- t = M s : 16 lanes per mode, 16-byte loads of the mode's row (rows are 16-byte aligned when nSig % 4 == 0) ----
    if (sizeof(T) == 4 && n_sig % 4 == 0) {
        constexpr int NB = 10;                                   // 16 lanes x 10 float4 = 640 signals per round
        const int n4 = n_sig / 4;
        for (int k0 = 0; k0 < n_modes; k0 += (int)blockDim.x / 16) {
            const int k = k0 + tid / 16, l16 = tid & 15;
            T acc = 0;
            const vec4* row = reinterpret_cast<const vec4*>(fac_m + (size_t)(k < n_modes ? k : 0) * n_sig);
            const vec4* s4 = reinterpret_cast<const vec4*>(sl);
            for (int q0 = l16; q0 < n4; q0 += 16 * NB) {
                vec4 mv[NB];
#pragma unroll
                for (int j = 0; j < NB; ++j) mv[j] = row[q0 + 16 * j < n4 ? q0 + 16 * j : 0];        // unconditional loads
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const bool ok = q0 + 16 * j < n4;
                    const vec4 sv = s4[ok ? q0 + 16 * j : 0];
                    const T d = ((mv[j][0] * sv[0] + mv[j][1] * sv[1]) + mv[j][2] * sv[2]) + mv[j][3] * sv[3];
                    acc += ok ? d : (T)0;
                }
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
            if (k < n_modes && l16 == 0) tm[k] = acc;
        }
    } else {
        constexpr int NB = 40;
        for (int k0 = 0; k0 < n_modes; k0 += (int)blockDim.x / 16) {
            const int k = k0 + tid / 16, l16 = tid & 15;
            T acc = 0;
            if (k < n_modes) {
                const T* row = fac_m + (size_t)k * n_sig;
                for (int q0 = l16; q0 < n_sig; q0 += 16 * NB) {
                    T mv[NB];
#pragma unroll
                    for (int j = 0; j < NB; ++j) mv[j] = row[q0 + 16 * j < n_sig ? q0 + 16 * j : 0];      // unconditional loads
#pragma unroll
                    for (int j = 0; j < NB; ++j) {
                        const bool ok = q0 + 16 * j < n_sig;
                        acc += ok ? mv[j] * sl[ok ? q0 + 16 * j : 0] : (T)0;
                    }
                }
            }
#pragma unroll
            for (int off = 8; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 16);
            if (k < n_modes && l16 == 0) tm[k] = acc;
        }
    }
    lds_barrier();
    AO_STAMP(19);
    // ---- o = -M2C t, integrator, image ------------------------------------------------------------------------------------
    // 4 lanes per group of 4 consecutive actuators: lane part p takes the modes q = p, p + 4, ... with 16-byte loads along
    // the actuator axis, all of them in flight at once (one memory round trip), then two shuffles fold the 4 partial sums.
    double ss = 0.0;
    const int n_grp = (A + 3) / 4;
    for (int g0 = 0; g0 < n_grp; g0 += (int)blockDim.x / 4) {
        constexpr int NB = 13;                                    // modes per lane and round: 52 modes per round
        const int g = g0 + tid / 4, part = tid & 3;
        const bool g_ok = g < n_grp;
        const int k4 = 4 * (g_ok ? g : 0);
        T acc[4] = {0, 0, 0, 0};
        for (int q0 = part; q0 < n_modes; q0 += 4 * NB) {
            vec4 cv[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                // rows of M2C^T are A floats apart: only element-aligned; the last group of a row may read into the next row
                // (inside the buffer, sized for kMaxModes rows) and its extra elements are discarded
                __builtin_memcpy(&cv[j], fac_m2c_t + (q0 + 4 * j < n_modes ? (size_t)(q0 + 4 * j) * A + k4 : 0), sizeof(vec4));
            }
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const bool ok = q0 + 4 * j < n_modes;
                const T t = tm[ok ? q0 + 4 * j : 0];
#pragma unroll
                for (int d = 0; d < 4; ++d) acc[d] += ok ? cv[j][d] * t : (T)0;
            }
        }
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            acc[d] += __shfl_xor(acc[d], 1, 4);
            acc[d] += __shfl_xor(acc[d], 2, 4);
        }
        // lane `part` finishes actuator k4 + part
        const int k = k4 + part;
        if (g_ok && k < A) {
            const T mine = part == 0 ? acc[0] : (part == 1 ? acc[1] : (part == 2 ? acc[2] : acc[3]));
            const int px = f.act_idx[k];
            T* ob = f.obs + (size_t)e * img;
            if (f.do_integrate) {
                const T prev = (f.gain_from_obs != (T)0) ? ob[px] : f.action[(size_t)e * img + px];
                const T act = (f.gain_from_obs != (T)0) ? f.gain_from_obs * prev : prev;
                const float af = (float)act;                      // float32 increment, see k_recon_finish
                const T inc = ((T)af == act) ? (T)(af * 1e-6f) : act * (T)1e-6;
                const T cn = f.dm_prev[(size_t)e * A + k] * f.leak + inc;      // OOPAOEnv.py:508-509
                f.coefs[(size_t)e * A + k] = cn;
                f.dm_prev[(size_t)e * A + k] = cn;
            }
            const T o = -mine * (T)1e6;
            img_s[px] = o;
            ss += (double)o * (double)o;
        }
    }
    lds_barrier();
    AO_STAMP(20);
    T* ob = f.obs + (size_t)e * img;
    for (int q = tid; q < img; q += blockDim.x) ob[q] = img_s[q];
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);
    if ((tid & (kWave - 1)) == 0) red[tid / kWave] = ss;
    lds_barrier();
    AO_STAMP(21);
    if (tid == 0) {
        double tot = 0;
        for (int q = 0; q < (int)blockDim.x / kWave; ++q) tot += red[q];
        if (f.reward) f.reward[e] = (T)(-sqrt(tot));
        if (f.ret && f.do_integrate) f.ret[e] += (T)(-sqrt(tot));
        if (TELEMETRY) {
            double v[4] = {0, 0, 0, 0};
            const double* pp = f.part + (size_t)e * f.n_tiles * 4;
            for (int t = 0; t < f.n_tiles; ++t)
                for (int k = 0; k < 4; ++k) v[k] += pp[t * 4 + k];
            telemetry_scalars<T>(f, e, n_env, v);
        }
    }
}

}  // namespace ao
