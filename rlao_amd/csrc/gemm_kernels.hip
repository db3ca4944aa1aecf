// Batched contractions over the env dimension and the small per-env epilogues.
//   X = Z A^T + xi B^T      OOPAO/Atmosphere.py:308      (one product with the operands concatenated: [Z|xi] [A|B]^T)
//   v = R s                 MAIN/OOPAOEnv/OOPAOEnv.py:517
//   dm.OPD = modes @ coefs  OOPAO/DeformableMirror.py:556 (dense-DM path)
// All three are "NT" products  C[M][N] = X[M][K] . W[N][K]^T  with M = n_env and W shared by every env.
#include "common.hpp"
#include "ring_device.hpp"

namespace ao {

// Generic tiled kernel, any T.  64 x 64 output tile per workgroup, K staged 16 deep through LDS,
// 4 x 4 register tile per lane; both operands are read along K (unit stride) and stored transposed in
// LDS (+1 padding) so that the inner product reads are conflict-free.
template <typename T>
__global__ void __launch_bounds__(256) k_gemm_nt(const T* __restrict__ X, const T* __restrict__ W, T* __restrict__ C,
                                                 int M, int N, int K, int ldx, int ldw, int ldc) {
    constexpr int BM = 64, BN = 64, BK = 16;
    __shared__ T xs[BK][BM + 1];
    __shared__ T ws[BK][BN + 1];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    T acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (T)0;

    for (int k0 = 0; k0 < K; k0 += BK) {
        // 64 rows x 16 k: 1024 elements per operand, 4 per lane; lanes run along k (contiguous)
        for (int q = threadIdx.x; q < BM * BK; q += 256) {
            const int r = q / BK, kk = q % BK;
            const int gm = m0 + r, gn = n0 + r, gk = k0 + kk;
            xs[kk][r] = (gm < M && gk < K) ? X[(size_t)gm * ldx + gk] : (T)0;
            ws[kk][r] = (gn < N && gk < K) ? W[(size_t)gn * ldw + gk] : (T)0;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < BK; ++kk) {
            T a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = xs[kk][ty * 4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = ws[kk][tx * 4 + j];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] += a[i] * b[j];
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int gm = m0 + ty * 4 + i;
        if (gm >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int gn = n0 + tx * 4 + j;
            if (gn < N) C[(size_t)gm * ldc + gn] = acc[i][j];
        }
    }
}

template <typename T>
int launch_gemm_nt(const T* X, const T* W, T* C, int M, int N, int K, int ldx, int ldw, int ldc, hipStream_t st) {
    dim3 grid(cdiv(N, 64), cdiv(M, 64));
    hipLaunchKernelGGL(k_gemm_nt<T>, grid, dim3(256), 0, st, X, W, C, M, N, K, ldx, ldw, ldc);
    AO_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// float32 MFMA path: v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain), split-K.
// M = n_env is small (256) and W is shared by every env, so a plain 64 x 64 tiling leaves most CUs idle:
// the K range is cut into `splits` slices (gridDim.z) and every slice writes its own partial slab
// Cpart[z][M][N]; the consumer kernel adds the slabs in slice order, which keeps the result bitwise
// reproducible (no float atomics).  Workgroup = 4 waves as 2 (M) x 2 (N), one 32 x 32 accumulator each.
// LDS tiles are [64][BK + 1] floats: lane l of a wave reads row (l & 31), k = 2 kk + (l >> 5), i.e. the 32
// rows of a half-wave fall in 32 different banks (row stride 33).
// ---------------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));

// xsplits > 1: X is itself a stack of split-K slabs X[z][M][ldx] (z < xsplits, xslab floats apart) of an earlier product; they are
// summed in slab order while the operand is staged (chained products, e.g. v = M2C (M s), need no reduction launch in between).
__device__ inline void gemm_nt_mfma_tile(const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ Cpart, int M,
                                         int N, int K, int ldx, int ldw, int kslice, int xsplits, size_t xslab, int bz) {
    constexpr int BM = 64, BN = 64, BK = 32, LDT = BK + 1;
    __shared__ float xs[BM * LDT];
    __shared__ float ws[BN * LDT];
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = bz * kslice, kend = min(K, kbeg + kslice);
    const int wave = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    const int wm = (wave >> 1) * 32, wn = (wave & 1) * 32;
    const int lr = lane & 31, lh = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const bool vec_ok = ((ldx | ldw) & 3) == 0;            // rows 16-byte aligned -> float4 staging loads

    // stage 64 x 32 of X and of W: thread t -> row t/8 (+32), k offset (t%8)*4.  The next slice's global loads are issued
    // before the matrix-core loop of the current one (register double buffer): with 2-4 waves per SIMD the HBM latency of a
    // load -> LDS -> MFMA sequence is otherwise exposed every slice (ELT-size reconstructor: K = 10^4, 134 -> see DESIGN.md).
    float4 xv[2], wv[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = (threadIdx.x >> 3) + pass * 32, kq = (threadIdx.x & 7) * 4;
            const int gk = k0 + kq;
            xv[pass] = make_float4(0.f, 0.f, 0.f, 0.f);
            wv[pass] = xv[pass];
            const int gm = m0 + r, gn = n0 + r;
            if (vec_ok && gk + 3 < kend) {
                if (gm < M) {
                    xv[pass] = *reinterpret_cast<const float4*>(X + (size_t)gm * ldx + gk);
                    for (int z = 1; z < xsplits; ++z) {
                        const float4 t = *reinterpret_cast<const float4*>(X + (size_t)z * xslab + (size_t)gm * ldx + gk);
                        xv[pass].x += t.x; xv[pass].y += t.y; xv[pass].z += t.z; xv[pass].w += t.w;
                    }
                }
                if (gn < N) wv[pass] = *reinterpret_cast<const float4*>(W + (size_t)gn * ldw + gk);
            } else {
                float* xp = reinterpret_cast<float*>(&xv[pass]);
                float* wp = reinterpret_cast<float*>(&wv[pass]);
                for (int d = 0; d < 4; ++d) {
                    if (gk + d < kend) {
                        if (gm < M) {
                            xp[d] = X[(size_t)gm * ldx + gk + d];
                            for (int z = 1; z < xsplits; ++z) xp[d] += X[(size_t)z * xslab + (size_t)gm * ldx + gk + d];
                        }
                        if (gn < N) wp[d] = W[(size_t)gn * ldw + gk + d];
                    }
                }
            }
        }
    };
    if (kbeg < kend) fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += BK) {
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
            const int r = (threadIdx.x >> 3) + pass * 32, kq = (threadIdx.x & 7) * 4;
            float* xd = xs + r * LDT + kq;
            float* wd = ws + r * LDT + kq;
            xd[0] = xv[pass].x; xd[1] = xv[pass].y; xd[2] = xv[pass].z; xd[3] = xv[pass].w;
            wd[0] = wv[pass].x; wd[1] = wv[pass].y; wd[2] = wv[pass].z; wd[3] = wv[pass].w;
        }
        __syncthreads();
        if (k0 + BK < kend) fetch(k0 + BK);
        const float* xa = xs + (wm + lr) * LDT + lh;
        const float* wb = ws + (wn + lr) * LDT + lh;
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[2 * kk], wb[2 * kk], acc, 0, 0, 0);
        __syncthreads();
    }
    // C/D map of the 32x32 shape: column = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    float* C = Cpart + (size_t)bz * M * N;
    const int gn = n0 + wn + lr;
    if (gn < N) {
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int gm = m0 + wm + (rg & 3) + 8 * (rg >> 2) + 4 * lh;
            if (gm < M) C[(size_t)gm * N + gn] = acc[rg];
        }
    }
}

__global__ void __launch_bounds__(256) k_gemm_nt_mfma(const float* __restrict__ X, const float* __restrict__ W,
                                                      float* __restrict__ Cpart, int M, int N, int K, int ldx,
                                                      int ldw, int kslice, int xsplits, size_t xslab) {
    gemm_nt_mfma_tile(X, W, Cpart, M, N, K, ldx, ldw, kslice, xsplits, xslab, (int)blockIdx.z);
}

// The ring GEMM X = [Z | xi] [A | B]^T with the NEXT crossing's innovations drawn beside it: the z-slices beyond `splits` are
// not slices of the product but one workgroup per env running the layer's MT19937 stream one draw ahead (mt_normal_body,
// out of place: the caller commits the advanced stream at the next crossing).  Both kinds of workgroup are small -- a CU holds
// several of each -- so the 9 us of the Gaussian draw, serial in front of the product before, now hide behind it.
__global__ void __launch_bounds__(256) k_ring_gemm_draw_ahead(const float* __restrict__ X, const float* __restrict__ W,
                                                              float* __restrict__ Cpart, int M, int N, int K, int ldx, int ldw,
                                                              int kslice, int splits, const MtAhead m) {
    // the draws first (lowest workgroup ids: dispatched first): they are the longer of the two kinds
    const int draw_slices = (int)gridDim.z - splits;
    if ((int)blockIdx.z >= draw_slices) {
        gemm_nt_mfma_tile(X, W, Cpart, M, N, K, ldx, ldw, kslice, 1, 0, (int)blockIdx.z - draw_slices);
        return;
    }
    const int e = ((int)blockIdx.z * (int)gridDim.y + (int)blockIdx.y) * (int)gridDim.x + (int)blockIdx.x;
    if (e < m.n_env) mt_normal_body<float>(m.mt_in, m.pos_in, m.mt_out, m.pos_out, m.zx_out, m.K, m.n_inner, m.n_outer, e);
}

int launch_ring_gemm_draw_ahead(const float* X, const float* W, float* Cpart, int M, int N, int K, int splits, const MtAhead& m,
                                hipStream_t st) {
    if (m.n_outer % 2) return fail("mt_normal: n_outer=%d must be even", m.n_outer);
    const int kslice = cdiv(cdiv(K, splits), 32) * 32;
    const int gx = cdiv(N, 64), gy = cdiv(M, 64);
    dim3 grid(gx, gy, splits + cdiv(m.n_env, gx * gy));
    hipLaunchKernelGGL(k_ring_gemm_draw_ahead, grid, dim3(256), 0, st, X, W, Cpart, M, N, K, K, K, kslice, splits, m);
    AO_HIP(hipGetLastError());
    return 0;
}

int gemm_splits(int M, int N, int K) {
    const int tiles = cdiv(M, 64) * cdiv(N, 64);
    if (K > 2048) {
        // long reductions (ELT-size reconstructor): the K range is cut so that the workgroups come in whole rounds of the
        // 256 CUs with >= 4 resident per CU.  cost = rounds x (slice length + fixed per-workgroup part) + the slab traffic
        // (one write and one read of M N floats per slice, in units of one k step of one round ~ 20 ns)
        int best = 1;
        double best_cost = 1e300;
        for (int s = 1; s <= kMaxSplits; ++s) {
            const int wgs = tiles * s, kslice = cdiv(cdiv(K, s), 32) * 32;
            if (wgs < 1024 && s < kMaxSplits && tiles * (s + 1) <= 1024) continue;      // too few to hide the HBM latency
            const double cost = (double)cdiv(wgs, 256) * (kslice + 96) + (double)s * M * N * 8.0 / 4e12 / 20e-9;
            if (cost < best_cost) { best_cost = cost; best = s; }
        }
        return best;
    }
    // short reductions (the ring extrusion of the 8 m geometries): the split count -- and with it the order in which a ring value is
    // summed -- does not depend on M, so an env's trajectory is the same bit for bit in a shard of 1, 256 or 4096 envs
    (void)tiles;
    const int smax = K / 64 > 0 ? K / 64 : 1;
    return smax > kMaxSplits ? kMaxSplits : smax;
}

// t[i] = slab 0 + slab 1 + ... (in that order: the order in which a chained product sums them while staging, gemm_nt_mfma_tile), in place
// in slab 0.  A chained product re-sums the slabs in every workgroup that reads the tile -- 82 column tiles at the ELT size: 1 GB of L2
// reads for 10 MB of slabs -- so a long chain is summed once, here, first.
__global__ void __launch_bounds__(256) k_sum_slabs(float* __restrict__ x, int n4, int slabs, size_t slab4) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    float4* p = reinterpret_cast<float4*>(x);
    float4 v[kMaxSplits];
#pragma unroll
    for (int z = 0; z < kMaxSplits; ++z) v[z] = p[(size_t)(z < slabs ? z : 0) * slab4 + i];
    float4 a = v[0];
#pragma unroll
    for (int z = 1; z < kMaxSplits; ++z)
        if (z < slabs) { a.x += v[z].x; a.y += v[z].y; a.z += v[z].z; a.w += v[z].w; }
    p[i] = a;
}
int launch_sum_slabs(float* x, size_t n, int slabs, hipStream_t st) {
    if (n % 4) return fail("sum_slabs: %zu elements per slab, not a multiple of 4", n);
    hipLaunchKernelGGL(k_sum_slabs, dim3(cdiv((int)(n / 4), 256)), dim3(256), 0, st, x, (int)(n / 4), slabs, n / 4);
    AO_HIP(hipGetLastError());
    return 0;
}

int launch_gemm_nt_mfma(const float* X, const float* W, float* Cpart, int M, int N, int K, int ldx, int ldw, int splits,
                        hipStream_t st, int xsplits, size_t xslab) {
    const int kslice = cdiv(cdiv(K, splits), 32) * 32;
    dim3 grid(cdiv(N, 64), cdiv(M, 64), splits);
    hipLaunchKernelGGL(k_gemm_nt_mfma, grid, dim3(256), 0, st, X, W, Cpart, M, N, K, ldx, ldw, kslice, xsplits, xslab);
    AO_HIP(hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------------
// Step epilogue, one workgroup per env   (MAIN/OOPAOEnv/OOPAOEnv.py:491, 508-518, 536):
//   obs_img = vec_to_img(-v) * 1e6 ; reward = -||obs_img||_2
//   dm.coefs = dm_prev * leak + img_to_vec(action) * 1e-6 ; dm_prev = dm.coefs          (do_integrate)
// With gain_from_obs != 0 the action is the integrator command gain * obs_img of the PREVIOUS
// observation held in `obs` (closed-loop driver MAIN/integrator_oopao_razor.py:70-71); the previous
// image is consumed before it is overwritten.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(1024) k_recon_finish(const FinishArgs<T> f) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* img_s = reinterpret_cast<T*>(lds_raw);          // the full image, zero at non-actuators (vec_to_img)
    __shared__ double red[16];
    __shared__ double tel[4];
    const int e = blockIdx.x, n_env = gridDim.x;
    const int img = f.n_act * f.n_act;
    T* ob = f.obs + (size_t)e * img;
    const size_t slab = (size_t)n_env * f.n_valid_act;
    const T* vv = f.v + (size_t)e * f.n_valid_act;
    for (int q = threadIdx.x; q < img; q += blockDim.x) img_s[q] = (T)0;
    __syncthreads();
    double ss = 0.0;
    for (int k = threadIdx.x; k < f.n_valid_act; k += blockDim.x) {
        const int px = f.act_idx[k];
        if (f.do_integrate) {
            const T a = (f.gain_from_obs != (T)0) ? f.gain_from_obs * ob[px] : f.action[(size_t)e * img + px];
            const size_t ck = (size_t)e * f.n_valid_act + k;
            // img_to_vec(action)*1e-6 (OOPAOEnv.py:491): the wrappers hand over float32 actions and NumPy keeps
            // float32 for array * python-float, so the increment is a float32 product; a float64 action that is
            // not float32-representable (env driven without the torch wrapper) keeps the float64 product.
            const float af = (float)a;
            const T inc = ((T)af == a) ? (T)(af * 1e-6f) : a * (T)1e-6;
            const T cn = f.dm_prev[ck] * f.leak + inc;           // dm.coefs = dm_prev * leak + action ; dm_prev = dm.coefs  (:508-509)
            f.coefs[ck] = cn;
            f.dm_prev[ck] = cn;
        }
        T acc = vv[k];
        for (int z = 1; z < f.splits; ++z) acc += vv[(size_t)z * slab + k];      // split-K slabs, fixed order
        const T o = -acc * (T)1e6;
        img_s[px] = o;
        ss += (double)o * (double)o;
    }
    // per-tile telemetry sums of the phase kernel: the last wave adds them, lane t the tiles t, t + 64, ... and then a
    // fixed shuffle tree (one lane walking 120 tiles of an ELT pupil was a chain of 480 memory latencies = 40 us)
    double tv[4] = {0, 0, 0, 0};
    const bool tel_wave = threadIdx.x / kWave == blockDim.x / kWave - 1;
    if (tel_wave) {
        const double* pp = f.part + (size_t)e * f.n_tiles * 4;
        for (int t = threadIdx.x & (kWave - 1); t < f.n_tiles; t += kWave)
#pragma unroll
            for (int k = 0; k < 4; ++k) tv[k] += pp[t * 4 + k];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            for (int off = 32; off > 0; off >>= 1) tv[k] += __shfl_down(tv[k], off);
        if ((threadIdx.x & (kWave - 1)) == 0)
            for (int k = 0; k < 4; ++k) tel[k] = tv[k];
    }
    __syncthreads();
    for (int q = threadIdx.x; q < img; q += blockDim.x) ob[q] = img_s[q];
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = ss;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0;
        for (int q = 0; q < (int)blockDim.x / kWave; ++q) tot += red[q];
        if (f.reward) f.reward[e] = (T)(-sqrt(tot));
        if (f.ret && f.do_integrate) f.ret[e] += (T)(-sqrt(tot));
        // telemetry from the phase kernel's per-tile sums (fixed order): std(OPD[pupil]) * 1e9, exp(-var(phase[pupil]))
        const double v[4] = {tel[0], tel[1], tel[2], tel[3]};
        const double n = (double)f.n_pupil;
        double var_atm = v[1] / n - (v[0] / n) * (v[0] / n);
        double var_res = v[3] / n - (v[2] / n) * (v[2] / n);
        var_atm = var_atm > 0 ? var_atm : 0;
        var_res = var_res > 0 ? var_res : 0;
        const double total = sqrt(var_atm) * 1e9, resid = sqrt(var_res) * 1e9;
        const double sr = exp(-var_res * f.src_scale * f.src_scale);
        T* sc = f.scal + 4 * e;
        sc[0] = (T)total;
        sc[1] = (T)resid;
        sc[2] = (T)sr;
        if (f.strehl) f.strehl[e] = (T)sr;
        if (f.telemetry_index >= 0) {
            const size_t o = (size_t)f.telemetry_index * n_env + e;
            f.total[o] = (T)total;
            f.residual[o] = (T)resid;
        }
    }
}

template <typename T>
int launch_recon_finish(const FinishArgs<T>& fa, int n_env, hipStream_t st) {
    // one workgroup per env; ELT-size DMs (thousands of actuators, each a short chain of dependent loads) get 1024 lanes
    hipLaunchKernelGGL(k_recon_finish<T>, dim3(n_env), dim3(fa.n_valid_act > 1024 ? 1024 : 256),
                       (size_t)fa.n_act * fa.n_act * sizeof(T), st, fa);
    AO_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ void k_convert(const double* __restrict__ src, T* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = (T)src[i];
}
template <typename T>
int launch_convert_from_f64(const double* src, T* dst, size_t n, hipStream_t st) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(k_convert<T>, dim3(blocks ? blocks : 1), dim3(256), 0, st, src, dst, n);
    AO_HIP(hipGetLastError());
    return 0;
}

#define INST(T)                                                                                        \
    template int launch_gemm_nt<T>(const T*, const T*, T*, int, int, int, int, int, int, hipStream_t); \
    template int launch_recon_finish<T>(const FinishArgs<T>&, int, hipStream_t);                       \
    template int launch_convert_from_f64<T>(const double*, T*, size_t, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace ao
