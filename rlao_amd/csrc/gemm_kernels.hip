// Batched contractions over the env dimension and the small per-env epilogues.
//   X = Z A^T + xi B^T      OOPAO/Atmosphere.py:308      (one product with the operands concatenated: [Z|xi] [A|B]^T)
//   v = R s                 MAIN/OOPAOEnv/OOPAOEnv.py:517
//   dm.OPD = modes @ coefs  OOPAO/DeformableMirror.py:556 (dense-DM path)
// All three are "NT" products  C[M][N] = X[M][K] . W[N][K]^T  with M = n_env and W shared by every env.
#include "common.hpp"

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
// Step epilogue, one workgroup per env   (MAIN/OOPAOEnv/OOPAOEnv.py:491, 508-518, 536):
//   obs_img = vec_to_img(-v) * 1e6 ; reward = -||obs_img||_2
//   dm.coefs = dm.coefs * leak + img_to_vec(action) * 1e-6          (do_integrate)
// With gain_from_obs != 0 the action is the integrator command gain * obs_img of the PREVIOUS
// observation held in `obs` (closed-loop driver MAIN/integrator_oopao_razor.py:70-71); the previous
// image is consumed before it is overwritten.
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) k_recon_finish(const T* __restrict__ v, const int* __restrict__ act_idx,
                                                      const T* __restrict__ action, T* __restrict__ coefs,
                                                      T* __restrict__ obs, T* __restrict__ reward, int n_act,
                                                      int n_valid_act, T leak, int do_integrate, T gain_from_obs) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* img_s = reinterpret_cast<T*>(lds_raw);          // the full image, zero at non-actuators (vec_to_img)
    __shared__ double red[4];
    const int e = blockIdx.x;
    const int img = n_act * n_act;
    T* ob = obs + (size_t)e * img;
    const T* vv = v + (size_t)e * n_valid_act;
    for (int q = threadIdx.x; q < img; q += blockDim.x) img_s[q] = (T)0;
    __syncthreads();
    double ss = 0.0;
    for (int k = threadIdx.x; k < n_valid_act; k += blockDim.x) {
        const int px = act_idx[k];
        if (do_integrate) {
            const T a = (gain_from_obs != (T)0) ? gain_from_obs * ob[px] : action[(size_t)e * img + px];
            T* c = coefs + (size_t)e * n_valid_act + k;
            // img_to_vec(action)*1e-6 (OOPAOEnv.py:491): the wrappers hand over float32 actions and NumPy keeps
            // float32 for array * python-float, so the increment is a float32 product; a float64 action that is
            // not float32-representable (env driven without the torch wrapper) keeps the float64 product.
            const float af = (float)a;
            const T inc = ((T)af == a) ? (T)(af * 1e-6f) : a * (T)1e-6;
            *c = (*c) * leak + inc;
        }
        const T o = -vv[k] * (T)1e6;
        img_s[px] = o;
        ss += (double)o * (double)o;
    }
    __syncthreads();
    for (int q = threadIdx.x; q < img; q += blockDim.x) ob[q] = img_s[q];
    for (int off = 32; off > 0; off >>= 1) ss += __shfl_down(ss, off);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = ss;
    __syncthreads();
    if (threadIdx.x == 0 && reward) reward[e] = (T)(-sqrt(red[0] + red[1] + red[2] + red[3]));
}

template <typename T>
int launch_recon_finish(const T* v, const int* act_idx, const T* action, T* coefs, T* obs, T* reward, int n_env,
                        int n_act, int n_valid_act, double leak, int do_integrate, double gain_from_obs,
                        hipStream_t st) {
    hipLaunchKernelGGL(k_recon_finish<T>, dim3(n_env), dim3(256), (size_t)n_act * n_act * sizeof(T), st, v, act_idx, action, coefs, obs, reward,
                       n_act, n_valid_act, (T)leak, do_integrate, (T)gain_from_obs);
    AO_HIP(hipGetLastError());
    return 0;
}

template <typename T>
__global__ void k_copy_scal(const T* __restrict__ scal, T* __restrict__ d_strehl, int n_env) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n_env) d_strehl[e] = scal[4 * e + 2];
}
template <typename T>
int launch_copy_scal(const T* scal, T* d_strehl, int n_env, hipStream_t st) {
    hipLaunchKernelGGL(k_copy_scal<T>, dim3(cdiv(n_env, 256)), dim3(256), 0, st, scal, d_strehl, n_env);
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

#define INST(T)                                                                                                  \
    template int launch_gemm_nt<T>(const T*, const T*, T*, int, int, int, int, int, int, hipStream_t);           \
    template int launch_recon_finish<T>(const T*, const int*, const T*, T*, T*, T*, int, int, int, double, int,  \
                                        double, hipStream_t);                                                    \
    template int launch_copy_scal<T>(const T*, T*, int, hipStream_t);                                            \
    template int launch_convert_from_f64<T>(const double*, T*, size_t, hipStream_t);
INST(float)
INST(double)
#undef INST

}  // namespace ao
