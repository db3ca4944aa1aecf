// Episode reset on the device: atm.generateNewPhaseScreen(seed) for every env of the shard.
//   OOPAO/Atmosphere.py:560-592   per layer: RandomState(seed + layer) -> ft_sh_phase_screen; ring RandomState(seed + 1000 layer)
//   OOPAO/phaseStats.py:190-235   ft_phase_screen : cn = (N(0,1) + i N(0,1)) sqrt(PSD) del_f ; fftshift(fft2(fftshift(cn))).real
//   OOPAO/phaseStats.py:243-318   ft_sh_phase_screen: + 3 sub-harmonic 3x3 grids (only the i, j in {0, 1} corner is summed),
//                                 mean-removed; both generators are seeded with the SAME seed (:268, :272), so the
//                                 sub-harmonic draws are the first 54 normals of the high-frequency stream
// Everything here runs in float64 whatever the shard's dtype (it is once per episode and must agree with the NumPy
// generator to ~1e-12 so that a float32 shard starts from the same screens as the reference); only the final
// store converts.  The normals come from k_mt_normal (atm_kernels.hip): same MT19937 + polar stream as NumPy.
#include "common.hpp"
#include "fft.hpp"

namespace ao {


// rows: cn with the input fftshift folded into the load, FFT along x
__global__ void __launch_bounds__(256) k_screen_rows(const ScreenArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int N = a.N, RB = a.seq_per_block, h = N / 2;
    cx<double>* A = reinterpret_cast<cx<double>*>(lds_raw);
    const int NP = a.plan.np;
    cx<double>* B = A + RB * NP;
    cx<double>* twl = B + RB * NP;
    fft_load_twiddles<double>(twl, a.tw, N);
    const int e = blockIdx.y, y0 = blockIdx.x * RB;
    const int nrow = min(RB, N - y0);
    const double* re = a.nrm + (size_t)e * 2 * N * N;
    const double* im = re + (size_t)N * N;
    for (int i = threadIdx.x; i < RB * N; i += blockDim.x) {
        const int r = i / N, x = i - r * N;
        cx<double> v = {0, 0};
        if (r < nrow) {
            const int ys = (y0 + r + N - h) % N, xs = (x + N - h) % N;   // fftshift(cn)[y][x] = cn[(y - N//2) % N][(x - N//2) % N] (odd N too)
            const size_t q = (size_t)ys * N + xs;
            const double w = a.amp[q];
            v = {re[q] * w, im[q] * w};
        }
        A[r * NP + fpad(x)] = v;
    }
    cx<double>* out = fft_lds<double>(A, B, a.plan, RB, twl, 0);
    cx<double>* dst = a.scratch + ((size_t)e * N + y0) * N;
    for (int i = threadIdx.x; i < nrow * N; i += blockDim.x) {
        const int r = i / N;
        dst[i] = out[r * NP + fpad(i - r * N)];
    }
}

// columns: FFT along y, output fftshift, real part
__global__ void __launch_bounds__(256) k_screen_cols(const ScreenArgs a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int N = a.N, CB = a.seq_per_block, h = N / 2;
    cx<double>* A = reinterpret_cast<cx<double>*>(lds_raw);
    const int NP = a.plan.np;
    cx<double>* B = A + CB * NP;
    cx<double>* twl = B + CB * NP;
    fft_load_twiddles<double>(twl, a.tw, N);
    const int e = blockIdx.y, x0 = blockIdx.x * CB;
    const int ncol = min(CB, N - x0);
    const cx<double>* src = a.scratch + (size_t)e * N * N;
    for (int i = threadIdx.x; i < N * CB; i += blockDim.x) {
        const int y = i / CB, c = i - y * CB;
        A[c * NP + fpad(y)] = c < ncol ? src[(size_t)y * N + x0 + c] : cx<double>{0, 0};
    }
    cx<double>* out = fft_lds<double>(A, B, a.plan, CB, twl, 0);
    double* hi = a.hi + (size_t)e * N * N;
    for (int i = threadIdx.x; i < N * CB; i += blockDim.x) {
        const int y = i / CB, c = i - y * CB;
        if (c < ncol) hi[(size_t)((y + h) % N) * N + (x0 + c + h) % N] = out[c * NP + fpad(y)].re;
    }
}

// sub-harmonics + mean removal + sum, one workgroup per env; writes layer.phase into the interior of mapShift
template <typename T>
__global__ void __launch_bounds__(1024) k_screen_finish(const ScreenArgs a, T* __restrict__ map, int S) {
    __shared__ double red[16];
    __shared__ double cr[12], ci[12];
    const int N = a.N, e = blockIdx.x;
    const double* nrm = a.nrm + (size_t)e * 2 * N * N;
    if (threadIdx.x < 12) {
        // draws of grid p: normal(size=(3,3)) real parts, then (3,3) imaginary parts, row-major (i, j)
        const int p = threadIdx.x / 4, ij = threadIdx.x % 4, i = ij / 2, j = ij % 2;
        const double amp = a.sub[3 * threadIdx.x];
        cr[threadIdx.x] = nrm[18 * p + 3 * i + j] * amp;
        ci[threadIdx.x] = nrm[18 * p + 9 + 3 * i + j] * amp;
    }
    __syncthreads();
    const double two_pi = 6.283185307179586476925286766559;
    double s = 0;
    double* hi = a.hi + (size_t)e * N * N;
    // pass 1: low-frequency part (kept in `hi` scratch as a second plane is not needed: lo is recomputed in pass 2)
    for (int q = threadIdx.x; q < N * N; q += blockDim.x) {
        const int y = q / N, x = q - y * N;
        const double xc = (x - N / 2.0) * a.delta, yc = (y - N / 2.0) * a.delta;
        double lo = 0;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            double sn, cs;
            sincos(two_pi * (a.sub[3 * t + 1] * xc + a.sub[3 * t + 2] * yc), &sn, &cs);
            lo += cr[t] * cs - ci[t] * sn;                         // Re(cn exp(i 2 pi (fx x + fy y)))
        }
        s += lo;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = s;
    __syncthreads();
    double mean = 0;
    for (int w = 0; w < (int)blockDim.x / kWave; ++w) mean += red[w];
    mean /= (double)N * N;
    T* m = map + (size_t)e * S * S;
    for (int q = threadIdx.x; q < N * N; q += blockDim.x) {
        const int y = q / N, x = q - y * N;
        const double xc = (x - N / 2.0) * a.delta, yc = (y - N / 2.0) * a.delta;
        double lo = 0;
#pragma unroll
        for (int t = 0; t < 12; ++t) {
            double sn, cs;
            sincos(two_pi * (a.sub[3 * t + 1] * xc + a.sub[3 * t + 2] * yc), &sn, &cs);
            lo += cr[t] * cs - ci[t] * sn;
        }
        m[(size_t)(y + 1) * S + x + 1] = (T)((lo - mean) + hi[q]);
    }
}

template <typename T>
int launch_screen(const ScreenArgs& base, T* map, int S, hipStream_t st) {
    ScreenArgs a = base;
    const int N = a.N;
    const int NP = a.plan.np;
    int rb = (int)(60 * 1024 / (2 * (size_t)NP * sizeof(cx<double>)));
    if (rb < 1) return fail("screen generator: N = %d does not fit two LDS row buffers", N);
    rb = rb > 8 ? 8 : rb;
    a.seq_per_block = rb;
    const size_t lds = (2 * (size_t)rb * NP + N) * sizeof(cx<double>);
    if (lds > 64 * 1024) {
        AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_screen_rows), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_screen_cols), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    }
    hipLaunchKernelGGL(k_screen_rows, dim3(cdiv(N, rb), a.n_env), dim3(256), lds, st, a);
    hipLaunchKernelGGL(k_screen_cols, dim3(cdiv(N, rb), a.n_env), dim3(256), lds, st, a);
    hipLaunchKernelGGL(k_screen_finish<T>, dim3(a.n_env), dim3(1024), 0, st, a, map, S);
    AO_HIP(hipGetLastError());
    return 0;
}

template int launch_screen<float>(const ScreenArgs&, float*, int, hipStream_t);
template int launch_screen<double>(const ScreenArgs&, double*, int, hipStream_t);

}  // namespace ao
