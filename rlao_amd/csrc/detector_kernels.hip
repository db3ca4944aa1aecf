// Camera noise applied to a whole WFS frame in place (detector.hpp); used by every measurement path that leaves its
// frame in HBM: the separate Shack-Hartmann kernels, the Pyramid, the calibration shards.
#include "detector.hpp"
#include "sh_device.hpp"

namespace ao {

template <typename T>
__global__ void __launch_bounds__(256) k_detector(T* __restrict__ frame, T* __restrict__ wfs_max,
                                                  const uint8_t* __restrict__ valid2d, int cam, int n_subap,
                                                  const DetectorCfg d) {
    const int e = blockIdx.y;
    T* fr = frame + (size_t)e * cam * cam;
    const int p = valid2d ? cam / n_subap : 1;
    T mx = 0;
    for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < cam * cam; q += gridDim.x * blockDim.x) {
        const float v = detector_pixel((float)fr[q], d, (uint32_t)q, (uint32_t)e);
        fr[q] = (T)v;
        if (valid2d) {                                            // threshold reference: maximum over the valid spots
            const int r = q / cam, c = q - r * cam;
            if (valid2d[(r / p) * n_subap + c / p]) mx = (T)v > mx ? (T)v : mx;
        }
    }
    if (valid2d) {
        for (int off = 32; off > 0; off >>= 1) {
            const T o = __shfl_down(mx, off);
            mx = o > mx ? o : mx;
        }
        if ((threadIdx.x & (kWave - 1)) == 0) atomic_max_nonneg(&wfs_max[e], mx);      // noisy counts can be negative: max(., 0)
    }
}

template <typename T>
__global__ void k_zero(T* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = (T)0;
}

template <typename T>
int launch_detector(T* frame, T* wfs_max, const uint8_t* valid2d, int n_env, int cam, int n_subap, const DetectorCfg& d,
                    hipStream_t st) {
    if (!d.active) return 0;
    if (valid2d) hipLaunchKernelGGL(k_zero<T>, dim3(cdiv(n_env, 256)), dim3(256), 0, st, wfs_max, n_env);
    const int chunks = cdiv(cam * cam, 256 * 4) < 1 ? 1 : cdiv(cam * cam, 256 * 4);
    hipLaunchKernelGGL(k_detector<T>, dim3(chunks, n_env), dim3(256), 0, st, frame, wfs_max, valid2d, cam, n_subap, d);
    AO_HIP(hipGetLastError());
    return 0;
}

template int launch_detector<float>(float*, float*, const uint8_t*, int, int, int, const DetectorCfg&, hipStream_t);
template int launch_detector<double>(double*, double*, const uint8_t*, int, int, int, const DetectorCfg&, hipStream_t);

}  // namespace ao
