// Camera noise applied to a whole WFS frame in place (detector.hpp); used by every measurement path that leaves its
// frame in HBM: the separate Shack-Hartmann kernels, the Pyramid, the calibration shards.  One lane per QUAD of 4 pixels
// (detector.hpp, "Stream layout"): Shack-Hartmann frames with 6-pixel lenslets use the quads of the fused step kernel, so
// that both paths draw the same noise; every other frame 4 consecutive pixels of a row.
#include "detector.hpp"
#include "sh_device.hpp"

namespace ao {

template <typename T>
__global__ void __launch_bounds__(256) k_detector(T* __restrict__ frame, T* __restrict__ wfs_max,
                                                  const uint8_t* __restrict__ valid2d, int cam, int n_subap,
                                                  const DetectorCfg d, int n_quads, int sh6) {
    const int e = blockIdx.y;
    T* fr = frame + (size_t)e * cam * cam;
    const float rtab = recip_table_lane();
    const int qpr = (cam + 3) / 4;                                // quads per frame row (generic layout)
    T mx = 0;
    // whole waves walk the quads: the samplers vote across the wave, every lane of a wave must reach them together
    const int stride = gridDim.x * blockDim.x;
    for (int q0 = blockIdx.x * blockDim.x; q0 < n_quads; q0 += stride) {
        const int q = q0 + threadIdx.x;
        const bool live = q < n_quads;
        uint32_t pix[4];
        bool in[4] = {live, live, live, live};
        bool lit = false;                                         // inside a valid lenslet (threshold reference)
        if (sh6) {
            const int k = (live ? q : 0) / 9, j = (live ? q : 0) - 9 * k;
            const int li = k / n_subap, lj = k - li * n_subap;
            sh6_quad_pixels(j, li * 6, lj * 6, cam, pix);
            lit = live && valid2d && valid2d[k];
        } else {
            const int r = (live ? q : 0) / qpr, c4 = 4 * ((live ? q : 0) - r * qpr);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                pix[s] = (uint32_t)(r * cam + c4 + s);
                in[s] = live && c4 + s < cam;
            }
            if (valid2d && live) {
                const int p = cam / n_subap;
                lit = valid2d[(r / p) * n_subap + c4 / p] != 0;  // (a quad never straddles lenslets it matters for: p % 4 == 0 or sh6)
            }
        }
        f32x4d v;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[s] = in[s] ? (float)fr[pix[s]] : 0.f;
        detector_quad<true>(v, pix, pix[0], (uint32_t)e, d, rtab);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (in[s]) fr[pix[s]] = (T)v[s];
            if (in[s] && lit) mx = (T)v[s] > mx ? (T)v[s] : mx;
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
    const int sh6 = (valid2d && n_subap > 0 && cam == 6 * n_subap) ? 1 : 0;
    if (valid2d && !sh6 && (cam / n_subap) % 4 != 0)
        return fail("camera noise on a Shack-Hartmann frame needs 6 or a multiple of 4 pixels per lenslet, got %d", cam / n_subap);
    const int n_quads = sh6 ? n_subap * n_subap * 9 : cam * ((cam + 3) / 4);
    const int chunks = cdiv(n_quads, 256 * 2) < 1 ? 1 : cdiv(n_quads, 256 * 2);
    hipLaunchKernelGGL(k_detector<T>, dim3(chunks, n_env), dim3(256), 0, st, frame, wfs_max, valid2d, cam, n_subap, d, n_quads, sh6);
    AO_HIP(hipGetLastError());
    return 0;
}

template int launch_detector<float>(float*, float*, const uint8_t*, int, int, int, const DetectorCfg&, hipStream_t);
template int launch_detector<double>(double*, double*, const uint8_t*, int, int, int, const DetectorCfg&, hipStream_t);

}  // namespace ao
