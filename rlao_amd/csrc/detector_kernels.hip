// Camera noise applied to a whole WFS frame in place (detector.hpp); used by every measurement path that leaves its
// frame in HBM: the separate Shack-Hartmann kernels, the Pyramid, the calibration shards.  One lane per QUAD of 4 pixels
// (detector.hpp, "Stream layout"): Shack-Hartmann frames with 6-pixel lenslets use the quads of the fused step kernel, so
// that both paths draw the same noise; every other frame 4 consecutive pixels of a row.
#include "detector.hpp"
#include "sh_device.hpp"
#include "camera_sh6.hpp"

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

// Shack-Hartmann frames with 6-pixel lenslets, float32: the queue-based camera of the fused step kernel (camera_sh6.hpp), one lane
// per (lenslet, column pair) = 12 pixels, 4 waves per workgroup.  Same streams, same samplers: the same frame as k_detector, but
// the undecided bright pixels of a workgroup are finished densely instead of wave by wave in lock-step (ELT frames, 480^2 pixels
// x 512 envs: 1.39 ms per frame with k_detector -- more than the phase and the spots kernels).
__global__ void __launch_bounds__(256) k_detector_sh6(float* __restrict__ frame, float* __restrict__ wfs_max,
                                                      const uint8_t* __restrict__ valid2d, int cam, int n_subap, const DetectorCfg d) {
    constexpr int WAVES = 4, Q0_CAP = 1024, Q1_CAP = 512;
    __shared__ f32x4d q0[Q0_CAP];
    __shared__ float res[Q0_CAP];
    __shared__ float q1[3 * Q1_CAP];
    __shared__ int counts[WAVES + 1];
    const int e = blockIdx.y, tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    const int task = blockIdx.x * 256 + tid, n_tasks = n_subap * n_subap * 3;
    const bool live = task < n_tasks;
    const int k = (live ? task : 0) / 3, q3 = (live ? task : 0) - 3 * k;
    const int li = k / n_subap, lj = k - li * n_subap;
    const uint32_t px0 = (uint32_t)((li * 6) * cam + lj * 6 + q3);
    float* fr = frame + (size_t)e * cam * cam + px0;
    if (tid < WAVES + 1) counts[tid] = 0;
    f32x16s pxv;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
        pxv[u] = live ? fr[(size_t)u * cam] : 0.f;
        pxv[6 + u] = live ? fr[(size_t)u * cam + 3] : 0.f;
    }
#pragma unroll
    for (int u = 12; u < 16; ++u) pxv[u] = 0.f;
    __syncthreads();
    const CameraLds Q{q0, Q0_CAP, res, q1, Q1_CAP, counts};
    camera_sh6_lane<WAVES>(pxv, live, px0, cam, (uint32_t)e, d, Q, w, lane);
    float mx = 0.f;
    if (live) {
        const bool lit = valid2d[k] != 0;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            fr[(size_t)u * cam] = pxv[u];
            fr[(size_t)u * cam + 3] = pxv[6 + u];
            if (lit) {
                mx = pxv[u] > mx ? pxv[u] : mx;
                mx = pxv[6 + u] > mx ? pxv[6 + u] : mx;
            }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float o = __shfl_down(mx, off);
        mx = o > mx ? o : mx;
    }
    if (lane == 0) atomic_max_nonneg(&wfs_max[e], mx);            // noisy counts can be negative: max(., 0)
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
    if constexpr (sizeof(T) == 4) {
        if (sh6) {
            hipLaunchKernelGGL(k_detector_sh6, dim3(cdiv(n_subap * n_subap * 3, 256), n_env), dim3(256), 0, st, frame, wfs_max, valid2d, cam,
                               n_subap, d);
            AO_HIP(hipGetLastError());
            return 0;
        }
    }
    const int n_quads = sh6 ? n_subap * n_subap * 9 : cam * ((cam + 3) / 4);
    const int chunks = cdiv(n_quads, 256 * 2) < 1 ? 1 : cdiv(n_quads, 256 * 2);
    hipLaunchKernelGGL(k_detector<T>, dim3(chunks, n_env), dim3(256), 0, st, frame, wfs_max, valid2d, cam, n_subap, d, n_quads, sh6);
    AO_HIP(hipGetLastError());
    return 0;
}

template int launch_detector<float>(float*, float*, const uint8_t*, int, int, int, const DetectorCfg&, hipStream_t);
template int launch_detector<double>(double*, double*, const uint8_t*, int, int, int, const DetectorCfg&, hipStream_t);

}  // namespace ao
