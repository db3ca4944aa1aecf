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
                                                  const DetectorCfg d, const PoissonAlias pa, int n_quads, int sh6) {
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
        detector_quad<true>(v, pix, pix[0], (uint32_t)e, d, rtab, pa);
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

// Shack-Hartmann frames with 6-pixel lenslets, float32: the camera block of the fused step kernel (camera_sh6.hpp), one lane per
// (lenslet, column pair) = 12 pixels, with the alias tables of the photon draw in LDS (58 KB, copied by direct loads while the
// first pixels are on their way; a workgroup of 8 waves serves `chunks` runs of 512 lanes so that the copy is paid once per ~25 k
// pixels).  Same streams, same sampler as k_detector (whose table reads go through the caches): the same frame.
// (ELT frames, 480^2 pixels x 512 envs: 1.39 ms per frame with k_detector and NumPy's samplers, 1.03 ms with the round-2 queues.)
__global__ void __launch_bounds__(512) k_detector_sh6(float* __restrict__ frame, float* __restrict__ wfs_max,
                                                      const uint8_t* __restrict__ valid2d, int cam, int n_subap, const DetectorCfg d,
                                                      const PoissonAlias pa, int chunks) {
    constexpr int WAVES = 8;
    __shared__ __align__(16) uint32_t tab_s[palias::kMaxWords];
    const int e = blockIdx.y, tid = threadIdx.x, w = tid / kWave, lane = tid & (kWave - 1);
    const int n_tasks = n_subap * n_subap * 3;
    if (d.photon_noise) alias_table_to_lds(pa.tab, pa.words, tab_s, w, WAVES, lane);
    float mx = 0.f;
    for (int c = 0; c < chunks; ++c) {
        const int task0 = (blockIdx.x * chunks + c) * 512;
        if (task0 >= n_tasks) break;                                // (uniform over the workgroup)
        const int task = task0 + tid;
        const bool live = task < n_tasks;
        const int k = (live ? task : 0) / 3, q3 = (live ? task : 0) - 3 * k;
        const int li = k / n_subap, lj = k - li * n_subap;
        const uint32_t px0 = (uint32_t)((li * 6) * cam + lj * 6 + q3);
        float* fr = frame + (size_t)e * cam * cam + px0;
        float Ia[6], Ib[6];
#pragma unroll
        for (int u = 0; u < 6; ++u) {
            Ia[u] = live ? fr[(size_t)u * cam] : 0.f;
            Ib[u] = live ? fr[(size_t)u * cam + 3] : 0.f;
        }
        f32x16s pxv = camera_pack(Ia, Ib);
        if (c == 0) __syncthreads();                                // vmcnt(0) + barrier: every wave's share of the table has landed
        camera_sh6_lane(pxv, live, px0, cam, (uint32_t)e, d, tab_s, pa.lmax);
        camera_unpack(pxv, Ia, Ib);
        if (live) {
            const bool lit = valid2d[k] != 0;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                fr[(size_t)u * cam] = Ia[u];
                fr[(size_t)u * cam + 3] = Ib[u];
                if (lit) {
                    mx = Ia[u] > mx ? Ia[u] : mx;
                    mx = Ib[u] > mx ? Ib[u] : mx;
                }
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
                    const PoissonAlias& pa, hipStream_t st) {
    if (d.active && d.photon_noise && (!pa.tab || pa.lmax < palias::kCoarseStep)) return fail("camera: the photon-noise tables are missing");
    if (!d.active) return 0;
    if (valid2d) hipLaunchKernelGGL(k_zero<T>, dim3(cdiv(n_env, 256)), dim3(256), 0, st, wfs_max, n_env);
    const int sh6 = (valid2d && n_subap > 0 && cam == 6 * n_subap) ? 1 : 0;
    if (valid2d && !sh6 && (cam / n_subap) % 4 != 0)
        return fail("camera noise on a Shack-Hartmann frame needs 6 or a multiple of 4 pixels per lenslet, got %d", cam / n_subap);
    if constexpr (sizeof(T) == 4) {
        if (sh6) {
            const int n_runs = cdiv(n_subap * n_subap * 3, 512);
            int chunks = (int)(((size_t)n_runs * n_env) / 2048);   // >= 2048 workgroups where there is that much work, <= 8 runs each
            chunks = chunks < 1 ? 1 : (chunks > 8 ? 8 : chunks);
            PoissonAlias pl = pa;
            pl.words = pa.words < palias::kMaxWords ? pa.words : palias::kMaxWords;
            hipLaunchKernelGGL(k_detector_sh6, dim3(cdiv(n_runs, chunks), n_env), dim3(512), 0, st, frame, wfs_max, valid2d, cam, n_subap, d,
                               pl, chunks);
            AO_HIP(hipGetLastError());
            return 0;
        }
    }
    const int n_quads = sh6 ? n_subap * n_subap * 9 : cam * ((cam + 3) / 4);
    const int chunks = cdiv(n_quads, 256 * 2) < 1 ? 1 : cdiv(n_quads, 256 * 2);
    hipLaunchKernelGGL(k_detector<T>, dim3(chunks, n_env), dim3(256), 0, st, frame, wfs_max, valid2d, cam, n_subap, d, pa, n_quads, sh6);
    AO_HIP(hipGetLastError());
    return 0;
}

template int launch_detector<float>(float*, float*, const uint8_t*, int, int, int, const DetectorCfg&, const PoissonAlias&, hipStream_t);
template int launch_detector<double>(double*, double*, const uint8_t*, int, int, int, const DetectorCfg&, const PoissonAlias&, hipStream_t);

}  // namespace ao
