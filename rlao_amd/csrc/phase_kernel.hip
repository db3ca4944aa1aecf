// Residual-phase kernel: frozen-flow sub-pixel translation of every layer + footprint crop + layer sum
// + deformable-mirror surface + pupil + phase scaling + partial sums of the scalar telemetry.
//
// Reference stages fused here:
//   OOPAO/Atmosphere.py:406-407   layer.phase = warp(mapShift, translate(buff), order=3)[1:-1,1:-1]
//   OOPAO/Atmosphere.py:439-450   phase_support += phase[footprint] * sqrt(fractionalR0)
//   OOPAO/Atmosphere.py:474-477   OPD_no_pupil = phase_support * lambda_500 / 2 pi ; OPD = . * pupil
//   OOPAO/DeformableMirror.py:556 dm.OPD = modes @ coefs            (separable: Gy . C . Gx^T)
//   OOPAO/DeformableMirror.py:469 + Telescope.py:540-542   OPD = (OPD_atm + dm.OPD) * pupil
//   OOPAO/Telescope.py:404-412    src.phase = OPD * 2 pi / lambda_src
//   MAIN/OOPAOEnv/OOPAOEnv.py:497,522,554-555  sums for total[i], residual[i], strehl (finished in the step epilogue)
//
// Launch shape: grid = (x tiles, y tiles, n_env), workgroup = 64 x 4 lanes; a tile is TX <= 128 columns by
// TY = 16 rows, i.e. 8 pixels per lane and thousands of workgroups for a few hundred envs.
//   * the layer tile (TY+3 rows x TX+3 columns: the 4 x 4 stencil apron) is staged in LDS with row-contiguous
//     reads; the translation is the same for every pixel, so the Catmull-Rom interpolation is separable:
//     one horizontal 4-tap pass into LDS, one vertical 4-tap pass into registers (8 taps instead of 16)
//   * DM surface of the tile: s1 = Gy[tile rows] . C first (TY x nAct), then s1 . Gx^T per pixel: no work is
//     repeated between tiles, and Gx^T of the tile's columns sits in LDS
//   * Sum x, Sum x^2 of the atmosphere and residual OPD over the pupil are accumulated in float64 and
//     written per tile; the epilogue kernel adds the tiles in a fixed order (bitwise reproducible).
#include "common.hpp"

namespace ao {

constexpr int kTY = 16, kTXmax = 128;

__device__ inline double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

template <typename T>
__global__ void __launch_bounds__(256) k_phase(const KArgs<T> a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int R = a.R, nA = a.n_act, TX = a.tx;
    const int MW = TX + 4;                                   // row stride of the staged layer tile (TX + 3 used)
    T* cimg = reinterpret_cast<T*>(lds_raw);                 // [nA][nA]
    T* s1 = cimg + nA * nA;                                  // [kTY][nA]
    T* gxt = s1 + kTY * nA;                                  // [nA][TX]
    T* mapt = gxt + nA * TX;                                 // [kTY + 3][MW]
    T* ht = mapt + (kTY + 3) * MW;                           // [kTY + 3][TX]
    __shared__ double red[4][4];

    const int e = blockIdx.z;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * kTY;
    const int txe = min(TX, R - x0), tye = min(kTY, R - y0);
    const int lx = threadIdx.x, ly = threadIdx.y, tid = ly * 64 + lx;
    const size_t pix0 = (size_t)e * R * R;
    const bool separable = (a.pb.dm_opd == nullptr);

    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && a.pa.store_phase) a.pb.wfs_max[e] = (T)0;

    if (separable) {
        if (a.pb.coefs_img) {
            const T* ci = a.pb.coefs_img + (size_t)e * nA * nA;
            for (int i = tid; i < nA * nA; i += 256) cimg[i] = ci[i];
        } else {
            for (int i = tid; i < nA * nA; i += 256) cimg[i] = (T)0;
            __syncthreads();
            const T* cf = a.pb.coefs + (size_t)e * a.n_valid_act;
            for (int k = tid; k < a.n_valid_act; k += 256) cimg[a.pb.act_idx[k]] = cf[k];
        }
        // Gx^T of this tile's columns (global reads run along ix then x: contiguous)
        for (int i = tid; i < txe * nA; i += 256) {
            const int x = i / nA, ix = i - x * nA;
            gxt[ix * TX + x] = a.pb.gx[(size_t)(x0 + x) * nA + ix];
        }
        __syncthreads();
        // s1[y][ix] = sum_iy gy[y0 + y][iy] * C[iy][ix]
        for (int i = tid; i < tye * nA; i += 256) {
            const int y = i / nA, ix = i - y * nA;
            const T* g = a.pb.gy + (size_t)(y0 + y) * nA;
            T acc = (T)0;
            for (int iy = 0; iy < nA; ++iy) acc += g[iy] * cimg[iy * nA + ix];
            s1[y * nA + ix] = acc;
        }
    }

    // this lane's pixels: x = lx + 64 i (i < 2), y = ly + 4 j (j < 4)
    T sup[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sup[i][j] = (T)0;

    if (a.pa.update_atm) {
        for (int l = 0; l < a.pa.n_layer; ++l) {
            const LayerTaps& tp = layer_taps(a.pa, l, e);
            const int S = a.pa.S_l[l], foot = a.pa.foot_l[l];       // the layer's own grid (fov != 0: it grows with the altitude)
            const T* map = static_cast<const T*>(a.pa.screen[l]) + (size_t)e * S * S;
            const int r0 = y0 + foot + tp.dy - 1, c0 = x0 + foot + tp.dx - 1;
            __syncthreads();                                  // previous layer's tiles are no longer read
            for (int r = ly; r < tye + 3; r += 4) {
                const int rr = r0 + r;
                for (int c = lx; c < txe + 3; c += 64) {
                    const int cc = c0 + c;
                    int pr = rr + tp.oy, pc = cc + tp.ox;           // torus: physical = (logical + origin) mod S
                    pr = pr >= S ? pr - S : pr;
                    pc = pc >= S ? pc - S : pc;
                    mapt[r * MW + c] = (rr >= 0 && rr < S && cc >= 0 && cc < S) ? map[(size_t)pr * S + pc] : (T)0;
                }
            }
            __syncthreads();
            const T wx0 = (T)tp.wx[0], wx1 = (T)tp.wx[1], wx2 = (T)tp.wx[2], wx3 = (T)tp.wx[3];
            for (int r = ly; r < tye + 3; r += 4) {
                const T* m = mapt + r * MW;
                for (int x = lx; x < txe; x += 64)
                    ht[r * TX + x] = ((wx0 * m[x] + wx1 * m[x + 1]) + wx2 * m[x + 2]) + wx3 * m[x + 3];
            }
            __syncthreads();
            const T wy0 = (T)tp.wy[0], wy1 = (T)tp.wy[1], wy2 = (T)tp.wy[2], wy3 = (T)tp.wy[3];
            const T* mm = static_cast<const T*>(a.pa.minmax[l]) + 2 * e;
            const T lo = mm[0], hi = mm[1], wl = (T)tp.weight;
            const bool zero_outside = (lo > (T)0 || hi < (T)0);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int x = lx + 64 * i;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int y = ly + 4 * j;
                    if (x < txe && y < tye) {
                        const T* h = ht + y * TX + x;
                        T v = ((wy0 * h[0] + wy1 * h[TX]) + wy2 * h[2 * TX]) + wy3 * h[3 * TX];
                        // skimage clip=True: clamp to the input range, keep exact zeros when 0 is outside it
                        if (!(zero_outside && v == (T)0)) v = v < lo ? lo : (v > hi ? hi : v);
                        sup[i][j] += v * wl;
                    }
                }
            }
        }
    }
    __syncthreads();                                          // s1 / gxt complete (and tiles done)

    double s_atm = 0.0, q_atm = 0.0, s_res = 0.0, q_res = 0.0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int x = lx + 64 * i;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int y = ly + 4 * j;
            if (x < txe && y < tye) {
                const size_t q = (size_t)(y0 + y) * R + (x0 + x);
                T atm;
                if (a.pa.update_atm) {
                    atm = sup[i][j] * a.atm_scale;
                    if (a.pa.store_atm) a.pb.opd_atm[pix0 + q] = atm;
                } else {
                    atm = a.pb.opd_atm[pix0 + q];
                }
                T dm;
                if (separable) {
                    dm = (T)0;
                    const T* sr = s1 + y * nA;
                    for (int ix = 0; ix < nA; ++ix) dm += sr[ix] * gxt[ix * TX + x];
                } else {
                    dm = a.pb.dm_opd[pix0 + q];
                }
                const bool in = a.pb.pupil[q] != 0;
                const T res = in ? (atm + dm) : (T)0;
                if (a.pa.store_phase && (!(a.ablate & 8) || res == 12345.f)) a.pb.phase[pix0 + q] = res * a.src_scale;
                if (in) {
                    const double da = (double)atm, dr = (double)res;
                    s_atm += da;
                    q_atm += da * da;
                    s_res += dr;
                    q_res += dr * dr;
                }
            }
        }
    }
    s_atm = wave_sum(s_atm);
    q_atm = wave_sum(q_atm);
    s_res = wave_sum(s_res);
    q_res = wave_sum(q_res);
    if (lx == 0) {
        red[0][ly] = s_atm;
        red[1][ly] = q_atm;
        red[2][ly] = s_res;
        red[3][ly] = q_res;
    }
    __syncthreads();
    if (tid < 4 && a.pa.store_phase) {
        const int tile = blockIdx.y * gridDim.x + blockIdx.x, n_tiles = gridDim.x * gridDim.y;
        a.pb.part[((size_t)e * n_tiles + tile) * 4 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    }
}

// ---------------------------------------------------------------------------------------------------
// float32 production kernel: same tile decomposition, the DM surface of the tile on the matrix cores.
//   dm_tile[16][TX] = s1[16][nAct] . Gx^T[nAct][TX]   with v_mfma_f32_16x16x4_f32 (exact f32 fma chain)
// Wave w owns the 16-column sub-tiles t = 2w, 2w+1; the MFMA result layout (column = lane & 15, rows
// 4 (lane >> 4) + r) IS the lane -> pixel map of the whole kernel, so the interpolated atmosphere, the DM
// surface and the pupil/telemetry epilogue of a pixel all live in the same lane and nothing is shuffled.
// Per MFMA (1024 MACs) a lane reads one s1 and one Gx^T value from LDS -- 32x less LDS traffic than the
// per-pixel dot products of the generic kernel, which were the bottleneck there.
// ---------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

// s1 tile rows on the matrix cores (see k_phase_mfma).  KS = k steps held in registers (n_act <= 4 KS).  Every load is
// unconditional from a clamped index: a predicated load compiles to a branch, and a run of them to a chain of latencies.
template <int KS>
__device__ inline void s1_tiles_mfma(const float* __restrict__ gya, int ga_stride, const float* cimg, float* s1, int y0, int R,
                                     int nA, int nAp, int SS, int lc, int lq, int wave) {
    float av[KS];
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(gya) + (size_t)(y0 >> 4) * (ga_stride >> 2) * 64 + (lq * 16 + lc);
#pragma unroll
        for (int q4 = 0; q4 < KS / 4; ++q4) {
            const f32x4 t = src[4 * q4 < ga_stride ? 64 * q4 : 0];
#pragma unroll
            for (int d = 0; d < 4; ++d) av[4 * q4 + d] = 4 * q4 < ga_stride ? t[d] : 0.f;     // A[i = lane & 15 -> row][k = lane >> 4]
        }
    }
    for (int ct = wave; 16 * ct < nA; ct += 4) {
        const int ix = 16 * ct + lc;
        float bv[KS];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int iy = lq + 4 * ks;
            const bool ok = iy < nA && ix < nA;
            const float t = cimg[ok ? iy * nA + ix : 0];
            bv[ks] = ok ? t : 0.f;                           // B[k = lane >> 4][j = lane & 15]
        }
        f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
            if (4 * ks < nAp) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], d, 0, 0, 0);
        if (ix < nA) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s1[(4 * lq + r) * SS + ix] = d[r];
        }
    }
}

__global__ void __launch_bounds__(256) k_phase_mfma(const KArgs<float> a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int R = a.R, nA = a.n_act;
    constexpr int TX = kTXmax, MW = TX + 4;
    const int nAp = (nA + 3) & ~3, SS = nAp + 1;             // K padded to 4, s1 row stride odd (bank spread)
    const bool rows_given = a.pb.s1a != nullptr;             // Gy C already in HBM (k_dm_rows): no command image, no s1 here
    float* cimg = reinterpret_cast<float*>(lds_raw);         // [nA][nA]
    float* s1 = cimg + nA * nA;                              // [16][SS]
    float* mapt = rows_given ? cimg : s1 + kTY * SS;         // [kTY + 3][MW]
    __shared__ double red[4][4];

    const int e = blockIdx.z;
    const int x0 = blockIdx.x * TX, y0 = blockIdx.y * kTY;
    const int txe = min(TX, R - x0), tye = min(kTY, R - y0);
    const int tid = threadIdx.x, lx = tid & 63, ly = tid >> 6;      // ly = wave
    const int lc = lx & 15, lq = lx >> 4;                            // MFMA lane decomposition
    const size_t pix0 = (size_t)e * R * R;

    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && a.pa.store_phase) a.pb.wfs_max[e] = 0.f;

    if (rows_given) {
    } else if (a.pb.coefs_img) {
        for (int i = tid; i < kTY * SS; i += 256) s1[i] = 0.f;
        // batches of 8 independent loads per lane (a load-then-store loop pays one memory latency per iteration)
        const float* ci = a.pb.coefs_img + (size_t)e * nA * nA;
        for (int i0 = tid; i0 < nA * nA; i0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ci[i0 + 256 * q < nA * nA ? i0 + 256 * q : i0];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (i0 + 256 * q < nA * nA) cimg[i0 + 256 * q] = v[q];
        }
    } else {
        for (int i = tid; i < kTY * SS; i += 256) s1[i] = 0.f;
        for (int i = tid; i < nA * nA; i += 256) cimg[i] = 0.f;
        __syncthreads();
        const float* cf = a.pb.coefs + (size_t)e * a.n_valid_act;
        for (int k = tid; k < a.n_valid_act; k += 256) cimg[a.pb.act_idx[k]] = cf[k];
    }
    __syncthreads();
    // s1[y][ix] = sum_iy gy[y0 + y][iy] C[iy][ix] on the matrix cores: 16 x 16 output tiles over the command columns,
    // wave w takes the tiles w, w + 4, ...; the A operands (the tile's 16 rows of gy) are the same for every column tile
    // and are loaded once, all loads in flight together.  (As a per-output dot product through LDS this was 2/3 of the
    // kernel at 81 actuators across: a chain of nA LDS latencies per output.)
    if (!rows_given && !(a.ablate & 1)) {
        if (nAp <= 32) s1_tiles_mfma<8>(a.pb.gya, a.pb.ga_stride, cimg, s1, y0, R, nA, nAp, SS, lc, lq, ly);
        else s1_tiles_mfma<32>(a.pb.gya, a.pb.ga_stride, cimg, s1, y0, R, nA, nAp, SS, lc, lq, ly);
    }

    // lane's pixels: sub-tile tt (x = 16 (2 ly + tt) + lc), rows y = 4 lq + r
    float sup[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sup[tt][r] = 0.f;

    if (a.pa.update_atm && !(a.ablate & 2)) {
        for (int l = 0; l < a.pa.n_layer; ++l) {
            const LayerTaps& tp = layer_taps(a.pa, l, e);
            const int S = a.pa.S_l[l], foot = a.pa.foot_l[l];       // the layer's own grid (fov != 0: it grows with the altitude)
            const float* map = static_cast<const float*>(a.pa.screen[l]) + (size_t)e * S * S;
            const int r0 = y0 + foot + tp.dy - 1, c0 = x0 + foot + tp.dx - 1;
            __syncthreads();                                  // the previous layer's tile is no longer read
            {
                constexpr int NV = ((kTY + 3) * MW + 255) / 256;      // 10 independent loads in flight per lane
                float v[NV];
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int idx = tid + 256 * k;
                    const int r = idx / MW, c = idx - r * MW;
                    const int rr = r0 + r, cc = c0 + c;
                    int pr = rr + tp.oy, pc = cc + tp.ox;           // torus: physical = (logical + origin) mod S
                    pr = pr >= S ? pr - S : pr;
                    pc = pc >= S ? pc - S : pc;
                    const bool okl = idx < (kTY + 3) * MW && r < tye + 3 && c < txe + 3 && rr >= 0 && rr < S && cc >= 0 && cc < S;
                    const float t = map[okl ? (size_t)pr * S + pc : 0];       // unconditional load from a clamped index
                    v[k] = okl ? t : 0.f;
                }
#pragma unroll
                for (int k = 0; k < NV; ++k) {
                    const int idx = tid + 256 * k;
                    if (idx < (kTY + 3) * MW) mapt[idx] = v[k];
                }
            }
            __syncthreads();
            const float wx0 = (float)tp.wx[0], wx1 = (float)tp.wx[1], wx2 = (float)tp.wx[2], wx3 = (float)tp.wx[3];
            const float wy0 = (float)tp.wy[0], wy1 = (float)tp.wy[1], wy2 = (float)tp.wy[2], wy3 = (float)tp.wy[3];
            const float* mm = static_cast<const float*>(a.pa.minmax[l]) + 2 * e;
            const float lo = mm[0], hi = mm[1], wl = (float)tp.weight;
            const bool zero_outside = (lo > 0.f || hi < 0.f);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int x = 16 * (2 * ly + tt) + lc;
                // horizontal 4-tap pass of the 7 tile rows that feed this lane's 4 output rows, then vertical
                float h[7];
#pragma unroll
                for (int k = 0; k < 7; ++k) {
                    const float* m = mapt + (4 * lq + k) * MW + x;
                    h[k] = ((wx0 * m[0] + wx1 * m[1]) + wx2 * m[2]) + wx3 * m[3];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = ((wy0 * h[r] + wy1 * h[r + 1]) + wy2 * h[r + 2]) + wy3 * h[r + 3];
                    if (!(zero_outside && v == 0.f)) v = v < lo ? lo : (v > hi ? hi : v);
                    sup[tt][r] += v * wl;
                }
            }
        }
    }
    __syncthreads();                                              // s1 complete

    // the pupil flags of the lane's 8 pixels, all loads issued together (a conditional load per pixel is a latency each)
    bool pup[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int xl = 16 * (2 * ly + tt) + lc, y = 4 * lq + r;
            const bool okp = xl < txe && y < tye;
            const uint8_t t = (a.ablate & 64) ? 1 : a.pb.pupil[okp ? (size_t)(y0 + y) * R + (x0 + xl) : 0];
            pup[tt][r] = okp && t != 0;
        }
    double s_atm = 0.0, q_atm = 0.0, s_res = 0.0, q_res = 0.0;
    // A[i = lane & 15 -> tile row][k = lane >> 4 + 4 s] = (Gy C)[y0 + i][k] from the operand-layout rows k_dm_rows wrote:
    // ga_stride / 4 16-byte loads per lane (<= 8: n_act <= 128), all in flight together
    constexpr int NQ = 8;
    f32x4 aq[NQ];
    const int nq = a.pb.ga_stride / 4;
    if (rows_given && !(a.ablate & 4)) {
        const int Rp = (R + 127) & ~127;
        const f32x4* asrc = reinterpret_cast<const f32x4*>(a.pb.s1a) + ((size_t)e * (Rp >> 4) + (y0 >> 4)) * nq * 64 + lx;
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) aq[q4] = asrc[q4 < nq ? 64 * q4 : 0];
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int xl = 16 * (2 * ly + tt) + lc;
        f32x4 dmv = {0.f, 0.f, 0.f, 0.f};
        const float* ap = s1 + lc * SS + lq;                      // A[i = lane & 15][k = lane >> 4]            (LDS)
        // B[k = lane >> 4 + 4 s][j = lane & 15 -> column] = gx[x][k]: the operand table gives a lane its k steps as 16-byte loads
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(a.pb.gxa) + (size_t)((x0 >> 4) + 2 * ly + tt) * nq * 64 + lx;
        if (rows_given) {
            if (!(a.ablate & 4)) {
            f32x4 bq[NQ];
#pragma unroll
            for (int q4 = 0; q4 < NQ; ++q4) bq[q4] = bsrc[q4 < nq ? 64 * q4 : 0];
#pragma unroll
            for (int q4 = 0; q4 < NQ; ++q4)
                if (16 * q4 < nAp) {                              // uniform
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (16 * q4 + 4 * d < nAp) dmv = __builtin_amdgcn_mfma_f32_16x16x4f32(aq[q4][d], bq[q4][d], dmv, 0, 0, 0);
                }
            }
        } else if (!(a.ablate & 4))
        for (int kb = 0; kb < nAp; kb += 32) {                    // 8 k steps = 2 loads per batch
            const f32x4 b0 = bsrc[64 * (kb / 16)], b1 = bsrc[64 * (kb / 16 + 1 < nq ? kb / 16 + 1 : kb / 16)];
            const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (kb + 4 * j < nAp) dmv = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[kb + 4 * j], bv[j], dmv, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int y = 4 * lq + r;
            if (xl < txe && y < tye) {
                const size_t q = (size_t)(y0 + y) * R + (x0 + xl);
                float atm;
                if (a.pa.update_atm) {
                    atm = sup[tt][r] * a.atm_scale;
                    if (a.pa.store_atm) a.pb.opd_atm[pix0 + q] = atm;
                } else {
                    atm = a.pb.opd_atm[pix0 + q];
                }
                const bool in = (a.ablate & 16) ? true : pup[tt][r];
                const float res = in ? (atm + dmv[r]) : 0.f;
                if (a.pa.store_phase && (!(a.ablate & 8) || res == 12345.f)) a.pb.phase[pix0 + q] = res * a.src_scale;
                if (in && !(a.ablate & 32)) {
                    const double da = (double)atm, dr = (double)res;
                    s_atm += da;
                    q_atm += da * da;
                    s_res += dr;
                    q_res += dr * dr;
                }
            }
        }
    }
    s_atm = wave_sum(s_atm);
    q_atm = wave_sum(q_atm);
    s_res = wave_sum(s_res);
    q_res = wave_sum(q_res);
    if (lx == 0) {
        red[0][ly] = s_atm;
        red[1][ly] = q_atm;
        red[2][ly] = s_res;
        red[3][ly] = q_res;
    }
    __syncthreads();
    if (tid < 4 && a.pa.store_phase) {
        const int tile = blockIdx.y * gridDim.x + blockIdx.x, n_tiles = gridDim.x * gridDim.y;
        a.pb.part[((size_t)e * n_tiles + tile) * 4 + tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
    }
}

// The same kernel with every global access a 16-byte one (R a multiple of 4: every BASELINE geometry).  The operands of the DM
// product are swapped -- gx as A, the Gy C row as B -- so that the matrix cores' output layout gives a lane FOUR CONSECUTIVE pixels
// of ONE row (the lane -> pixel map of the fused step kernel's stage A): the layer tile is staged with 3 float4 loads per lane
// instead of 10 dword loads, its taps are read as float4 from LDS, the pupil flags of 4 pixels are one load, the phase (and OPD)
// rows are written as float4 -- the vector-memory path takes ~16 cycles per wave-instruction whatever its width.  Same arithmetic
// per pixel (the same taps in the same order, the same k order of the product): results are bit-identical to k_phase_mfma.
// BAND (one layer): the layer tile of the WHOLE band -- 19 rows x (R + 4) columns, 37 KB at R = 480 -- is staged once, all its loads
// in flight together, instead of a 19 x 132 tile per 128-column chunk: one memory latency and one pair of barriers per workgroup
// instead of four (ELT size: a wave lived 43 us, 56 % of it waiting).
// NQ: 16-byte loads per lane of a DM operand (Gy C rows, gx columns): ga_stride / 4 rounded up to even, <= 8 (n_act <= 128)
template <bool BAND, int NQ>
__global__ void __launch_bounds__(256, 4) k_phase_mfma4(const KArgs<float> a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int R = a.R, nA = a.n_act;
    constexpr int TX = kTXmax, MW = TX + 4;
    const int mws = BAND ? R + 4 : MW;                       // row stride of the staged tile (R % 4 == 0)
    const int nAp = (nA + 3) & ~3, SS = nAp + 1;             // K padded to 4, s1 row stride odd (bank spread)
    const bool rows_given = a.pb.s1a != nullptr;             // Gy C already in HBM (k_dm_rows): no command image, no s1 here
    float* cimg = reinterpret_cast<float*>(lds_raw);         // [nA][nA]
    float* s1 = cimg + nA * nA;                              // [16][SS]
    float* mapt = rows_given ? cimg : s1 + kTY * SS;         // [kTY + 3][MW]
    __shared__ double red[4][4];

    const int e = blockIdx.z;
    const int y0 = blockIdx.y * kTY, tye = min(kTY, R - y0);      // a workgroup takes the whole 16-row band, 128 columns at a time
    const int tid = threadIdx.x, lx = tid & 63, ly = tid >> 6;      // ly = wave
    const int lc = lx & 15, lq = lx >> 4;                            // MFMA lane decomposition
    const size_t pix0 = (size_t)e * R * R;

    if (blockIdx.y == 0 && tid == 0 && a.pa.store_phase) a.pb.wfs_max[e] = 0.f;

    if (rows_given) {
    } else if (a.pb.coefs_img) {
        for (int i = tid; i < kTY * SS; i += 256) s1[i] = 0.f;
        // batches of 8 independent loads per lane (a load-then-store loop pays one memory latency per iteration)
        const float* ci = a.pb.coefs_img + (size_t)e * nA * nA;
        for (int i0 = tid; i0 < nA * nA; i0 += 8 * 256) {
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = ci[i0 + 256 * q < nA * nA ? i0 + 256 * q : i0];
#pragma unroll
            for (int q = 0; q < 8; ++q) if (i0 + 256 * q < nA * nA) cimg[i0 + 256 * q] = v[q];
        }
    } else {
        for (int i = tid; i < kTY * SS; i += 256) s1[i] = 0.f;
        for (int i = tid; i < nA * nA; i += 256) cimg[i] = 0.f;
        __syncthreads();
        const float* cf = a.pb.coefs + (size_t)e * a.n_valid_act;
        for (int k = tid; k < a.n_valid_act; k += 256) cimg[a.pb.act_idx[k]] = cf[k];
    }
    __syncthreads();
    // s1[y][ix] = sum_iy gy[y0 + y][iy] C[iy][ix] on the matrix cores: 16 x 16 output tiles over the command columns,
    // wave w takes the tiles w, w + 4, ...; the A operands (the tile's 16 rows of gy) are the same for every column tile
    // and are loaded once, all loads in flight together.  (As a per-output dot product through LDS this was 2/3 of the
    // kernel at 81 actuators across: a chain of nA LDS latencies per output.)
    if (!rows_given && !(a.ablate & 1)) {
        if (nAp <= 32) s1_tiles_mfma<8>(a.pb.gya, a.pb.ga_stride, cimg, s1, y0, R, nA, nAp, SS, lc, lq, ly);
        else s1_tiles_mfma<32>(a.pb.gya, a.pb.ga_stride, cimg, s1, y0, R, nA, nAp, SS, lc, lq, ly);
    }

    double s_atm = 0.0, q_atm = 0.0, s_res = 0.0, q_res = 0.0;
    // (Gy C)[y0 + i][k] of the band from the operand-layout rows k_dm_rows wrote, ONCE for all its column chunks (as separate
    // 16 x 128 tile workgroups every chunk re-read them: a quarter of the kernel's HBM traffic at 81 actuators across):
    // ga_stride / 4 16-byte loads per lane (<= 8: n_act <= 128), all in flight together
    f32x4 aq[NQ];
    const int nq = a.pb.ga_stride / 4;
    if (rows_given && !(a.ablate & 4)) {
        const int Rp = (R + 127) & ~127;
        const f32x4* asrc = reinterpret_cast<const f32x4*>(a.pb.s1a) + ((size_t)e * (Rp >> 4) + (y0 >> 4)) * nq * 64 + lx;
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) aq[q4] = asrc[q4 < nq ? 64 * q4 : 0];
    }
    // float4 number idx of a staged tile of `mw4` float4 per row whose element (r, c) is map[r0 + r][c0 + c] (rows of the map are only
    // 4-byte aligned); rows < nr and columns < nc are needed, the rest stays zero
    auto tile4 = [&](const float* map, const LayerTaps& tp, int S, int idx, int mw4, int nr, int nc, int r0, int c0) {
        const int r = idx / mw4, c = 4 * (idx - r * mw4);
        const int rr = r0 + r, cc = c0 + c;
        const bool need = r < nr && c < nc && rr >= 0 && rr < S && cc >= 0 && cc < S;
        int pr = rr + tp.oy, pc = cc + tp.ox;                   // torus: physical = (logical + origin) mod S
        pr = pr >= S ? pr - S : pr;
        pc = pc >= S ? pc - S : pc;
        const bool ok = need && cc + 3 < S && pc + 3 < S;        // the 4 columns are contiguous in memory
        f32x4 t;
        __builtin_memcpy(&t, map + (ok ? (size_t)pr * S + pc : 0), 16);
        f32x4 v = ok ? t : f32x4{0.f, 0.f, 0.f, 0.f};
        if (need && !ok) {
            // a float4 that straddles the wrap of the torus or the edge of the screen: element by element
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                int pd = pc + d;
                pd = pd >= S ? pd - S : pd;
                if (cc + d < S) v[d] = map[(size_t)pr * S + pd];
            }
        }
        return v;
    };
    if (BAND && a.pa.update_atm && !(a.ablate & 2)) {
        const LayerTaps& tp = layer_taps(a.pa, 0, e);
        const int S = a.pa.S_l[0], foot = a.pa.foot_l[0];
        const float* map = static_cast<const float*>(a.pa.screen[0]) + (size_t)e * S * S;
        const int r0 = y0 + foot + tp.dy - 1, c0 = foot + tp.dx - 1;
        const int mw4 = mws / 4, total = (kTY + 3) * mw4;
        for (int i0 = tid; i0 < total; i0 += 5 * 256) {           // 5 independent 16-byte loads in flight per lane, twice at R = 480
            f32x4 v[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = tile4(map, tp, S, i0 + 256 * k < total ? i0 + 256 * k : i0, mw4, tye + 3, R + 3, r0, c0);
#pragma unroll
            for (int k = 0; k < 5; ++k)
                if (i0 + 256 * k < total) *reinterpret_cast<f32x4*>(mapt + 4 * (i0 + 256 * k)) = v[k];
        }
        __syncthreads();
    }
    for (int xb = 0; xb < (R + TX - 1) / TX; ++xb) {
    const int x0 = xb * TX, txe = min(TX, R - x0);
    // lane's pixels: sub-tile tt (x = 16 (2 ly + tt) + lc), rows y = 4 lq + r
    float sup[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 4; ++r) sup[tt][r] = 0.f;

    if (a.pa.update_atm && !(a.ablate & 2)) {
        for (int l = 0; l < (BAND ? 1 : a.pa.n_layer); ++l) {
            const LayerTaps& tp = layer_taps(a.pa, l, e);
            if (!BAND) {
                const int S = a.pa.S_l[l], foot = a.pa.foot_l[l];       // the layer's own grid (fov != 0: it grows with the altitude)
                const float* map = static_cast<const float*>(a.pa.screen[l]) + (size_t)e * S * S;
                const int r0 = y0 + foot + tp.dy - 1, c0 = x0 + foot + tp.dx - 1;
                __syncthreads();                                  // the previous layer's tile is no longer read
                constexpr int MW4 = MW / 4, NV4 = ((kTY + 3) * MW4 + 255) / 256;     // 3 independent 16-byte loads in flight per lane
                f32x4 v[NV4];
#pragma unroll
                for (int k = 0; k < NV4; ++k) {
                    const int idx = tid + 256 * k;
                    v[k] = tile4(map, tp, S, idx < (kTY + 3) * MW4 ? idx : 0, MW4, idx < (kTY + 3) * MW4 ? tye + 3 : 0, txe + 3, r0, c0);
                }
#pragma unroll
                for (int k = 0; k < NV4; ++k) {
                    const int idx = tid + 256 * k;
                    if (idx < (kTY + 3) * MW4) *reinterpret_cast<f32x4*>(mapt + 4 * idx) = v[k];
                }
                __syncthreads();
            }
            const float wx0 = (float)tp.wx[0], wx1 = (float)tp.wx[1], wx2 = (float)tp.wx[2], wx3 = (float)tp.wx[3];
            const float wy0 = (float)tp.wy[0], wy1 = (float)tp.wy[1], wy2 = (float)tp.wy[2], wy3 = (float)tp.wy[3];
            const float* mm = static_cast<const float*>(a.pa.minmax[l]) + 2 * e;
            const float lo = mm[0], hi = mm[1], wl = (float)tp.weight;
            const bool zero_outside = (lo > 0.f || hi < 0.f);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int xt = 16 * (2 * ly + tt) + 4 * lq;     // tile column of the first tap of the lane's first pixel
                // the 4 rows x 8 columns of taps (7 used) as 16-byte LDS reads; horizontal pass per row, then vertical
                float h[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float* m = mapt + (lc + q) * mws + xt + (BAND ? x0 : 0);
                    const f32x4 m0 = *reinterpret_cast<const f32x4*>(m), m1 = *reinterpret_cast<const f32x4*>(m + 4);
                    const float t[7] = {m0[0], m0[1], m0[2], m0[3], m1[0], m1[1], m1[2]};
#pragma unroll
                    for (int r = 0; r < 4; ++r) h[q][r] = ((wx0 * t[r] + wx1 * t[r + 1]) + wx2 * t[r + 2]) + wx3 * t[r + 3];
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = ((wy0 * h[0][r] + wy1 * h[1][r]) + wy2 * h[2][r]) + wy3 * h[3][r];
                    if (!(zero_outside && v == 0.f)) v = v < lo ? lo : (v > hi ? hi : v);
                    sup[tt][r] += v * wl;
                }
            }
        }
    }
    if (xb == 0) __syncthreads();                                 // s1 complete

    // the pupil flags of the lane's 2 x 4 pixels: one 4-byte load each (x and R are multiples of 4)
    bool pup[2][4];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int xl = 16 * (2 * ly + tt) + 4 * lq, y = lc;
        const bool okp = xl < txe && y < tye;
        uint32_t t4 = 0x01010101u;
        if (!(a.ablate & 64)) __builtin_memcpy(&t4, a.pb.pupil + (okp ? (size_t)(y0 + y) * R + (x0 + xl) : 0), 4);
#pragma unroll
        for (int r = 0; r < 4; ++r) pup[tt][r] = okp && ((t4 >> (8 * r)) & 0xffu) != 0;
    }
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int xl = 16 * (2 * ly + tt) + 4 * lq;               // the lane's 4 consecutive columns; its row is lc
        f32x4 dmv = {0.f, 0.f, 0.f, 0.f};
        const float* ap = s1 + lc * SS + lq;                      // B[k = lane >> 4][j = lane & 15 -> row]      (LDS)
        // D[x][y] = sum_k gx[x][k] (Gy C)[y][k]: gx is the A operand (i = lane & 15 -> column of the 16-column sub-tile) and the
        // Gy C row the B operand, so that the matrix cores' output layout (rows 4 (lane >> 4) + r, column lane & 15) hands a
        // lane FOUR CONSECUTIVE pixels of ONE row: every global access below is a 16-byte one
        const f32x4* bsrc = reinterpret_cast<const f32x4*>(a.pb.gxa) + (size_t)((x0 >> 4) + 2 * ly + tt) * nq * 64 + lx;
        if (rows_given) {
            if (!(a.ablate & 4)) {
            f32x4 bq[NQ];
#pragma unroll
            for (int q4 = 0; q4 < NQ; ++q4) bq[q4] = bsrc[q4 < nq ? 64 * q4 : 0];
#pragma unroll
            for (int q4 = 0; q4 < NQ; ++q4)
                if (16 * q4 < nAp) {                              // uniform
#pragma unroll
                    for (int d = 0; d < 4; ++d)
                        if (16 * q4 + 4 * d < nAp) dmv = __builtin_amdgcn_mfma_f32_16x16x4f32(bq[q4][d], aq[q4][d], dmv, 0, 0, 0);
                }
            }
        } else if (!(a.ablate & 4))
        for (int kb = 0; kb < nAp; kb += 32) {                    // 8 k steps = 2 loads per batch
            const f32x4 b0 = bsrc[64 * (kb / 16)], b1 = bsrc[64 * (kb / 16 + 1 < nq ? kb / 16 + 1 : kb / 16)];
            const float bv[8] = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (kb + 4 * j < nAp) dmv = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], ap[kb + 4 * j], dmv, 0, 0, 0);
        }
        if (xl < txe && lc < tye) {
            const size_t q = (size_t)(y0 + lc) * R + (x0 + xl);
            f32x4 atm, phi;
            if (!a.pa.update_atm) atm = *reinterpret_cast<const f32x4*>(a.pb.opd_atm + pix0 + q);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (a.pa.update_atm) atm[r] = sup[tt][r] * a.atm_scale;
                const bool in = (a.ablate & 16) ? true : pup[tt][r];
                const float res = in ? (atm[r] + dmv[r]) : 0.f;
                phi[r] = res * a.src_scale;
                if (in && !(a.ablate & 32)) {
                    const double da = (double)atm[r], dr = (double)res;
                    s_atm += da;
                    q_atm += da * da;
                    s_res += dr;
                    q_res += dr * dr;
                }
            }
            if (a.pa.update_atm && a.pa.store_atm) *reinterpret_cast<f32x4*>(a.pb.opd_atm + pix0 + q) = atm;
            if (a.pa.store_phase && !(a.ablate & 8)) *reinterpret_cast<f32x4*>(a.pb.phase + pix0 + q) = phi;
        }
    }
    }
    s_atm = wave_sum(s_atm);
    q_atm = wave_sum(q_atm);
    s_res = wave_sum(s_res);
    q_res = wave_sum(q_res);
    if (lx == 0) {
        red[0][ly] = s_atm;
        red[1][ly] = q_atm;
        red[2][ly] = s_res;
        red[3][ly] = q_res;
    }
    __syncthreads();
    if (tid < 4 && a.pa.store_phase) {
        // (the telemetry buffer has one slot per 16 x 128 tile, phase_tiles(): the band's sums go to its first slot)
        const int gx = (R + TX - 1) / TX, n_tiles = gx * gridDim.y;
        double* pp = a.pb.part + ((size_t)e * n_tiles + (size_t)blockIdx.y * gx) * 4;
        pp[tid] = red[tid][0] + red[tid][1] + red[tid][2] + red[tid][3];
        for (int t = 1; t < gx; ++t) pp[4 * t + tid] = 0.0;
    }
}

template <typename T>
int launch_phase_mfma(const KArgs<T>&, int, hipStream_t) { return -1; }
template <>
int launch_phase_mfma<float>(const KArgs<float>& a, int n_env, hipStream_t st) {
    const int nA = a.n_act, nAp = (nA + 3) & ~3, TX = kTXmax, MW = TX + 4;
    const size_t lds = sizeof(float) * ((a.pb.s1a ? 0 : (size_t)nA * nA + (size_t)kTY * (nAp + 1)) + (size_t)(kTY + 3) * MW);
    if (a.pb.gxa == nullptr || nA > 128) return -1;
    if (lds > 160 * 1024) return -1;
    if (lds > 64 * 1024)
        AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_phase_mfma), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    dim3 grid(cdiv(a.R, TX), cdiv(a.R, kTY), n_env);
    if (a.R % 4 == 0 && !(a.ablate & 256)) {
        // one layer: its tile of the whole 16-row band in LDS (19 x (R + 4) floats) instead of one 19 x 132 tile per chunk
        const size_t lds_band = sizeof(float) * ((a.pb.s1a ? 0 : (size_t)nA * nA + (size_t)kTY * (nAp + 1)) + (size_t)(kTY + 3) * (a.R + 4));
        const bool band = a.pa.n_layer == 1 && a.pa.update_atm && lds_band <= 160 * 1024 && !(a.ablate & 512);
        const size_t l4 = band ? lds_band : lds;
        const int nq = a.pb.ga_stride / 4;
        void (*kern)(const KArgs<float>);
        if (nq <= 2) kern = band ? k_phase_mfma4<true, 2> : k_phase_mfma4<false, 2>;
        else if (nq <= 4) kern = band ? k_phase_mfma4<true, 4> : k_phase_mfma4<false, 4>;
        else if (nq <= 6) kern = band ? k_phase_mfma4<true, 6> : k_phase_mfma4<false, 6>;
        else kern = band ? k_phase_mfma4<true, 8> : k_phase_mfma4<false, 8>;
        if (l4 > 64 * 1024)
            AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4));
        hipLaunchKernelGGL(kern, dim3(1, grid.y, grid.z), dim3(256), l4, st, a);      // one workgroup per 16-row band
    } else {
        hipLaunchKernelGGL(k_phase_mfma, grid, dim3(256), lds, st, a);
    }
    AO_HIP(hipGetLastError());
    return 0;
}

// Tile width of the generic kernel: as wide as its LDS image allows (command image + Gy C + Gx^T + two staged tiles);
// large DMs in float64 (ELT calibration shards: 81 actuators across) take narrower tiles.
static size_t phase_lds_bytes(int n_act, int tx, size_t esz) {
    const int MW = tx + 4;
    return esz * ((size_t)n_act * n_act + (size_t)kTY * n_act + (size_t)n_act * tx + (size_t)(kTY + 3) * MW + (size_t)(kTY + 3) * tx);
}
template <typename T>
__global__ void __launch_bounds__(256) k_coefs_image(const T* __restrict__ coefs, const int* __restrict__ act_idx, T* __restrict__ img,
                                                     int n_act, int n_valid_act) {
    const int e = blockIdx.y;
    T* im = img + (size_t)e * n_act * n_act;
    // grid.x workgroups share the image: zero first (its own slice), then scatter -- the scatter targets are disjoint from
    // every zero store of a valid actuator only within ONE workgroup, so one workgroup per env does both
    for (int i = threadIdx.x; i < n_act * n_act; i += blockDim.x) im[i] = (T)0;
    __syncthreads();
    const T* cf = coefs + (size_t)e * n_valid_act;
    for (int k = threadIdx.x; k < n_valid_act; k += blockDim.x) im[act_idx[k]] = cf[k];
}
// Gy C for every pixel row of every env, once per step (float32, separable DM): s1[e][y][k] = sum_iy gy[y][iy] C_e[iy][k], stored in
// the MFMA operand layout (ga_index(), common.hpp) that k_phase_mfma reads with coalesced 16-byte loads.  Each 16 x 128 tile workgroup
// of the phase kernel used to rebuild the command image (n_act^2 floats of LDS) and its 16 rows of this product: at 81
// actuators across that was a third of the kernel, 30 x 4 times per env.  Workgroup = (128-row band, env): the command image
// is scattered into LDS once, wave w takes the row tiles w and w + 4, all column tiles.
template <int KS>
__global__ void __launch_bounds__(256) k_dm_rows(const float* __restrict__ coefs, const int* __restrict__ act_idx,
                                                 const float* __restrict__ gya, float* __restrict__ s1a, int R, int nA,
                                                 int n_valid_act, int ga_stride) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    float* cimg = reinterpret_cast<float*>(lds_raw);         // [nA][nA]
    const int e = blockIdx.y, y0 = 128 * blockIdx.x, Rp = 128 * gridDim.x;
    const int tid = threadIdx.x, lx = tid & 63, wave = tid >> 6, lc = lx & 15, lq = lx >> 4;
    const int nAp = (nA + 3) & ~3;
    for (int i = tid; i < nA * nA; i += 256) cimg[i] = 0.f;
    __syncthreads();
    {
        const float* cf = coefs + (size_t)e * n_valid_act;
        for (int k0 = tid; k0 < n_valid_act; k0 += 4 * 256) {            // 4 independent (index, value) pairs in flight
            int ix[4];
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = k0 + 256 * q < n_valid_act ? k0 + 256 * q : k0;
                ix[q] = act_idx[k];
                v[q] = cf[k];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) if (k0 + 256 * q < n_valid_act) cimg[ix[q]] = v[q];
        }
    }
    __syncthreads();
    for (int rt = wave; rt < 8; rt += 4) {
        float av[KS];
        const f32x4* src = reinterpret_cast<const f32x4*>(gya) + (size_t)((y0 >> 4) + rt) * (ga_stride >> 2) * 64 + lx;
#pragma unroll
        for (int q4 = 0; q4 < KS / 4; ++q4) {
            const f32x4 t = src[4 * q4 < ga_stride ? 64 * q4 : 0];
#pragma unroll
            for (int d = 0; d < 4; ++d) av[4 * q4 + d] = 4 * q4 < ga_stride ? t[d] : 0.f;     // A[i = lane & 15 -> row][k = lane >> 4]
        }
        for (int ct = 0; 16 * ct < nAp; ++ct) {
            const int ix = 16 * ct + lc;
            float bv[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int iy = lq + 4 * ks;
                const bool ok = iy < nA && ix < nA;
                const float t = cimg[ok ? iy * nA + ix : 0];
                bv[ks] = ok ? t : 0.f;                           // B[k = lane >> 4][j = lane & 15]
            }
            f32x4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KS; ++ks)
                if (4 * ks < nAp) d = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], bv[ks], d, 0, 0, 0);
            if (ix < nAp) {                                      // d[r] = s1[y0 + 16 rt + 4 lq + r][ix]; rows >= R are zero (gya is)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s1a[(size_t)e * Rp * 4 * ga_stride + ga_index(y0 + 16 * rt + 4 * lq + r, ix, ga_stride)] = d[r];
            }
        }
    }
}
int launch_dm_rows(const float* coefs, const int* act_idx, const float* gya, float* s1a, int n_env, int R, int n_act, int n_valid_act,
                   int ga_stride, hipStream_t st) {
    const size_t lds = sizeof(float) * (size_t)n_act * n_act;
    dim3 grid(cdiv(R, 128), n_env);
    if (n_act <= 32)
        hipLaunchKernelGGL(k_dm_rows<8>, grid, dim3(256), lds, st, coefs, act_idx, gya, s1a, R, n_act, n_valid_act, ga_stride);
    else if (n_act <= 128)
        hipLaunchKernelGGL(k_dm_rows<32>, grid, dim3(256), lds, st, coefs, act_idx, gya, s1a, R, n_act, n_valid_act, ga_stride);
    else
        return -1;
    AO_HIP(hipGetLastError());
    return 0;
}

template <typename T>
int launch_coefs_image(const T* coefs, const int* act_idx, T* img, int n_env, int n_act, int n_valid_act, hipStream_t st) {
    hipLaunchKernelGGL(k_coefs_image<T>, dim3(1, n_env), dim3(256), 0, st, coefs, act_idx, img, n_act, n_valid_act);
    AO_HIP(hipGetLastError());
    return 0;
}
template int launch_coefs_image<float>(const float*, const int*, float*, int, int, int, hipStream_t);
template int launch_coefs_image<double>(const double*, const int*, double*, int, int, int, hipStream_t);

int phase_tx(int R, int n_act, size_t esz) {
    int tx = R < kTXmax ? R : kTXmax;
    while (tx > 16 && phase_lds_bytes(n_act, tx, esz) > 160 * 1024) tx = (tx + 1) / 2;
    return tx;
}
int phase_tiles(int R, int n_act, size_t esz) {
    return cdiv(R, phase_tx(R, n_act, esz)) * cdiv(R, kTY);
}

template <typename T>
KArgs<T> make_phase_kargs(const PhaseArgs& pa, const PhaseBuffers<T>& pb, int R, int n_act, int n_valid_act,
                          double atm_wavelength, double src_wavelength) {
    KArgs<T> a;
    a.pa = pa;
    a.pb = pb;
    a.R = R;
    a.n_act = n_act;
    a.n_valid_act = n_valid_act;
    a.tx = phase_tx(R, n_act, sizeof(T));
    a.rp = cdiv(R, kTXmax) * kTXmax;
    a.ablate = 0;
    a.atm_scale = (T)(atm_wavelength / 2 / 3.14159265358979323846);
    a.src_scale = (T)(6.283185307179586476925286766559 / src_wavelength);
    return a;
}
template KArgs<float> make_phase_kargs<float>(const PhaseArgs&, const PhaseBuffers<float>&, int, int, int, double, double);
template KArgs<double> make_phase_kargs<double>(const PhaseArgs&, const PhaseBuffers<double>&, int, int, int, double, double);

template <typename T>
int launch_phase(const PhaseArgs& pa, const PhaseBuffers<T>& pb, int n_env, int R, int n_act, int n_valid_act,
                 double atm_wavelength, double src_wavelength, int use_mfma, hipStream_t st) {
    KArgs<T> a = make_phase_kargs<T>(pa, pb, R, n_act, n_valid_act, atm_wavelength, src_wavelength);
    a.ablate = use_mfma >> 8;                      // diagnostic builds only (bench of kernel sections)
    if ((use_mfma & 1) && sizeof(T) == 4 && pb.dm_opd == nullptr) {
        // the MFMA kernel tiles with TX = 128 whatever R is (phase_tiles() is the same count for R <= 128 and
        // for R a multiple of 128; otherwise fall through to the generic kernel)
        if (cdiv(R, kTXmax) == cdiv(R, a.tx)) {
            const int rc = launch_phase_mfma<T>(a, n_env, st);
            if (rc >= 0) return rc;
        }
    }
    const int TX = a.tx;
    const size_t lds = phase_lds_bytes(n_act, TX, sizeof(T));
    if (lds > 160 * 1024) return fail("phase kernel: %d actuators across need %zu B of LDS", n_act, lds);
    if (lds > 64 * 1024)
        AO_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_phase<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)lds));
    dim3 grid(cdiv(R, TX), cdiv(R, kTY), n_env);
    hipLaunchKernelGGL(k_phase<T>, grid, dim3(64, 4), lds, st, a);
    AO_HIP(hipGetLastError());
    return 0;
}

template int launch_phase<float>(const PhaseArgs&, const PhaseBuffers<float>&, int, int, int, int, double, double, int,
                                 hipStream_t);
template int launch_phase<double>(const PhaseArgs&, const PhaseBuffers<double>&, int, int, int, int, double, double,
                                  int, hipStream_t);

}  // namespace ao
