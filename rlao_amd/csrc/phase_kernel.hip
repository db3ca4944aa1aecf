// Residual-phase kernel: frozen-flow sub-pixel translation of every layer + footprint crop + layer sum
// + deformable-mirror surface + pupil + phase scaling + the scalar telemetry reductions.
//
// Reference stages fused here (one workgroup per env):
//   OOPAO/Atmosphere.py:406-407   layer.phase = warp(mapShift, translate(buff), order=3)[1:-1,1:-1]
//   OOPAO/Atmosphere.py:439-450   phase_support += phase[footprint] * sqrt(fractionalR0)
//   OOPAO/Atmosphere.py:474-477   OPD_no_pupil = phase_support * lambda_500 / 2 pi ; OPD = . * pupil
//   MAIN/OOPAOEnv/OOPAOEnv.py:497 total[i] = std(tel.OPD[pupil]) * 1e9
//   OOPAO/DeformableMirror.py:556 dm.OPD = modes @ coefs            (separable: Gy . C . Gx^T)
//   OOPAO/DeformableMirror.py:469 + Telescope.py:540-542   OPD = (OPD_atm + dm.OPD) * pupil
//   OOPAO/Telescope.py:404-412    src.phase = OPD * 2 pi / lambda_src
//   MAIN/OOPAOEnv/OOPAOEnv.py:522,554-555  residual[i] = std(OPD[pupil])*1e9 ; strehl = exp(-var(phase[pupil]))
#include "common.hpp"

namespace ao {

template <typename T>
struct KArgs {
    PhaseArgs pa;
    PhaseBuffers<T> pb;
    int R, n_act, n_valid_act, n_pupil, xchunk;
    T atm_scale;   // lambda_atm / 2 pi
    T src_scale;   // 2 pi / lambda_src
};

__device__ inline double wave_sum(double v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}

template <typename T>
__global__ void __launch_bounds__(1024) k_phase(const KArgs<T> a) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    T* cimg = reinterpret_cast<T*>(lds_raw);            // [n_act][n_act] command image
    T* t1 = cimg + a.n_act * a.n_act;                   // [n_act][xchunk]  C . Gx^T for one column chunk
    __shared__ double red[4][16];

    const int e = blockIdx.x;
    const int R = a.R, nA = a.n_act, S = a.pa.S;
    const size_t pix0 = (size_t)e * R * R;
    const bool separable = (a.pb.dm_opd == nullptr);

    if (separable) {
        for (int i = threadIdx.x; i < nA * nA; i += blockDim.x) cimg[i] = (T)0;
        __syncthreads();
        const T* cf = a.pb.coefs + (size_t)e * a.n_valid_act;
        for (int k = threadIdx.x; k < a.n_valid_act; k += blockDim.x) cimg[a.pb.act_idx[k]] = cf[k];
    }
    if (threadIdx.x == 0) a.pb.wfs_max[e] = (T)0;

    double s_atm = 0.0, q_atm = 0.0, s_res = 0.0, q_res = 0.0;

    for (int x0 = 0; x0 < R; x0 += a.xchunk) {
        const int xw = min(a.xchunk, R - x0);
        __syncthreads();
        if (separable) {
            // t1[iy][x] = sum_ix C[iy][ix] * gx[x][ix]
            for (int i = threadIdx.x; i < nA * xw; i += blockDim.x) {
                const int iy = i / xw, x = i % xw;
                const T* g = a.pb.gx + (size_t)(x0 + x) * nA;
                const T* c = cimg + iy * nA;
                T acc = (T)0;
                for (int ix = 0; ix < nA; ++ix) acc += c[ix] * g[ix];
                t1[iy * a.xchunk + x] = acc;
            }
            __syncthreads();
        }
        for (int i = threadIdx.x; i < R * xw; i += blockDim.x) {
            const int y = i / xw, xl = i % xw, x = x0 + xl;
            const size_t p = pix0 + (size_t)y * R + x;
            T atm;
            if (a.pa.update_atm) {
                T sup = (T)0;
                for (int l = 0; l < a.pa.n_layer; ++l) {
                    const LayerTaps& tp = a.pa.taps[l];
                    const T* map = static_cast<const T*>(a.pa.screen[l]) + (size_t)e * S * S;
                    const int r0 = y + a.pa.foot + tp.dy - 1, c0 = x + a.pa.foot + tp.dx - 1;
                    T v = (T)0;
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) {
                        const int rr = r0 + pr;
                        T row = (T)0;
                        if (rr >= 0 && rr < S) {
                            const T* m = map + (size_t)rr * S;
#pragma unroll
                            for (int pc = 0; pc < 4; ++pc) {
                                const int cc = c0 + pc;
                                const T f = (cc >= 0 && cc < S) ? m[cc] : (T)0;
                                row += (T)tp.wx[pc] * f;
                            }
                        }
                        v += (T)tp.wy[pr] * row;
                    }
                    // skimage clip=True: clamp to the input range, keep exact zeros when 0 is outside it
                    const T* mm = static_cast<const T*>(a.pa.minmax[l]) + 2 * e;
                    const T lo = mm[0], hi = mm[1];
                    if (!((lo > (T)0 || hi < (T)0) && v == (T)0)) v = v < lo ? lo : (v > hi ? hi : v);
                    sup += v * (T)tp.weight;
                }
                atm = sup * a.atm_scale;
                a.pb.opd_atm[p] = atm;
            } else {
                atm = a.pb.opd_atm[p];
            }
            T dm;
            if (separable) {
                const T* g = a.pb.gy + (size_t)y * nA;
                dm = (T)0;
                for (int iy = 0; iy < nA; ++iy) dm += g[iy] * t1[iy * a.xchunk + xl];
            } else {
                dm = a.pb.dm_opd[p];
            }
            const bool in = a.pb.pupil[(size_t)y * R + x] != 0;
            const T res = in ? (atm + dm) : (T)0;
            a.pb.phase[p] = res * a.src_scale;
            if (in) {
                const double da = (double)atm, dr = (double)res;
                s_atm += da;
                q_atm += da * da;
                s_res += dr;
                q_res += dr * dr;
            }
        }
    }

    s_atm = wave_sum(s_atm);
    q_atm = wave_sum(q_atm);
    s_res = wave_sum(s_res);
    q_res = wave_sum(q_res);
    const int w = threadIdx.x / kWave;
    if ((threadIdx.x & (kWave - 1)) == 0) {
        red[0][w] = s_atm;
        red[1][w] = q_atm;
        red[2][w] = s_res;
        red[3][w] = q_res;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        double v[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; ++k)
            for (int i = 0; i < (int)blockDim.x / kWave; ++i) v[k] += red[k][i];
        const double n = (double)a.n_pupil;
        double var_atm = v[1] / n - (v[0] / n) * (v[0] / n);
        double var_res = v[3] / n - (v[2] / n) * (v[2] / n);
        var_atm = var_atm > 0 ? var_atm : 0;
        var_res = var_res > 0 ? var_res : 0;
        const double total = sqrt(var_atm) * 1e9, resid = sqrt(var_res) * 1e9;
        const double k2 = (double)a.src_scale * (double)a.src_scale;
        T* sc = a.pb.scal + 4 * e;
        sc[0] = (T)total;
        sc[1] = (T)resid;
        sc[2] = (T)exp(-var_res * k2);
        if (a.pa.telemetry_index >= 0) {
            const size_t o = (size_t)a.pa.telemetry_index * gridDim.x + e;
            a.pb.total[o] = (T)total;
            a.pb.residual[o] = (T)resid;
        }
    }
}

template <typename T>
int launch_phase(const PhaseArgs& pa, const PhaseBuffers<T>& pb, int n_env, int R, int n_act, int n_valid_act,
                 int n_pupil, double atm_wavelength, double src_wavelength, hipStream_t st) {
    KArgs<T> a;
    a.pa = pa;
    a.pb = pb;
    a.R = R;
    a.n_act = n_act;
    a.n_valid_act = n_valid_act;
    a.n_pupil = n_pupil;
    const double two_pi = 6.283185307179586476925286766559;
    a.atm_scale = (T)(atm_wavelength / 2 / 3.14159265358979323846);
    a.src_scale = (T)(two_pi / src_wavelength);
    // column chunk so that the C.Gx^T slab stays within 48 KiB of LDS
    int xchunk = R;
    const size_t budget = 48 * 1024;
    while ((size_t)n_act * xchunk * sizeof(T) > budget && xchunk > 8) xchunk = (xchunk + 1) / 2;
    a.xchunk = xchunk;
    const size_t lds = ((size_t)n_act * n_act + (size_t)n_act * xchunk) * sizeof(T);
    hipLaunchKernelGGL(k_phase<T>, dim3(n_env), dim3(1024), lds, st, a);
    AO_HIP(hipGetLastError());
    return 0;
}

template int launch_phase<float>(const PhaseArgs&, const PhaseBuffers<float>&, int, int, int, int, int, double,
                                 double, hipStream_t);
template int launch_phase<double>(const PhaseArgs&, const PhaseBuffers<double>&, int, int, int, int, int, double,
                                  double, hipStream_t);

}  // namespace ao
