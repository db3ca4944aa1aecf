// VALU issue-rate microbenchmark (gfx950): cycles per wave-instruction per SIMD of the instruction mixes the camera model and
// the spots stage are made of, at 1, 2 and 4 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

enum { FMA8, PKFMA8, FMA1, MADU64, CNDMASK, ALIGNBIT, RCP, PHILOX, XOSHIRO, CMPADD, N_MODES };
static const char* kNames[N_MODES] = {"v_fma_f32 x8 indep", "v_pk_fma_f32 x8 indep", "v_fma_f32 1 chain", "v_mad_u64_u32 x4 indep",
                                      "v_cmp+v_cndmask x4", "v_alignbit x8 indep", "v_rcp_f32 x8 indep", "philox4x32-7 call",
                                      "xoshiro128++ x4 words", "v_cmp+v_addc x4"};
// instructions (or calls) per loop turn, for the per-instruction figures
static const int kPerTurn[N_MODES] = {64, 64, 64, 32, 64, 64, 64, 1, 1, 64};

template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* cyc, int iters) {
    float a[8];
    v2f p[8];
    uint32_t x[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x + i; p[i] = v2f{a[i], a[i] + 1}; x[i] = threadIdx.x * 2654435761u + i; }
    const float m = 1.0001f, c = 0.5f;
    const v2f mm = {m, m}, cc = {c, c};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == FMA8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
        } else if (MODE == PKFMA8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(mm), "v"(cc));
        } else if (MODE == FMA1) {
#pragma unroll
            for (int u = 0; u < 64; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[0]) : "v"(m), "v"(c));
        } else if (MODE == MADU64) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const uint64_t pr = (uint64_t)0xD2511F53u * x[i];
                    x[i] = (uint32_t)(pr >> 32) ^ (uint32_t)pr;
                    asm volatile("" : "+v"(x[i]));
                }
        } else if (MODE == CNDMASK) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = a[i] > a[i + 4] ? a[i + 4] : c;
                    asm volatile("" : "+v"(a[i]));
                }
        } else if (MODE == ALIGNBIT) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(x[i]) : "v"(x[(i + 1) & 7]));
        } else if (MODE == RCP) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
        } else if (MODE == PHILOX) {
            uint32_t c0 = x[0], c1 = x[1], c2 = x[2], c3 = x[3], k0 = 17u, k1 = 29u;
#pragma unroll
            for (int r = 0; r < 7; ++r) {
                const uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
                const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
                c0 = n0; c1 = (uint32_t)p1; c2 = n2; c3 = (uint32_t)p0;
                k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
            }
            x[0] = c0; x[1] = c1; x[2] = c2; x[3] = c3;
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(x[i]));
        } else if (MODE == XOSHIRO) {
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const uint32_t s03 = x[0] + x[3];
                x[4 + w] ^= ((s03 << 7) | (s03 >> 25)) + x[0];
                const uint32_t t = x[1] << 9;
                x[2] ^= x[0]; x[3] ^= x[1]; x[1] ^= x[2]; x[0] ^= x[3]; x[2] ^= t;
                x[3] = (x[3] << 11) | (x[3] >> 21);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(x[i]));
        } else if (MODE == CMPADD) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] += a[i + 4] > c ? 1.f : 0.f;
                    asm volatile("" : "+v"(a[i]));
                }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(float* out, unsigned long long* cyc, int iters) {
    for (int threads : {64, 256, 512, 1024}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
            hipDeviceSynchronize();
        }
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double wps = threads >= 256 ? threads / 256.0 : 1;     // waves per busy SIMD
        const double ns = ms * 1e6 / ((double)iters * kPerTurn[MODE] * wps);
        printf("%-24s waves/SIMD %.2f: kernel %8.1f us => %.3f ns = %.2f cycles (2.4 GHz) per wave-%s per SIMD\n", kNames[MODE],
               threads / 256.0, ms * 1e3, ns, ns * 2.4, kPerTurn[MODE] == 1 ? "call" : "instruction");
    }
}

int main() {
    float* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 4000;
    run<FMA8>(out, cyc, iters);
    run<PKFMA8>(out, cyc, iters);
    run<FMA1>(out, cyc, iters);
    run<MADU64>(out, cyc, iters);
    run<CNDMASK>(out, cyc, iters);
    run<CMPADD>(out, cyc, iters);
    run<ALIGNBIT>(out, cyc, iters);
    run<RCP>(out, cyc, iters);
    run<PHILOX>(out, cyc, iters * 8);
    run<XOSHIRO>(out, cyc, iters * 8);
    return 0;
}
