// VALU issue-rate microbenchmark: cycles per wave-instruction of v_fma_f32 / v_pk_fma_f32 at 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* cyc, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 1.0001f, c = 0.5f;
    const v2f mm = {m, m}, cc = {c, c};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = fmaf(a0, m, c); a1 = fmaf(a1, m, c); a2 = fmaf(a2, m, c); a3 = fmaf(a3, m, c);
                a4 = fmaf(a4, m, c); a5 = fmaf(a5, m, c); a6 = fmaf(a6, m, c); a7 = fmaf(a7, m, c);
            }
        } else {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(mm), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(mm), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(mm), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(mm), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p4) : "v"(mm), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p5) : "v"(mm), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p6) : "v"(mm), "v"(cc));
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p7) : "v"(mm), "v"(cc));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 20000;
    for (int mode = 0; mode < 2; ++mode)
        for (int threads : {64, 256, 512, 1024}) {
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
            else hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
            hipEventRecord(e1, 0); hipDeviceSynchronize();
            float ms = 0; hipEventElapsedTime(&ms, e0, e1);
            unsigned long long h[256]; hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double per = (double)h[5] / (iters * 64.0);
            double wps = threads >= 256 ? threads / 256.0 : 1;     // waves per busy SIMD
            double ns_per_instr_simd = ms * 1e6 / (iters * 64.0 * wps);
            printf("%s threads/WG=%4d (waves/SIMD=%.2f): wave0 %.2f ticks/instr ; kernel %.1f us => %.3f ns per wave-instr per SIMD (= %.2f cycles at 2.4 GHz), tick rate %.2f GHz\n",
                   mode ? "v_pk_fma_f32" : "v_fma_f32   ", threads, threads / 256.0, per, ms * 1e3, ns_per_instr_simd, ns_per_instr_simd * 2.4, h[5] / (ms * 1e6));
        }
    return 0;
}
