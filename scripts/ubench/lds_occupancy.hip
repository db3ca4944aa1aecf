// How many workgroups does a CU of gfx950 hold at a time as a function of their LDS size?  Each workgroup spins for 10 us (100 MHz
// s_memrealtime) and records its CU; 16 workgroups per CU are launched: elapsed / 10 us = rounds, and the overlap per CU is counted
// directly from the recorded intervals.   hipcc -O3 --offload-arch=gfx950 lds_occupancy.hip -o lds_occupancy && ./lds_occupancy
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <map>
#include <vector>

__global__ void k_spin(unsigned long long* rec, int lds_words, int spin_ticks) {
    extern __shared__ float lds[];
    if (threadIdx.x == 0) lds[lds_words - 1] = 1.f;                // touch the end of the allocation
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        rec[3 * blockIdx.x] = t0;
        rec[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
        rec[3 * blockIdx.x + 2] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | __builtin_amdgcn_s_getreg(63492);
    }
}

int main() {
    const int n_wg = 256 * 16;
    unsigned long long* d;
    hipMalloc(&d, sizeof(unsigned long long) * 3 * n_wg);
    std::vector<unsigned long long> h(3 * n_wg);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_spin), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int lanes : {64, 192, 384}) {
        for (int kb : {4, 16, 20, 24, 32, 36, 40, 44, 48, 52, 56, 64, 72, 80, 96, 128, 160}) {
            const size_t bytes = (size_t)kb * 1024 - (kb == 160 ? 0 : 0);
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_spin, dim3(n_wg), dim3(lanes), bytes, 0, d, (int)(bytes / 4), 1000);
            hipEventRecord(e1);
            if (hipDeviceSynchronize() != hipSuccess) { std::printf("%3d KB x %3d lanes: launch failed\n", kb, lanes); continue; }
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), d, sizeof(unsigned long long) * 3 * n_wg, hipMemcpyDeviceToHost);
            std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> per_cu;
            for (int i = 0; i < n_wg; ++i) {
                const unsigned long long hw = h[3 * i + 2], key = ((hw >> 32) & 0xF) << 16 | (hw & 0xFF00);   // XCC, SE / SH / CU
                per_cu[key].push_back({h[3 * i], 1});
                per_cu[key].push_back({h[3 * i + 1], -1});
            }
            int best_max = 0, best_min = 1 << 30;
            for (auto& kv : per_cu) {
                std::sort(kv.second.begin(), kv.second.end());
                int cur = 0, best = 0;
                for (auto& ev : kv.second) { cur += ev.second; best = std::max(best, cur); }
                best_max = std::max(best_max, best);
                best_min = std::min(best_min, best);
            }
            std::printf("%3d KB x %3d lanes: %7.1f us for 16 workgroups of 10 us per CU -> %.1f at a time; per-CU overlap %d..%d on %zu CUs\n", kb, lanes,
                        1e3 * ms, 160.0 / (1e3 * ms), best_min, best_max, per_cu.size());
        }
    }
    return 0;
}
