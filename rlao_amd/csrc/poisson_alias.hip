// Host side of the photon-noise sampler (poisson_alias.hpp): the alias tables of Poisson(lambda0) on the grid lambda0 = j / 4
// (j < 128) and lambda0 = 32 c (c < 32), built in float64 with Vose's algorithm, and the test hooks of include/aoenv.h.
#include "poisson_alias.hpp"

#include <cmath>
#include <cstring>

#include "detector.hpp"

namespace ao {

namespace {

// the outcomes k = kmin .. kmin + n - 1 that carry all but ~2^-34 of the mass on either side, and their probabilities (renormalised)
void poisson_row(double lam, int* kmin, std::vector<double>& p) {
    p.clear();
    if (!(lam > 0)) { *kmin = 0; p.push_back(1.0); return; }
    const double eps = std::ldexp(1.0, -34);
    auto pmf = [&](int k) { return std::exp(k * std::log(lam) - lam - std::lgamma(k + 1.0)); };
    const int mode = (int)std::floor(lam);
    // walk outwards from the mode until the remaining tail (bounded by a geometric series) is below eps
    int lo = mode, hi = mode;
    {
        double t = pmf(lo);
        while (lo > 0) {                                           // tail below lo: <= pmf(lo - 1) / (1 - (lo - 1) / lam) for lo - 1 < lam
            const double next = t * lo / lam;                      // pmf(lo - 1)
            const double ratio = (lo - 1) / lam;
            if (next / (1.0 - (ratio < 1 ? ratio : 0.999999)) < eps) break;
            t = next;
            --lo;
        }
        t = pmf(hi);
        for (;;) {                                                 // tail above hi: <= pmf(hi + 1) / (1 - lam / (hi + 2))
            const double next = t * lam / (hi + 1);                // pmf(hi + 1)
            const double ratio = lam / (hi + 2);
            if (ratio < 1 && next / (1.0 - ratio) < eps) break;
            t = next;
            ++hi;
        }
    }
    *kmin = lo;
    double sum = 0;
    for (int k = lo; k <= hi; ++k) { p.push_back(pmf(k)); sum += p.back(); }
    for (double& v : p) v /= sum;
}

// Vose's alias construction: cell i keeps outcome i with probability q[i], else gives alias[i]
void vose(const std::vector<double>& p, std::vector<double>& q, std::vector<int>& alias) {
    const int n = (int)p.size();
    q.assign(n, 0.0);
    alias.assign(n, 0);
    std::vector<int> small, large;
    std::vector<double> s(n);
    for (int i = 0; i < n; ++i) { s[i] = p[i] * n; (s[i] < 1.0 ? small : large).push_back(i); }
    while (!small.empty() && !large.empty()) {
        const int a = small.back(), g = large.back();
        small.pop_back();
        large.pop_back();
        q[a] = s[a];
        alias[a] = g;
        s[g] = (s[g] + s[a]) - 1.0;
        (s[g] < 1.0 ? small : large).push_back(g);
    }
    for (int i : large) { q[i] = 1.0; alias[i] = i; }
    for (int i : small) { q[i] = 1.0; alias[i] = i; }              // (rounding left-overs)
}

}  // namespace

void build_poisson_alias(PoissonAliasHost& out) {
    using namespace palias;
    const int rows = kFineRows + kMaxCoarseRows;
    std::vector<uint32_t>& t = out.tab;
    t.assign(kHeader + 2 * rows, 0u);
    out.words_upto.assign(kMaxCoarseRows + 1, 0);
    std::vector<double> p, q;
    std::vector<int> alias;
    for (int r = 0; r < rows; ++r) {
        const double lam = r < kFineRows ? (double)kFineStep * r : (double)kCoarseStep * (r - kFineRows);
        int kmin = 0;
        poisson_row(lam, &kmin, p);
        vose(p, q, alias);
        const int n = (int)p.size();
        if (r >= kFineRows) out.words_upto[r - kFineRows] = (int)t.size();
        t[kHeader + 2 * r] = (uint32_t)t.size();
        t[kHeader + 2 * r + 1] = ((uint32_t)kmin << 16) | (uint32_t)n;       // (kmin < 2^16, n < 512: checked by the test hook's caller)
        for (int i = 0; i < n; ++i) {
            double thr = std::floor(q[i] * 8388608.0 + 0.5);
            int al = alias[i];
            if (thr >= 8388607.0) { thr = 8388607.0; al = i; }               // a full cell: both branches give the cell itself
            t.push_back(((uint32_t)thr << 9) | (uint32_t)al);
        }
    }
    while (t.size() % 4) t.push_back(0u);
    out.words_upto[kMaxCoarseRows] = (int)t.size();
    t[0] = kFineRows;
    t[1] = kMaxCoarseRows;
    t[2] = (uint32_t)t.size();
}

const PoissonAliasHost& poisson_alias_host() {
    static const PoissonAliasHost h = [] { PoissonAliasHost t; build_poisson_alias(t); return t; }();
    return h;
}

void PoissonAliasHost::prefix(int budget, int* words, float* lmax) const {
    int c = palias::kMaxCoarseRows;
    while (c > 0 && ((words_upto[c] + 3) & ~3) > budget) --c;
    *words = c > 0 ? ((words_upto[c] + 3) & ~3) : 0;
    *lmax = palias::kCoarseStep * (float)c;
}

// ---- test hook kernel: one draw per element, the same code path as the cameras (table read from global memory) ---------------
__global__ void k_test_poisson(const float* __restrict__ lam, float* __restrict__ out, int n, PoissonAlias pa, DetectorCfg d) {
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < n;
    const int i = live ? i0 : 0;
    uint32_t o[4], o2[4], o3[4];
    quad_bits((uint32_t)(i >> 2), 0u, d, kDrawPhoton, o);
    quad_bits((uint32_t)(i >> 2), 0u, d, kDrawPhoton2, o2);
    quad_bits((uint32_t)(i >> 2), 0u, d, kDrawPhoton3, o3);
    // (every lane draws its whole quad and keeps its own slot: photon_quad is the code path of the cameras)
    const int q4 = i & ~3;
    f32x4d v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v[s] = (live && q4 + s < n) ? fmaxf(lam[q4 + s], 0.f) : 0.f;
    const u32x4d pix = {(uint32_t)q4, (uint32_t)q4 + 1u, (uint32_t)q4 + 2u, (uint32_t)q4 + 3u};
    photon_quad(v, o, o2, o3, pix, 0u, d, pa.lmax, pa.tab);
    if (live) out[i] = v[i & 3];
}

}  // namespace ao

using namespace ao;

extern "C" {

int aoenv_test_poisson_table(uint32_t* h_out, size_t cap_words, size_t* h_words) {
    const PoissonAliasHost& h = poisson_alias_host();
    if (h_words) *h_words = h.tab.size();
    if (h_out) {
        if (cap_words < h.tab.size()) return fail("aoenv_test_poisson_table: %zu words needed, %zu given", h.tab.size(), cap_words);
        std::memcpy(h_out, h.tab.data(), h.tab.size() * 4);
    }
    return 0;
}

int aoenv_test_poisson(int device, const float* h_lambda, int n, uint64_t seed, uint32_t frame, float lmax, float* h_out) {
    if (!h_lambda || !h_out || n < 1) return fail("aoenv_test_poisson: bad arguments");
    AO_HIP(hipSetDevice(device));
    const PoissonAliasHost& h = poisson_alias_host();
    uint32_t* d_tab = nullptr;
    float *d_lam = nullptr, *d_out = nullptr;
    AO_HIP(hipMalloc((void**)&d_tab, h.tab.size() * 4));
    AO_HIP(hipMalloc((void**)&d_lam, (size_t)n * 4));
    AO_HIP(hipMalloc((void**)&d_out, (size_t)n * 4));
    AO_HIP(hipMemcpy(d_tab, h.tab.data(), h.tab.size() * 4, hipMemcpyHostToDevice));
    AO_HIP(hipMemcpy(d_lam, h_lambda, (size_t)n * 4, hipMemcpyHostToDevice));
    PoissonAlias pa{d_tab, (int)h.tab.size(), palias::kCoarseStep * palias::kMaxCoarseRows};
    if (lmax > 0 && lmax < pa.lmax) pa.lmax = palias::kCoarseStep * std::floor(lmax / palias::kCoarseStep);   // (exercise the PTRS hand-over)
    if (pa.lmax < palias::kCoarseStep) pa.lmax = palias::kCoarseStep;
    DetectorCfg d{};
    d.active = 1;
    d.photon_noise = 1;
    d.qe = d.gain = 1.f;
    d.seed_lo = (uint32_t)seed;
    d.seed_hi = (uint32_t)(seed >> 32);
    d.frame_counter = frame;
    hipLaunchKernelGGL(k_test_poisson, dim3(cdiv(n, 256)), dim3(256), 0, nullptr, d_lam, d_out, n, pa, d);
    AO_HIP(hipGetLastError());
    AO_HIP(hipMemcpy(h_out, d_out, (size_t)n * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_tab);
    (void)hipFree(d_lam);
    (void)hipFree(d_out);
    return 0;
}

}  // extern "C"
