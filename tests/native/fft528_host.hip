// Host-side check of the register-resident factors of rlao_amd/csrc/fft528.hpp (the same inline functions the kernels use,
// compiled for the CPU): 24- and 22-point transforms against the O(n^2) definition in float64, and the two-factor 528-point
// forward / inverse chain with the lane <-> index maps of pyr528_kernels.hip.  Prints the maximum errors; exit code 0 = pass.
#include <cmath>
#include <complex>
#include <cstdio>
#include <vector>
#include "../../rlao_amd/csrc/fft528.hpp"
using namespace ao::f528;
typedef std::complex<double> cd;
static const double PI = 3.14159265358979323846;

template <int n, bool INV, typename F>
double check_small(F f) {
    v2 v[n];
    std::vector<cd> x(n);
    unsigned s = 12345u + n + INV;
    for (int i = 0; i < n; ++i) {
        s = s * 1664525u + 1013904223u; const double a = (s >> 8) / 16777216.0 - 0.5;
        s = s * 1664525u + 1013904223u; const double b = (s >> 8) / 16777216.0 - 0.5;
        x[i] = cd(a, b);
        v[i] = v2{(float)a, (float)b};
    }
    f(v);
    double err = 0;
    for (int k = 0; k < n; ++k) {
        cd acc = 0;
        for (int i = 0; i < n; ++i) acc += x[i] * std::polar(1.0, (INV ? 2 : -2) * PI * ((k * i) % n) / n);
        err = std::fmax(err, std::abs(acc - cd(v[k].x, v[k].y)));
    }
    return err;
}

int main() {
    double e24f = check_small<24, false>([](v2 (&v)[24]) { dft24<false>(v); });
    double e24i = check_small<24, true>([](v2 (&v)[24]) { dft24<true>(v); });
    double e22f = check_small<22, false>([](v2 (&v)[22]) { dft22<false>(v); });
    double e22i = check_small<22, true>([](v2 (&v)[22]) { dft22<true>(v); });
    std::printf("dft24 fwd %.2e inv %.2e   dft22 fwd %.2e inv %.2e\n", e24f, e24i, e22f, e22i);
    // 528 = 24 x 22: forward with lanes n2 then k1, inverse with lanes k1 then m1
    std::vector<cd> x(kN), tw(kN);
    std::vector<v2> xf(kN), twf(kN);
    unsigned s = 99u;
    for (int i = 0; i < kN; ++i) {
        s = s * 1664525u + 1013904223u; const double a = (s >> 8) / 16777216.0 - 0.5;
        s = s * 1664525u + 1013904223u; const double b = (s >> 8) / 16777216.0 - 0.5;
        x[i] = cd(a, b); xf[i] = v2{(float)a, (float)b};
        tw[i] = std::polar(1.0, -2 * PI * i / kN); twf[i] = v2{(float)tw[i].real(), (float)tw[i].imag()};
    }
    std::vector<v2> Z(kN), X(kN), V(kN), out(kN);
    for (int n2 = 0; n2 < kN2; ++n2) {
        v2 v[24];
        for (int n1 = 0; n1 < 24; ++n1) v[n1] = xf[22 * n1 + n2];
        dft24<false>(v);
        for (int k1 = 0; k1 < 24; ++k1) Z[n2 * 24 + k1] = cmul_tw<false>(v[k1], twf[n2 * k1]);
    }
    for (int k1 = 0; k1 < kN1; ++k1) {
        v2 v[22];
        for (int n2 = 0; n2 < 22; ++n2) v[n2] = Z[n2 * 24 + k1];
        dft22<false>(v);
        for (int k2 = 0; k2 < 22; ++k2) X[k1 + 24 * k2] = v[k2];
    }
    double ef = 0, ei = 0, scale = 0;
    std::vector<cd> Xr(kN);
    for (int k = 0; k < kN; ++k) {
        cd acc = 0;
        for (int i = 0; i < kN; ++i) acc += x[i] * tw[(k * i) % kN];
        Xr[k] = acc;
        scale = std::fmax(scale, std::abs(acc));
        ef = std::fmax(ef, std::abs(acc - cd(X[k].x, X[k].y)));
    }
    // inverse of the (exact, float-rounded) spectrum: lane k1 holds k1 + 24 k2
    for (int k1 = 0; k1 < kN1; ++k1) {
        v2 v[22];
        for (int k2 = 0; k2 < 22; ++k2) v[k2] = X[k1 + 24 * k2];
        dft22<true>(v);
        for (int m1 = 0; m1 < 22; ++m1) V[k1 * 22 + m1] = cmul_tw<true>(v[m1], twf[k1 * m1]);
    }
    for (int m1 = 0; m1 < kN2; ++m1) {
        v2 v[24];
        for (int k1 = 0; k1 < 24; ++k1) v[k1] = V[k1 * 22 + m1];
        dft24<true>(v);
        for (int m2 = 0; m2 < 24; ++m2) out[m1 + 22 * m2] = v[m2];
    }
    for (int i = 0; i < kN; ++i) ei = std::fmax(ei, std::abs(cd(out[i].x, out[i].y) / (double)kN - x[i]));
    std::printf("fft528 fwd err %.2e of max %.2f   round trip err %.2e\n", ef, scale, ei);
    const bool ok = e24f < 3e-6 && e24i < 3e-6 && e22f < 3e-6 && e22i < 3e-6 && ef < 2e-5 * scale && ei < 2e-6;
    std::printf(ok ? "PASS\n" : "FAIL\n");
    return ok ? 0 : 1;
}
