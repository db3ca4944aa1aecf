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

template <int N1, int N2>
bool check_chain() {
    constexpr int N = N1 * N2;
    std::vector<cd> x(N), tw(N);
    std::vector<v2> xf(N), twf(N);
    unsigned s = 99u + N;
    for (int i = 0; i < N; ++i) {
        s = s * 1664525u + 1013904223u; const double a = (s >> 8) / 16777216.0 - 0.5;
        s = s * 1664525u + 1013904223u; const double b = (s >> 8) / 16777216.0 - 0.5;
        x[i] = cd(a, b); xf[i] = v2{(float)a, (float)b};
        tw[i] = std::polar(1.0, -2 * PI * i / N); twf[i] = v2{(float)tw[i].real(), (float)tw[i].imag()};
    }
    // forward with lanes n2 (length N1 over n1) then k1 (length N2 over n2); inverse with lanes k1 then m1
    std::vector<v2> Z(N), X(N), V(N), out(N);
    for (int n2 = 0; n2 < N2; ++n2) {
        v2 v[N1];
        for (int n1 = 0; n1 < N1; ++n1) v[n1] = xf[N2 * n1 + n2];
        dft_len<N1, false>(v);
        for (int k1 = 0; k1 < N1; ++k1) Z[n2 * N1 + k1] = cmul_tw<false>(v[k1], twf[n2 * k1]);
    }
    for (int k1 = 0; k1 < N1; ++k1) {
        v2 v[N2];
        for (int n2 = 0; n2 < N2; ++n2) v[n2] = Z[n2 * N1 + k1];
        dft_len<N2, false>(v);
        for (int k2 = 0; k2 < N2; ++k2) X[k1 + N1 * k2] = v[k2];
    }
    double ef = 0, ei = 0, scale = 0;
    for (int k = 0; k < N; ++k) {
        cd acc = 0;
        for (int i = 0; i < N; ++i) acc += x[i] * tw[(k * i) % N];
        scale = std::fmax(scale, std::abs(acc));
        ef = std::fmax(ef, std::abs(acc - cd(X[k].x, X[k].y)));
    }
    for (int k1 = 0; k1 < N1; ++k1) {
        v2 v[N2];
        for (int k2 = 0; k2 < N2; ++k2) v[k2] = X[k1 + N1 * k2];
        dft_len<N2, true>(v);
        for (int m1 = 0; m1 < N2; ++m1) V[k1 * N2 + m1] = cmul_tw<true>(v[m1], twf[k1 * m1]);
    }
    for (int m1 = 0; m1 < N2; ++m1) {
        v2 v[N1];
        for (int k1 = 0; k1 < N1; ++k1) v[k1] = V[k1 * N2 + m1];
        dft_len<N1, true>(v);
        for (int m2 = 0; m2 < N1; ++m2) out[m1 + N2 * m2] = v[m2];
    }
    for (int i = 0; i < N; ++i) ei = std::fmax(ei, std::abs(cd(out[i].x, out[i].y) / (double)N - x[i]));
    std::printf("fft%d = %d x %d: fwd err %.2e of max %.2f   round trip err %.2e\n", N, N1, N2, ef, scale, ei);
    return ef < 2e-5 * scale && ei < 2e-6;
}

int main() {
    double e24f = check_small<24, false>([](v2 (&v)[24]) { dft24<false>(v); });
    double e24i = check_small<24, true>([](v2 (&v)[24]) { dft24<true>(v); });
    double e22f = check_small<22, false>([](v2 (&v)[22]) { dft22<false>(v); });
    double e22i = check_small<22, true>([](v2 (&v)[22]) { dft22<true>(v); });
    double e16f = check_small<16, false>([](v2 (&v)[16]) { dft16<false>(v); });
    double e16i = check_small<16, true>([](v2 (&v)[16]) { dft16<true>(v); });
    double e18f = check_small<18, false>([](v2 (&v)[18]) { dft18<false>(v); });
    double e18i = check_small<18, true>([](v2 (&v)[18]) { dft18<true>(v); });
    std::printf("dft24 fwd %.2e inv %.2e   dft22 fwd %.2e inv %.2e   dft16 fwd %.2e inv %.2e   dft18 fwd %.2e inv %.2e\n", e24f, e24i, e22f, e22i, e16f, e16i,
                e18f, e18i);
    const bool c528 = check_chain<24, 22>(), c288 = check_chain<16, 18>();
    const bool ok = e24f < 3e-6 && e24i < 3e-6 && e22f < 3e-6 && e22i < 3e-6 && e16f < 3e-6 && e16i < 3e-6 && e18f < 3e-6 && e18i < 3e-6 && c528 && c288;
    std::printf(ok ? "PASS\n" : "FAIL\n");
    return ok ? 0 : 1;
}
