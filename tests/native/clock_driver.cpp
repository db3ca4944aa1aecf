// Host-side check of the atmosphere clock the library and its kernels share (rlao_amd/csrc/common.hpp: clock_subpixel, taps_from_buff):
// prints, for every "rx ry n" line on stdin, n steps of one layer's clock from buff = 0: "bx by buff_x buff_y dy dx wy0..3 wx0..3".
// Built with hipcc for the HOST only (tests/test_host_logic.py); the same source is compiled into k_ring_prepare_env for the device.
#include <cstdio>

#include "common.hpp"

int main() {
    double rx, ry;
    int n;
    while (std::scanf("%lf %lf %d", &rx, &ry, &n) == 3) {
        const double ratio[2] = {rx, ry};
        double buff[2] = {0, 0};
        for (int i = 0; i < n; ++i) {
            int bx, by;
            ao::clock_subpixel(ratio, buff, &bx, &by);
            ao::LayerTaps t{};
            ao::taps_from_buff(buff, t);
            std::printf("%d %d %.17g %.17g %d %d", bx, by, buff[0], buff[1], t.dy, t.dx);
            for (int k = 0; k < 4; ++k) std::printf(" %.17g", t.wy[k]);
            for (int k = 0; k < 4; ++k) std::printf(" %.17g", t.wx[k]);
            std::printf("\n");
        }
    }
    return 0;
}
