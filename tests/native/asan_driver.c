/* Host-side sanitizer driver (AddressSanitizer + UBSan build of libaoenv, `make -C rlao_amd/csrc asan`): walks the C ABI's
 * argument validation and error paths -- everything that runs before a device is touched.  Exit code 0 and no sanitizer report
 * = pass.  On a machine without a GPU aoenv_create must fail cleanly, with an error string. */
#include <stdio.h>
#include <string.h>
#include "aoenv.h"

#define EXPECT(cond) do { if (!(cond)) { fprintf(stderr, "FAILED: %s (line %d): %s\n", #cond, __LINE__, aoenv_last_error()); return 1; } } while (0)

int main(void) {
    AoEnv* env = NULL;
    AoCfg cfg;
    memset(&cfg, 0, sizeof cfg);
    EXPECT(aoenv_abi_version() == AOENV_ABI_VERSION);
    EXPECT(aoenv_create(NULL, 0, &env) != 0);
    EXPECT(aoenv_create(&cfg, 0, NULL) != 0);
    cfg.abi_version = 999;
    EXPECT(aoenv_create(&cfg, 0, &env) != 0 && strstr(aoenv_last_error(), "ABI") != NULL);
    cfg.abi_version = AOENV_ABI_VERSION;
    cfg.dtype = 7;
    EXPECT(aoenv_create(&cfg, 0, &env) != 0 && strstr(aoenv_last_error(), "dtype") != NULL);
    cfg.dtype = AOENV_F32; cfg.n_env = 2; cfg.resolution = 24; cfg.n_layer = 99;
    EXPECT(aoenv_create(&cfg, 0, &env) != 0 && strstr(aoenv_last_error(), "n_layer") != NULL);
    cfg.n_layer = 1; cfg.n_subap = 5;                                  /* 24 % 5 != 0 */
    EXPECT(aoenv_create(&cfg, 0, &env) != 0);
    cfg.n_subap = 4; cfg.layer_res = 28; cfg.n_inner = 1; cfg.n_outer = 1;
    EXPECT(aoenv_create(&cfg, 0, &env) != 0 && strstr(aoenv_last_error(), "n_inner") != NULL);
    cfg.n_inner = 8 * 28 - 16; cfg.n_outer = 4 * 28 + 4; cfg.n_valid_subap = 12; cfg.n_signal = 23;
    EXPECT(aoenv_create(&cfg, 0, &env) != 0 && strstr(aoenv_last_error(), "n_signal") != NULL);
    cfg.n_signal = 24; cfg.wfs_type = 5;
    EXPECT(aoenv_create(&cfg, 0, &env) != 0 && strstr(aoenv_last_error(), "wfs_type") != NULL);
    cfg.wfs_type = AOENV_WFS_PYRAMID; cfg.pyr_n_res = 95; cfg.cam_res = 16;
    EXPECT(aoenv_create(&cfg, 0, &env) != 0 && strstr(aoenv_last_error(), "pyramid") != NULL);
    cfg.wfs_type = AOENV_WFS_SH; cfg.cam_res = 24; cfg.max_group = 0;
    EXPECT(aoenv_create(&cfg, 0, &env) != 0 && strstr(aoenv_last_error(), "max_group") != NULL);
    cfg.max_group = 1; cfg.n_act = 5; cfg.n_valid_act = 21; cfg.dm_separable = 1; cfg.n_loop = 8;
    cfg.atm_wavelength = 500e-9; cfg.src_wavelength = 790e-9; cfg.leak = 0.99; cfg.threshold_cog = 0.01;
    {
        const int rc = aoenv_create(&cfg, 0, &env);                    /* a valid configuration: succeeds only where a GPU exists */
        if (rc == 0) {
            double buff[2] = {0.5, -0.25}, back[2] = {0, 0};
            EXPECT(aoenv_set_buff(env, buff) == 0 && aoenv_get_buff(env, back) == 0 && back[0] == 0.5 && back[1] == -0.25);
            buff[0] = 1.5;
            EXPECT(aoenv_set_buff(env, buff) != 0);
            EXPECT(aoenv_upload(env, 999, buff, 8) != 0);
            EXPECT(aoenv_upload(env, AOENV_C_PUPIL, buff, 3) != 0);
            EXPECT(aoenv_step(env, 0, NULL, NULL, NULL, NULL, NULL, NULL) != 0);
            EXPECT(aoenv_destroy(env) == 0);
        } else {
            EXPECT(strlen(aoenv_last_error()) > 0);
        }
    }
    EXPECT(aoenv_destroy(NULL) == 0);
    EXPECT(aoenv_get_buff(NULL, NULL) != 0 && aoenv_set_buff(NULL, NULL) != 0);
    EXPECT(aoenv_measure(NULL, NULL) != 0 && aoenv_set_option(NULL, 0, 0) != 0 && aoenv_set_detector(NULL, NULL, NULL) != 0);
    EXPECT(aoenv_test_normal(0, 1u, 3, 1, NULL) != 0);
    printf("asan driver ok\n");
    return 0;
}
