"""ctypes binding of libaoenv.so (include/aoenv.h).  There is no CPU fallback: if the HIP library is
missing or does not load, importing the binding raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AOENV_LIB") or os.path.join(_HERE, "csrc", "libaoenv.so")   # AOENV_LIB: A/B kernel builds

ABI_VERSION = 6
F32, F64 = 0, 1
WFS_SH, WFS_PYRAMID = 0, 1

# enum AoConst
(C_PUPIL, C_AB, C_INNER_IDX, C_OUTER_IDX, C_LAYER_WEIGHT, C_DM_GX, C_DM_GY, C_DM_MODES, C_ACT_IDX, C_WFS_AMP,
 C_SH_SUBAP_IDX, C_SH_REF, C_WFS_UNITS, C_RECON, C_PYR_MASK, C_PYR_TT, C_RECON_FACTORS) = range(17)
# enum AoBuf
(B_SCREEN, B_OPD_ATM, B_COEFS, B_PHASE, B_FRAME, B_SIGNAL, B_TOTAL, B_RESIDUAL, B_WFS_MAX, B_XI, B_MT_STATE,
 B_COUNTERS, B_DM_PREV) = range(13)


(OPT_FAST_WFS, OPT_MFMA_GEMM, OPT_FAST_TRIG, OPT_STORE_ATM_OPD, OPT_FUSED_TAIL, OPT_FUSED_STEP, OPT_DEFER_RING, OPT_COEFS_IMAGE,
 OPT_FACTORED_RECON, OPT_RING_LOOKAHEAD) = range(10)
KERNEL_NAMES = ("ring_prepare", "mt_normal", "gemm_ring", "ring_scatter", "phase", "sh_spots", "sh_centroid",
                "gemm_recon", "recon_finish", "pyramid", "sh_tail", "env_step", "detector")


class AoCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "dtype", "n_env", "resolution", "n_layer", "layer_res", "n_inner", "n_outer", "n_act",
        "n_valid_act", "dm_separable", "wfs_type", "n_subap", "n_valid_subap", "n_signal", "cam_res", "n_loop",
        "max_group", "pyr_n_res", "pyr_n_theta", "pyr_centering", "pyr_norm_valid", "pyr_q_lo", "pyr_q_hi")] + [
            ("layer_res_l", C.c_int32 * 8)] + [(n, C.c_double) for n in (
            "atm_wavelength", "src_wavelength", "leak", "threshold_cog")]


class AoDetector(C.Structure):
    _fields_ = [("photon_noise", C.c_int32), ("bits", C.c_int32), ("emccd", C.c_int32), ("env_index_offset", C.c_int32),
                ("qe", C.c_double), ("dark_electrons", C.c_double), ("fwc", C.c_double), ("gain", C.c_double),
                ("readout_noise", C.c_double), ("seed", C.c_uint64)]


EXPORTS = {
    # name: (restype, argtypes)
    "aoenv_last_error": (C.c_char_p, []),
    "aoenv_abi_version": (C.c_int, []),
    "aoenv_create": (C.c_int, [C.POINTER(AoCfg), C.c_int, C.POINTER(C.c_void_p)]),
    "aoenv_destroy": (C.c_int, [C.c_void_p]),
    "aoenv_upload": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t]),
    "aoenv_upload_layer": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "aoenv_set_wind": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "aoenv_new_screens": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aoenv_new_screens_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double,
                                           C.c_void_p]),
    "aoenv_set_atm_opd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "aoenv_set_coefs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "aoenv_measure": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aoenv_fused_step_active": (C.c_int, [C.c_void_p]),
    "aoenv_atm_update": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aoenv_reset_soft": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "aoenv_step": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                             C.c_void_p]),
    "aoenv_run_integrator": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p, C.c_void_p,
                                       C.c_void_p, C.c_void_p, C.c_void_p]),
    "aoenv_compute_psf": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "aoenv_set_detector": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "aoenv_set_return_accumulator": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aoenv_buffer": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "aoenv_download": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "aoenv_upload_state": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "aoenv_set_wind_env": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "aoenv_get_clock_env": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aoenv_set_clock_env": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aoenv_get_buff": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aoenv_set_buff": (C.c_int, [C.c_void_p, C.c_void_p]),
    "aoenv_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "aoenv_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "aoenv_profile_read": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "aoenv_test_normal": (C.c_int, [C.c_int, C.c_uint32, C.c_int, C.c_int, C.c_void_p]),
    "aoenv_test_poisson_table": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "aoenv_test_poisson": (C.c_int, [C.c_int, C.c_void_p, C.c_int, C.c_uint64, C.c_uint32, C.c_float, C.c_void_p]),
}


class AoEnvError(RuntimeError):
    pass


_lib = None


def load():
    """Load libaoenv.so and declare every entry point of include/aoenv.h.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise AoEnvError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback)")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1.
    # Importing torch first makes the dynamic loader resolve libaoenv's dependency (same SONAMEs) to the
    # copies torch already mapped, so torch streams / device pointers and libaoenv kernels share one runtime.
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)            # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.aoenv_abi_version() != ABI_VERSION:
        raise AoEnvError(f"libaoenv ABI {lib.aoenv_abi_version()} != binding ABI {ABI_VERSION}")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise AoEnvError(load().aoenv_last_error().decode("utf-8", "replace"))
