"""BatchedAOEnv: the gym-style surface of drl4ao's OOPAO environment, N loops per GPU, on libaoenv.

Mirrors ``MAIN/OOPAOEnv/OOPAOEnv.py`` (class ``OOPAO``): ``set_params_file / set_params / reset_soft /
step(i, action) / sample_noise / vec_to_img / img_to_vec / calculate_strehl_AVG / get_strehl`` and the
attributes and reach-through objects the trainers touch (``env.atm.generateNewPhaseScreen``,
``env.dm.coefs = 0``, ``env.tel*env.dm*env.wfs``, ``env.wfs.cam.frame`` ... -- MAIN/PO4AO/mbrl.py:49-52,
MAIN/integrator_oopao_razor.py:36-91).

All per-step physics runs in the HIP library; this module only owns PyTorch tensors for I/O and the
one-off calibration driver.  There is no CPU path: constructing an env without a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace

import numpy as np

from . import _lib as L
from . import calib

_NP_DT = {"f32": np.float32, "f64": np.float64}
# wfs.cam settings per env flavour: (before the calibration, after it) -- MAIN/OOPAOEnv/OOPAOEnv.py:379; OOPAOEnvRazor.py:243-250, 332-333
CAMERAS = {
    "ideal": ({}, {}),
    "papyrus": ({}, dict(photonNoise=True)),
    "razor": (dict(sensor="CMOS", FWC=10000, bits=10, QE=0.56, darkCurrent=5, integrationTime="samplingTime"),
              dict(photonNoise=True, readoutNoise=14)),
}


def _torch():
    import torch
    return torch


class Shard:
    """One AoEnv handle of libaoenv (a shard of independent loops on one GPU)."""

    def __init__(self, cfg: dict, device: int):
        self.lib = L.load()
        full = dict(pyr_n_res=0, pyr_n_theta=1, pyr_centering=0, pyr_norm_valid=0, pyr_q_lo=0, pyr_q_hi=0)
        full.update(cfg)
        self.cfg = L.AoCfg(abi_version=L.ABI_VERSION, **full)
        self.device = device
        self.np_dtype = np.float32 if cfg["dtype"] == L.F32 else np.float64
        h = C.c_void_p()
        L.check(self.lib.aoenv_create(C.byref(self.cfg), device, C.byref(h)))
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.aoenv_destroy(self.h)
            self.h = None

    __del__ = close

    def upload(self, kind: int, arr: np.ndarray, layer=None):
        """``layer``: the ring tables (C_AB, C_INNER_IDX, C_OUTER_IDX) of that layer alone (layers on grids of their own)."""
        want = {L.C_PUPIL: np.uint8, L.C_INNER_IDX: np.int32, L.C_OUTER_IDX: np.int32, L.C_ACT_IDX: np.int32,
                L.C_SH_SUBAP_IDX: np.int32}.get(kind, np.float64)
        a = np.ascontiguousarray(arr, dtype=want)
        if layer is None:
            L.check(self.lib.aoenv_upload(self.h, kind, a.ctypes.data_as(C.c_void_p), a.nbytes))
        else:
            L.check(self.lib.aoenv_upload_layer(self.h, kind, int(layer), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def upload_ring_tables(self, at, only_ab=False):
        """[A | B] and the ring index tables of every layer (calib.AtmosphereTables): one set when all layers share a grid."""
        if at.uniform:
            self.upload(L.C_AB, at.AB)
            if not only_ab:
                self.upload(L.C_INNER_IDX, at.inner_idx)
                self.upload(L.C_OUTER_IDX, at.outer_idx)
            return
        for l, t in enumerate(at.layers):
            self.upload(L.C_AB, t.AB, layer=l)
            if not only_ab:
                self.upload(L.C_INNER_IDX, t.inner_idx, layer=l)
                self.upload(L.C_OUTER_IDX, t.outer_idx, layer=l)

    def set_wind(self, ratio: np.ndarray, reset_buff: bool):
        r = np.ascontiguousarray(ratio, dtype=np.float64)
        L.check(self.lib.aoenv_set_wind(self.h, r.ctypes.data_as(C.c_void_p), int(reset_buff)))

    def new_screens(self, screens, ring_seeds, stream=0):
        s = None if screens is None else np.ascontiguousarray(screens, dtype=np.float64)
        k = np.ascontiguousarray(ring_seeds, dtype=np.uint32)
        L.check(self.lib.aoenv_new_screens(self.h, None if s is None else s.ctypes.data_as(C.c_void_p),
                                           k.ctypes.data_as(C.c_void_p), C.c_void_p(stream)))

    def new_screens_device(self, screen_seeds, ring_seeds, r0, L0, pixel_size, stream=0):
        a = np.ascontiguousarray(screen_seeds, dtype=np.uint32)
        k = np.ascontiguousarray(ring_seeds, dtype=np.uint32)
        L.check(self.lib.aoenv_new_screens_device(self.h, a.ctypes.data_as(C.c_void_p), k.ctypes.data_as(C.c_void_p),
                                                  float(r0), float(L0), float(pixel_size), C.c_void_p(stream)))

    def set_atm_opd(self, opd, stream=0):
        a = None if opd is None else np.ascontiguousarray(opd, dtype=np.float64)
        L.check(self.lib.aoenv_set_atm_opd(self.h, None if a is None else a.ctypes.data_as(C.c_void_p), C.c_void_p(stream)))

    def set_coefs(self, coefs, stream=0):
        a = None if coefs is None else np.ascontiguousarray(coefs, dtype=np.float64)
        L.check(self.lib.aoenv_set_coefs(self.h, None if a is None else a.ctypes.data_as(C.c_void_p), C.c_void_p(stream)))

    def measure(self, stream=0):
        L.check(self.lib.aoenv_measure(self.h, C.c_void_p(stream)))

    def atm_update(self, stream=0):
        L.check(self.lib.aoenv_atm_update(self.h, C.c_void_p(stream)))

    def download(self, which: int, shape, stream=0, dtype=None) -> np.ndarray:
        out = np.empty(shape, dtype=self.np_dtype if dtype is None else dtype)
        L.check(self.lib.aoenv_download(self.h, which, out.ctypes.data_as(C.c_void_p), out.nbytes, C.c_void_p(stream)))
        return out

    def upload_state(self, which: int, arr, stream=0, dtype=None):
        a = np.ascontiguousarray(arr, dtype=self.np_dtype if dtype is None else dtype)
        L.check(self.lib.aoenv_upload_state(self.h, which, a.ctypes.data_as(C.c_void_p), a.nbytes, C.c_void_p(stream)))

    def profile(self, enable: bool):
        L.check(self.lib.aoenv_profile(self.h, int(enable)))

    def profile_read(self, stream=0) -> dict:
        """{kernel name: (total_ms, launches)} recorded since profile(True)."""
        n = len(L.KERNEL_NAMES)
        ms = np.zeros(n)
        cnt = np.zeros(n, dtype=np.int32)
        L.check(self.lib.aoenv_profile_read(self.h, ms.ctypes.data_as(C.c_void_p), cnt.ctypes.data_as(C.c_void_p),
                                            C.c_void_p(stream)))
        return {k: (float(ms[i]), int(cnt[i])) for i, k in enumerate(L.KERNEL_NAMES)}

    def get_buff(self, n_layer: int) -> np.ndarray:
        out = np.zeros((max(n_layer, 1), 2))
        L.check(self.lib.aoenv_get_buff(self.h, out.ctypes.data_as(C.c_void_p)))
        return out[:n_layer]

    def set_buff(self, buff):
        b = np.ascontiguousarray(buff, dtype=np.float64)
        L.check(self.lib.aoenv_set_buff(self.h, b.ctypes.data_as(C.c_void_p)))

    # per-env clocks: ratio [n_layer][n_env][2] pixels per frame; clock [n_layer][n_env][4] = (ratio x, y, buff x, y)
    def set_wind_env(self, ratio: np.ndarray, reset_buff: bool, stream=0):
        r = np.ascontiguousarray(ratio, dtype=np.float64)
        L.check(self.lib.aoenv_set_wind_env(self.h, r.ctypes.data_as(C.c_void_p), int(reset_buff), C.c_void_p(stream)))

    def get_clock_env(self, n_layer: int, n_env: int) -> np.ndarray:
        out = np.zeros((n_layer, n_env, 4))
        L.check(self.lib.aoenv_get_clock_env(self.h, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_clock_env(self, clock):
        c = np.ascontiguousarray(clock, dtype=np.float64)
        L.check(self.lib.aoenv_set_clock_env(self.h, c.ctypes.data_as(C.c_void_p)))


# ----------------------------------------------------------------------------------------------------
# reach-through proxies (only the uses listed in SURVEY.md 8b)
# ----------------------------------------------------------------------------------------------------
class _AtmProxy:
    tag = "atmosphere"

    def __init__(self, env):
        self._e = env

    @property
    def windSpeed(self):
        return list(self._e.param.windSpeed)

    @windSpeed.setter
    def windSpeed(self, val):
        """OOPAO/Atmosphere.py:829-847: new ratio, the sub-pixel accumulator is kept.  A 2-D value [n_envs, nLayer] gives every
        env its own wind speed (the shard switches to per-env clocks, see set_wind_per_env)."""
        e = self._e
        if np.ndim(val) == 2:
            e.set_wind_per_env(speed=val)
            return
        if len(val) != e.param.nLayer:
            print("Error! Wrong value for the wind-speed! Make sure that you inpute a wind-speed for each layer")
            return
        e.param.windSpeed = [float(v) for v in val]
        e._push_wind(reset=False)

    @property
    def windDirection(self):
        return list(self._e.param.windDirection)

    @windDirection.setter
    def windDirection(self, val):
        e = self._e
        if np.ndim(val) == 2:
            e.set_wind_per_env(direction=val)
            return
        if len(val) != e.param.nLayer:
            print("Error! Wrong value for the wind-speed! Make sure that you inpute a wind-speed for each layer")
            return
        e.param.windDirection = [float(v) for v in val]
        e._push_wind(reset=False)

    @property
    def r0(self):
        return self._e.param.r0

    @r0.setter
    def r0(self, val):
        """OOPAO/Atmosphere.py:792-807: rescales the covariances; only B changes (A is r0-invariant)."""
        e = self._e
        e.param.r0 = float(val)
        e._atm_tables.set_r0(e.param.r0)
        e._shard.upload_ring_tables(e._atm_tables, only_ab=True)

    @property
    def nLayer(self):
        return self._e.param.nLayer

    def generateNewPhaseScreen(self, seed=None):
        self._e.generate_new_phase_screen(seed)

    def update(self):
        """atm.update() (OOPAO/Atmosphere.py:439-477): every layer of every env one frame on; atm.OPD_no_pupil follows."""
        self._e._shard.atm_update(self._e._stream())

    @property
    def OPD_no_pupil(self):
        return self._e._fetch(L.B_OPD_ATM, (self._e.R, self._e.R))

    @property
    def OPD(self):
        return self.OPD_no_pupil * self._e.pupil


class _DmProxy:
    tag = "deformableMirror"

    def __init__(self, env):
        self._e = env

    @property
    def nValidAct(self):
        return self._e.nValidAct

    @property
    def coefs(self):
        return self._e._fetch(L.B_COEFS, (self._e.nValidAct,))

    @coefs.setter
    def coefs(self, val):
        e = self._e
        if np.isscalar(val):
            if val != 0:
                print("Error: wrong value for the coefficients")
                return
            e._shard.set_coefs(None, e._stream())
            return
        v = np.asarray(val, dtype=np.float64)
        if v.shape == (e.nValidAct,):
            v = np.broadcast_to(v, (e.n_envs, e.nValidAct))
        if v.shape != (e.n_envs, e.nValidAct):
            raise ValueError(f"coefs must have shape ({e.nValidAct},) or ({e.n_envs}, {e.nValidAct})")
        e._shard.set_coefs(v, e._stream())

    @property
    def modes(self):
        return self._e._dm_tables.dense_modes()


class _Cam:
    """``env.wfs.cam``: the WFS detector settings of OOPAO/Detector.py (``photonNoise, readoutNoise, QE, darkCurrent,
    integrationTime, FWC, bits, gain, sensor``).  Assigning one pushes the whole camera model to the device
    (``aoenv_set_detector``); the frame of every later measurement goes through it before the slopes are computed."""
    _FIELDS = dict(photonNoise=False, readoutNoise=0, QE=1, darkCurrent=0, integrationTime=None, FWC=None, bits=None, gain=1,
                   sensor="CCD")

    def __init__(self, env):
        object.__setattr__(self, "_e", env)
        for k, v in self._FIELDS.items():
            object.__setattr__(self, k, v)

    def __setattr__(self, name, value):
        if name == "sensor" and value not in ("EMCCD", "CCD", "CMOS"):
            raise ValueError("Sensor must be 'EMCCD', 'CCD', or 'CMOS'")          # OOPAO/Detector.py:40-41
        object.__setattr__(self, name, value)
        if name in self._FIELDS:
            self._e._push_detector()

    def configure(self, **fields):
        """Several camera fields at once, pushed to the device together (setting them one by one pushes every intermediate
        combination, and e.g. ``bits`` without ``FWC`` is not a camera the library builds)."""
        for k, v in fields.items():
            if k not in self._FIELDS:
                raise AttributeError(f"unknown camera field {k!r}")
            if k == "sensor" and v not in ("EMCCD", "CCD", "CMOS"):
                raise ValueError("Sensor must be 'EMCCD', 'CCD', or 'CMOS'")
            object.__setattr__(self, k, v)
        self._e._push_detector()

    @property
    def frame(self):
        return self._e._fetch(L.B_FRAME, (self._e.cam_res, self._e.cam_res))


class _WfsProxy:
    def __init__(self, env):
        self._e = env
        self.tag = "shackHartmann"
        self.cam = _Cam(env)

    @property
    def nSignal(self):
        return self._e.nSignal

    @property
    def signal(self):
        return self._e._fetch(L.B_SIGNAL, (self._e.nSignal,))


class _TelProxy:
    """``env.tel*env.dm*env.wfs`` (MAIN/PO4AO/mbrl.py:52): DM propagation then one WFS measurement."""
    tag = "telescope"

    def __init__(self, env):
        self._e = env
        self.PSF = None

    @property
    def resolution(self):
        return self._e.R

    @property
    def D(self):
        return self._e.param.diameter

    @property
    def pupil(self):
        return self._e.pupil.astype(int)

    @property
    def samplingTime(self):
        return self._e.param.samplingTime

    def __mul__(self, obj):
        if getattr(obj, "tag", None) == "deformableMirror":
            return self
        if getattr(obj, "tag", None) in ("shackHartmann", "pyramid"):
            self._e.measure()
            return self
        raise AttributeError("the telescope can be multiplied only with the DM and the WFS of this env")

    def resetOPD(self):
        """Flat wave-front (OOPAO/Telescope.py:566-579): ``env.tel.resetOPD(); env.tel.computePSF(4)`` gives the diffraction-limited
        PSF (MAIN/integrator_network.py:61-64).  The env stays paired to its atmosphere: the next measurement / step re-derives the
        residual phase from the screens and the DM."""
        e = self._e
        e._shard.upload_state(L.B_PHASE, np.zeros((e.n_envs, e.R * e.R)), e._stream())

    @property
    def OPD(self):
        e = self._e
        return e._fetch(L.B_PHASE, (e.R, e.R)) * (e.src_wavelength / (2 * np.pi))

    def computePSF(self, zeroPaddingFactor=2):
        """tel.computePSF (OOPAO/Telescope.py:258-357) of the current residual phase, on the device: sets ``tel.PSF``
        ([n_envs, M, M] tensor, M = zeroPaddingFactor * resolution; a NumPy array for the single-env flavour) and
        ``tel.PSF_norma``."""
        e = self._e
        torch = _torch()
        M = int(zeroPaddingFactor) * e.R
        psf = torch.empty((e.n_envs, M, M), device=e.device, dtype=e.tdtype)
        L.check(e._shard.lib.aoenv_compute_psf(e._shard.h, int(zeroPaddingFactor), C.c_void_p(psf.data_ptr()), C.c_void_p(e._stream())))
        if e.output == "numpy":
            self.PSF = psf[0].double().cpu().numpy()
            self.PSF_norma = self.PSF / self.PSF.max()
        else:
            self.PSF = psf
            self.PSF_norma = psf / psf.amax(dim=(1, 2), keepdim=True)
        return self.PSF


# ----------------------------------------------------------------------------------------------------
class BatchedAOEnv:
    """N independent closed AO loops on one GPU behind drl4ao's ``OOPAO`` env surface.

    ``output='torch'`` (default): device tensors with a leading N dimension.
    ``output='numpy'`` with ``n_envs == 1``: the reference's exact return types, so the stock
    ``TorchWrapper`` / trainers run unchanged.
    ``return_frame``: True (default) -- ``step`` returns a new tensor with the WFS frames (one device copy per step, as the
    reference hands out a new array); "view" -- a tensor aliasing the library's frame buffer (no copy, overwritten by the next
    measurement); False -- None (the PO4AO trainer never looks at the frame, MAIN/PO4AO/mbrl.py:64-89).
    """

    metadata = {"render.modes": ["rgb_array"]}

    def __init__(self, n_envs: int = 1, device=None, dtype: str = "f32", output: str = "torch",
                 return_frame=True, env_seed_stride: int = 1, env_index_offset: int = 0):
        if dtype not in _NP_DT:
            raise ValueError("dtype must be 'f32' or 'f64'")
        if output not in ("torch", "numpy"):
            raise ValueError("output must be 'torch' or 'numpy'")
        if output == "numpy" and n_envs != 1:
            raise ValueError("output='numpy' reproduces the single-env reference interface: n_envs must be 1")
        L.load()                                  # no CPU fallback: fail here if the HIP library is missing
        torch = _torch()
        if not torch.cuda.is_available():
            raise L.AoEnvError("BatchedAOEnv needs a ROCm GPU: the step path exists only as HIP kernels")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        if isinstance(device, str):
            device = torch.device(device).index or 0
        if isinstance(device, torch.device):
            device = device.index or 0
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self.n_envs = int(n_envs)
        self.dtype = dtype
        self.tdtype = torch.float32 if dtype == "f32" else torch.float64
        self.output = output
        self.return_frame = return_frame
        self.env_seed_stride = int(env_seed_stride)
        self.env_index_offset = int(env_index_offset)
        self.detector_seed = 0
        # attributes of the reference env (MAIN/OOPAOEnv/OOPAOEnv.py:19-72)
        self.gainCL = None
        self.net_gain = 0.5
        self.leak = 0.99
        self.delay = 1
        self.F = 1
        self.reconstructor = None
        self.nActuator = None
        self.xvalid = self.yvalid = None
        self.dm_mask = None
        self.SR = []
        self.LE_PSF = None
        self.name = "OOPAO"
        self.param_file = ""
        self.oopao_path = ""
        self.action_buffer = []
        self.atm = self.dm = self.tel = self.wfs = None
        self._shard = None
        self._done = None
        self._wind_env = None                                      # (speed, direction) [n_envs, nLayer] once per-env winds are set
        self._per_env_clock = False

    # -- construction --------------------------------------------------------------------------------
    def set_params_file(self, param_file, oopao_path):
        """Kept for call compatibility (MAIN/PO4AO/mbrl.py:27); the parameters come from ``set_params``."""
        self.param_file, self.oopao_path = param_file, oopao_path

    def set_params(self, args=None, wfs_type="pyramid", modal_basis="zernike", gainCL=0.5, m2c=None, second_dm=None,
                   camera="papyrus", atm_AB=None, **kw):
        """Builds the loop (MAIN/OOPAOEnv/OOPAOEnv.py:93-385).  ``wfs_type`` is "pyramid" (the reference's default,
        Papyrus) or "shackhartmann" (OOPAOEnvRazor.py:232-238).  ``second_dm=dict(nSubaperture=n)`` chains a second DM of
        that pitch behind the first (``tel*dm1*dm2*wfs``, BASELINE configs[4]): commands, observations and actions then
        cover both mirrors (``calib.CompositeDM``: one block-diagonal actuator image, separable like a single mirror).
        ``camera``: the WFS detector the env ends set_params with -- "papyrus" (default): ``wfs.cam.photonNoise = True``
        (OOPAOEnv.py:379); "razor": the Razor env's CMOS camera (QE 0.56, FWC 1e4, 10-bit ADC, dark current 5 e-/s set before the
        calibration, photon noise and 14 e- read-out noise after it, OOPAOEnvRazor.py:243-250, 332-333); "ideal": no noise (the
        parity configuration: the reference's noisy frames are wall-clock seeded, Detector.py:127-130).  The calibration itself always
        sees ideal spot intensities, as in the reference: the WFS constructor measures its reference slopes before the camera is
        configured, and the interaction-matrix pokes go through the Shack-Hartmann's multi-wave-front branch, which computes the
        centroids from the intensities without passing them through the detector (OOPAO/ShackHartmann.py:605-672)."""
        if camera not in CAMERAS:
            raise ValueError(f"camera must be one of {sorted(CAMERAS)}")
        self.camera = camera
        if wfs_type in ("shackhartmann", "sh"):
            self.wfs_type = "sh"
        elif wfs_type in ("pyramid", "pyr"):
            self.wfs_type = "pyr"
        else:
            raise ValueError(f"unknown wfs_type {wfs_type!r}")
        torch = _torch()
        self.gainCL = gainCL
        p = self.param = calib.params_from_args(args, **kw)
        self.leak = p.leak
        self.R = p.resolution
        self.pupil = calib.telescope_pupil(self.R, p.centralObstruction)
        self.src_wavelength, self.nPhoton = calib.source(p.opticalBand, p.magnitude)
        self._atm_tables = calib.AtmosphereTables(p)
        if atm_AB is not None:
            # the ring-extrusion operators handed over instead of recomputed: A = ZXt^T pinv(ZZt) goes through the pseudo-inverse
            # of a covariance matrix of condition ~1e9, whose result differs between CPUs / LAPACK builds at the 1e-9 .. 1e-7 level
            # (the parity tests inject the recorded operators to separate that from the device arithmetic)
            A_, B_ = (np.asarray(x, dtype=np.float64) for x in atm_AB)
            at = self._atm_tables
            if not at.uniform:
                raise ValueError("atm_AB: the layers of this atmosphere have grids (and operators) of their own")
            if A_.shape != at.A.shape or B_.shape != at.B.shape:
                raise ValueError(f"atm_AB must have shapes {at.A.shape} and {at.B.shape}")
            at.A, at.B = A_, B_
            at.AB = np.ascontiguousarray(np.concatenate([A_, B_], axis=1))
            at.layers[0].A, at.layers[0].B, at.layers[0].AB = at.A, at.B, at.AB
        self._dm_tables = dmt = (calib.DMTables(p) if not second_dm else
                                 calib.CompositeDM(p, int(second_dm["nSubaperture"])))
        self._dm_separable = 1 if dmt.gx is not None else 0
        self.nActuator, self.nValidAct = dmt.nAct, dmt.nValidAct
        self.dm_mask = dmt.dm_mask.astype(int)
        self.xvalid, self.yvalid = dmt.xvalid, dmt.yvalid
        if self.wfs_type == "sh":
            self._sh_tables = sht = calib.SHTables(p, self.pupil, self.nPhoton)
            self.nSignal, self.cam_res = sht.nSignal, sht.cam_res
            self._wfs_valid_idx, self._wfs_n_theta, self._pyr_tt = sht.subap_idx, 1, None
        else:
            self._pyr_tables = calib.PyramidTables(p, self.pupil, self.nPhoton, psf_centering=p.psfCentering,
                                                   n_pix_separation=p.n_pix_separation, post_processing=p.postProcessing)
            self.cam_res = self._pyr_tables.cam_res
        self._xv_t = torch.as_tensor(self.xvalid, device=self.device)
        self._yv_t = torch.as_tensor(self.yvalid, device=self.device)

        # -- calibration on the GPU, float64, same kernels as the loop ------------------------------
        ref, units = self._calibrate_wfs() if self.wfs_type == "sh" else self._calibrate_pyramid()
        self.reference_centroids, self.slopes_units = ref, units
        self.imat = self._interaction_matrix(ref, units)
        if m2c is None:
            m2c = calib.zernike_m2c(dmt, self.pupil, p.diameter, p.nModes)
        elif isinstance(m2c, (str, os.PathLike)):
            m2c = np.load(m2c)
        self.M2C_CL = np.asarray(m2c, dtype=np.float64)[:, :p.nModes]
        if self.M2C_CL.shape[0] != self.nValidAct:
            raise ValueError(f"M2C has {self.M2C_CL.shape[0]} rows, the DM has {self.nValidAct} valid actuators")
        self.reconstructor, self.F, self.modal_CM = calib.reconstructor_from_imat(self.imat, self.M2C_CL, True)
        self._F_t = torch.as_tensor(self.F, device=self.device, dtype=self.tdtype)

        # -- the loop shard -----------------------------------------------------------------------------
        self._shard = self._make_shard(self.n_envs, self.dtype, n_layer=p.nLayer, max_group=1)
        sh = self._shard
        at = self._atm_tables
        sh.upload_ring_tables(at)
        sh.upload(L.C_LAYER_WEIGHT, at.weights)
        sh.upload(L.C_SH_REF, ref)
        sh.upload(L.C_WFS_UNITS, np.array([units]))
        sh.upload(L.C_RECON, self.reconstructor)
        sh.upload(L.C_RECON_FACTORS, np.concatenate([self.modal_CM.reshape(-1), self.M2C_CL.reshape(-1)]))
        self._push_wind(reset=True)
        N, A_ = self.n_envs, self.nActuator
        self._obs = torch.zeros((N, A_, A_), device=self.device, dtype=self.tdtype)
        self._reward = torch.zeros((N,), device=self.device, dtype=self.tdtype)
        self._strehl = torch.zeros((N,), device=self.device, dtype=self.tdtype)
        self._frame = torch.zeros((N, self.cam_res, self.cam_res), device=self.device, dtype=self.tdtype) \
            if self.return_frame else None
        self.SR = []
        self.atm, self.dm, self.tel, self.wfs = _AtmProxy(self), _DmProxy(self), _TelProxy(self), _WfsProxy(self)
        self.wfs.tag = "shackHartmann" if self.wfs_type == "sh" else "pyramid"
        pre, post = CAMERAS[camera]
        self.wfs.cam.configure(**{k: (p.samplingTime if v == "samplingTime" else v) for k, v in pre.items()})
        # flat measurement, then the initial screens (MAIN/OOPAOEnv/OOPAOEnv.py:312-322)
        self.measure()
        self.generate_new_phase_screen(10)
        self.wfs.cam.configure(**post)                              # OOPAOEnv.py:379 / OOPAOEnvRazor.py:332-333
        return self

    def _make_shard(self, n_env, dtype, n_layer, max_group) -> Shard:
        p, at, dmt = self.param, self._atm_tables, self._dm_tables
        valid_idx = self._wfs_valid_idx
        cfg = dict(dtype=L.F32 if dtype == "f32" else L.F64, n_env=n_env, resolution=self.R, n_layer=n_layer,
                   layer_res=at.N, n_inner=at.n_inner, n_outer=at.n_outer, n_act=dmt.nAct, n_valid_act=dmt.nValidAct,
                   dm_separable=self._dm_separable, n_subap=p.nSubaperture, n_valid_subap=len(valid_idx), n_signal=2 * len(valid_idx),
                   cam_res=self.cam_res, n_loop=int(p.nLoop), max_group=max_group,
                   atm_wavelength=calib.ATM_WAVELENGTH, src_wavelength=self.src_wavelength, leak=p.leak,
                   threshold_cog=p.threshold_cog)
        if self.wfs_type == "sh":
            cfg.update(wfs_type=L.WFS_SH)
        else:
            pt = self._pyr_tables
            cfg.update(wfs_type=L.WFS_PYRAMID, pyr_n_res=pt.nRes, pyr_n_theta=self._wfs_n_theta,
                       pyr_centering=int(pt.psf_centering), pyr_norm_valid=pt.norm_valid, pyr_q_lo=pt.q_lo, pyr_q_hi=pt.q_hi)
        if n_layer > 0 and not at.uniform:                          # fov != 0: a layer above the ground has a grid of its own
            cfg["layer_res_l"] = (C.c_int32 * 8)(*(at.layer_res + [0] * (8 - len(at.layer_res))))
        sh = Shard(cfg, self.device_index)
        sh.upload(L.C_PUPIL, self.pupil.astype(np.uint8))
        if self._dm_separable:
            sh.upload(L.C_DM_GX, dmt.gx)
            sh.upload(L.C_DM_GY, dmt.gy)
        else:
            sh.upload(L.C_DM_MODES, dmt.dense_modes())
        sh.upload(L.C_ACT_IDX, dmt.act_idx)
        sh.upload(L.C_SH_SUBAP_IDX, valid_idx)
        if self.wfs_type == "sh":
            sh.upload(L.C_WFS_AMP, self._sh_tables.amp)
        else:
            sh.upload(L.C_WFS_AMP, self._pyr_tables.amplitude(self._wfs_n_theta))
            sh.upload(L.C_PYR_MASK, self._pyr_tables.mask_pairs)
            if self._wfs_n_theta > 1:
                sh.upload(L.C_PYR_TT, self._pyr_tt)
        return sh

    def _calibrate_pyramid(self):
        """Pyramid initialisation (OOPAO/Pyramid.py:306-314, 408-466): valid pixels from the flux at the (large)
        calibration modulation, then the reference slopes of a flat wave-front at the user modulation.  Both are
        measured by the HIP kernels in float64.  Returns (reference at the valid pixels, slopesUnits = 1)."""
        p, pt, R = self.param, self._pyr_tables, self.R
        ns = p.nSubaperture
        # 1. flux at calibModulation, all quadrant pixels provisionally valid
        self._wfs_valid_idx = np.arange(ns * ns, dtype=np.int32)
        self._wfs_n_theta, self._pyr_tt = pt.modulation_table(pt.calib_modulation)
        cal = self._make_shard(1, "f64", n_layer=0, max_group=1)
        try:
            cal.upload(L.C_SH_REF, np.zeros(2 * ns * ns))
            cal.upload(L.C_WFS_UNITS, np.array([1.0]))
            cal.measure()
            frame = cal.download(L.B_FRAME, (1, self.cam_res, self.cam_res))[0].astype(np.float64)
        finally:
            cal.close()
        i4q = pt.quadrant_sum(frame)
        light = 0.1 if p.lightThreshold is None else p.lightThreshold
        self.validI4Q = i4q >= light * i4q.max()
        self._wfs_valid_idx = np.flatnonzero(self.validI4Q.reshape(-1)).astype(np.int32)
        self.nSignal = 2 * len(self._wfs_valid_idx)
        # 2. reference slopes: OPD = 1 m of piston inside the pupil (wfs_calibration), user modulation
        self._wfs_n_theta, self._pyr_tt = pt.modulation_table(p.modulation)
        cal = self._make_shard(1, "f64", n_layer=0, max_group=1)
        try:
            cal.upload(L.C_SH_REF, np.zeros(self.nSignal))
            cal.upload(L.C_WFS_UNITS, np.array([1.0]))
            cal.set_atm_opd(np.ones((1, R * R)))
            cal.measure()
            ref = cal.download(L.B_SIGNAL, (1, self.nSignal))[0].astype(np.float64)
        finally:
            cal.close()
        return ref, 1.0

    def _calibrate_wfs(self):
        """initialize_wfs (OOPAO/ShackHartmann.py:254-312): reference centroids from a flat wave-front,
        slope units from a five-point tip ramp, measured by the HIP kernels in float64."""
        R, nv = self.R, self._sh_tables.nValid
        self.nSignal = 2 * nv
        cal = self._make_shard(5, "f64", n_layer=0, max_group=1)
        try:
            cal.upload(L.C_SH_REF, np.zeros(2 * nv))
            cal.upload(L.C_WFS_UNITS, np.array([1.0]))
            cal.measure()
            ref = cal.download(L.B_SIGNAL, (5, 2 * nv))[0].astype(np.float64)
            cal.upload(L.C_SH_REF, ref)
            tip = calib.tip_ramp(R)
            amp = 10e-9
            cal.set_atm_opd(np.stack([tip * (i - 2) * amp for i in range(5)]).reshape(5, R * R))
            cal.measure()
            sig = cal.download(L.B_SIGNAL, (5, 2 * nv))
            mean_slope = sig[:, :nv].mean(axis=1)
            fit = np.polyfit(np.linspace(-2, 2, 5) * amp, mean_slope, deg=1)
            units = float(np.abs(fit[0]) * (self.src_wavelength / 2 / np.pi))
        finally:
            cal.close()
        return ref, units

    def _interaction_matrix(self, ref, units):
        """Zonal push-only interaction matrix (OOPAO/calibration/InteractionMatrix.py:13-135 with
        single_pass=True, stroke = lambda/16, MAIN/OOPAOEnv/OOPAOEnv.py:270-288): every actuator is one
        "env" of a float64 calibration shard; consecutive groups of nMeasurements pokes share the
        centroid threshold as the reference's batched measurement does."""
        A_ = self.nValidAct
        stroke = self.src_wavelength / 16
        n_meas = int(self.param.nMeasurements)
        # pokes per calibration shard: whole measurement groups, bounded so that the Pyramid's nRes^2 scratch fits
        batch = A_ if self.wfs_type == "sh" else max(n_meas, (96 // n_meas) * n_meas)
        sig = np.zeros((A_, self.nSignal))
        # every DM is poked by its own InteractionMatrix call (an un-paired telescope keeps only the last DM's OPD,
        # OOPAO/DeformableMirror.py:474-476): the measurement groups do not straddle two mirrors
        dms = [self._dm_tables.dm1.nValidAct, self._dm_tables.dm2.nValidAct] if hasattr(self._dm_tables, "dm2") else [A_]
        spans, lo = [], 0
        for n_dm in dms:
            spans += [(a0, min(batch, lo + n_dm - a0)) for a0 in range(lo, lo + n_dm, batch)]
            lo += n_dm
        for a0, n in spans:
            cal = self._make_shard(n, "f64", n_layer=0, max_group=n_meas)
            try:
                cal.upload(L.C_SH_REF, ref)
                cal.upload(L.C_WFS_UNITS, np.array([units]))
                coefs = np.zeros((n, A_))
                coefs[np.arange(n), a0 + np.arange(n)] = stroke
                cal.set_coefs(coefs)
                cal.measure()
                sig[a0:a0 + n] = cal.download(L.B_SIGNAL, (n, self.nSignal)).astype(np.float64)
            finally:
                cal.close()
        return (sig / stroke).T                                     # [nSignal, A]

    # -- plumbing ---------------------------------------------------------------------------------------
    def _stream(self) -> int:
        return int(_torch().cuda.current_stream(self.device).cuda_stream)

    def _fetch(self, which, shape):
        out = self._shard.download(which, (self.n_envs,) + tuple(shape), self._stream())
        return out[0] if self.n_envs == 1 else out

    def _push_detector(self):
        """wfs.cam.* -> aoenv_set_detector (OOPAO/Detector.py:232-301).  ``detector_seed`` keys the noise streams."""
        if self._shard is None or self.wfs is None:
            return
        cam = self.wfs.cam
        t_int = cam.integrationTime if cam.integrationTime is not None else self.param.samplingTime
        d = L.AoDetector(photon_noise=int(bool(cam.photonNoise)), bits=int(cam.bits or 0), emccd=int(cam.sensor == "EMCCD"),
                         env_index_offset=int(self.env_index_offset), qe=float(cam.QE),
                         dark_electrons=float(cam.darkCurrent) * float(t_int), fwc=float(cam.FWC or 0), gain=float(cam.gain),
                         readout_noise=float(cam.readoutNoise), seed=int(self.detector_seed) & 0xFFFFFFFFFFFFFFFF)
        L.check(self._shard.lib.aoenv_set_detector(self._shard.h, C.byref(d), C.c_void_p(self._stream())))

    def _push_wind(self, reset: bool):
        p = self.param
        self._wind_env = None                                      # one wind for the shard again (per-env clocks stay per-env)
        self._shard.set_wind(self._atm_tables.wind_ratio(p.windSpeed, p.windDirection, p.samplingTime), reset)

    def set_wind_per_env(self, speed=None, direction=None, reset: bool = False):
        """Every env its own wind: ``speed`` [m/s] and ``direction`` [deg] of shape [n_envs, nLayer] (one of them may be None:
        the shard's current value).  What a trainer does that draws the wind per run (MAIN/integrator_oopao_razor.py:41-44,
        OOPAO/Atmosphere.py:829-873), batched: env e then evolves exactly like a shard whose shared wind is (speed[e],
        direction[e]).  At most one pixel per frame and axis.  ``atm.windSpeed = array2d`` / ``atm.windDirection = array2d`` call this."""
        p = self.param
        cur_s, cur_d = (self._wind_env if self._wind_env is not None else
                        (np.tile(np.asarray(p.windSpeed, float), (self.n_envs, 1)), np.tile(np.asarray(p.windDirection, float), (self.n_envs, 1))))
        s = cur_s if speed is None else np.asarray(speed, dtype=np.float64)
        d = cur_d if direction is None else np.asarray(direction, dtype=np.float64)
        if s.shape != (self.n_envs, p.nLayer) or d.shape != (self.n_envs, p.nLayer):
            raise ValueError(f"per-env wind: speed / direction must be [n_envs={self.n_envs}, nLayer={p.nLayer}]")
        ratio = np.zeros((p.nLayer, self.n_envs, 2))
        for e in range(self.n_envs):
            ratio[:, e] = self._atm_tables.wind_ratio(s[e], d[e], p.samplingTime)
        self._shard.set_wind_env(ratio, reset, self._stream())
        self._wind_env = (s.copy(), d.copy())
        self._per_env_clock = True

    def env_seeds(self, seed: int) -> np.ndarray:
        idx = np.arange(self.n_envs, dtype=np.int64) + self.env_index_offset
        return int(seed) + idx * self.env_seed_stride

    def generate_new_phase_screen(self, seed=None, screens=None):
        """atm.generateNewPhaseScreen(seed) for every env; env e uses ``seed + e * env_seed_stride``
        (layer l: screen seed + l, ring RandomState seed + 1000 l -- OOPAO/Atmosphere.py:574-579).
        The screens are drawn on the device (same MT19937 stream and spectrum as the reference, float64 FFT).
        ``screens`` [n_envs, nLayer, N, N] (rad @ 500 nm, N = resolution + 4): caller-made layer screens uploaded instead
        (``aoenv_new_screens``); the rings and their RandomStates are still seeded from ``seed``."""
        import time as _t
        if seed is None:
            t = _t.localtime()
            seed = t.tm_hour * 3600 + t.tm_min * 60 + t.tm_sec
        p, at = self.param, self._atm_tables
        seeds = self.env_seeds(seed)
        delta = at.layer_D / at.N
        ring = np.array([[(int(s) + 1000 * l) & 0xFFFFFFFF for l in range(p.nLayer)] for s in seeds], dtype=np.uint32)
        if screens is None:
            scr = np.array([[(int(s) + l) & 0xFFFFFFFF for l in range(p.nLayer)] for s in seeds], dtype=np.uint32)
            self._shard.new_screens_device(scr, ring, p.r0, p.L0, delta, self._stream())
        elif at.uniform:
            screens = np.asarray(screens, dtype=np.float64)
            if screens.shape != (self.n_envs, p.nLayer, at.N, at.N):
                raise ValueError(f"screens must have shape ({self.n_envs}, {p.nLayer}, {at.N}, {at.N})")
            self._shard.new_screens(screens.reshape(self.n_envs, p.nLayer, at.N * at.N), ring, self._stream())
        else:                                                       # one array [n_envs, N_l, N_l] per layer
            if len(screens) != p.nLayer or any(np.shape(x) != (self.n_envs, n, n) for x, n in zip(screens, at.layer_res)):
                raise ValueError(f"screens must be a list of per-layer arrays [{self.n_envs}, N_l, N_l] with N_l = {at.layer_res}")
            flat = np.concatenate([np.asarray(x, dtype=np.float64).reshape(-1) for x in screens])
            self._shard.new_screens(flat, ring, self._stream())
        if self._wind_env is not None:
            self.set_wind_per_env(reset=True)                       # every env keeps its own wind over the episodes
        else:
            self._push_wind(reset=True)

    def measure(self):
        """tel*dm*wfs: one WFS measurement of (atmosphere + DM), no turbulence update."""
        self._shard.measure(self._stream())

    # -- the reference surface ------------------------------------------------------------------------------
    def _out(self, t):
        if self.output == "numpy":
            return t.detach().to("cpu", dtype=_torch().float64).numpy()[0]
        return t

    def reset_soft(self):
        """MAIN/OOPAOEnv/OOPAOEnv.py:82-86."""
        self.action_buffer = []
        # a NEW tensor, like step(): the observation handed out by the previous step (a trainer may have kept it) is not written to
        obs = _torch().empty_like(self._obs)
        L.check(self._shard.lib.aoenv_reset_soft(self._shard.h, C.c_void_p(obs.data_ptr()), C.c_void_p(self._stream())))
        self._obs = obs
        return self._out(obs)

    def reset(self):
        raise NotImplementedError("reset() re-runs set_params() with no arguments in the reference and fails there "
                                  "(MAIN/OOPAOEnv/OOPAOEnv.py:76 vs :93); use set_params() + reset_soft()")

    def _action_tensor(self, action):
        torch = _torch()
        a = torch.as_tensor(action)
        if a.dim() == 2:
            a = a.unsqueeze(0)
        if tuple(a.shape) != (self.n_envs, self.nActuator, self.nActuator):
            raise ValueError(f"action must have shape ({self.n_envs}, {self.nActuator}, {self.nActuator}), got {tuple(a.shape)}")
        return a.to(device=self.device, dtype=self.tdtype).contiguous()

    def step(self, i, action):
        """MAIN/OOPAOEnv/OOPAOEnv.py:485-536.  Returns (obs, wfs_frame, reward, strehl, done, info).  Every call hands out NEW
        tensors, as the reference hands out new arrays (a replay buffer may keep them): the library writes straight into them, so
        there is no copy kernel behind the call -- only the allocator."""
        torch = _torch()
        a = self._action_tensor(action)
        N, A_ = self.n_envs, self.nActuator
        obs = torch.empty((N, A_, A_), device=self.device, dtype=self.tdtype)
        reward = torch.empty((N,), device=self.device, dtype=self.tdtype)
        strehl = torch.empty((N,), device=self.device, dtype=self.tdtype)
        view = self.return_frame == "view"
        fr = torch.empty((N, self.cam_res, self.cam_res), device=self.device, dtype=self.tdtype) if (self.return_frame and not view) else None
        L.check(self._shard.lib.aoenv_step(
            self._shard.h, int(i), C.c_void_p(a.data_ptr()), C.c_void_p(obs.data_ptr()),
            C.c_void_p(fr.data_ptr()) if fr is not None else None, C.c_void_p(reward.data_ptr()),
            C.c_void_p(strehl.data_ptr()), C.c_void_p(self._stream())))
        if view:
            fr = self._frame_alias()
        self._obs, self._reward, self._strehl, self._frame = obs, reward, strehl, fr
        self.SR.append(strehl)
        if self.output == "numpy":
            s = float(strehl[0])
            return (self._out(obs), None if fr is None else self._out(fr), float(reward[0]), s, False, {"strehl": s})
        if self._done is None:
            self._done = torch.zeros(N, dtype=torch.bool, device=self.device)      # never terminal (OOPAOEnv.py:531): one shared tensor
        return obs, fr, reward, strehl, self._done, {"strehl": strehl}

    @property
    def fused_step(self) -> bool:
        """True when ``step`` runs as ONE kernel per step (float32 Shack-Hartmann inside the fused kernel's envelope, see
        aoenv_fused_step_active in include/aoenv.h); False: the batched kernels (every geometry, float64, the Pyramid)."""
        return bool(self._shard.lib.aoenv_fused_step_active(self._shard.h))

    def _frame_alias(self):
        """wfs.cam.frame of every env as a tensor that ALIASES the library's buffer (no copy: 14.7 MB per step at 256 envs of the
        8 m geometry); the next measurement overwrites it.  ``return_frame="view"``."""
        if getattr(self, "_frame_view", None) is None:
            ptr, nbytes = C.c_void_p(), C.c_size_t()
            L.check(self._shard.lib.aoenv_buffer(self._shard.h, L.B_FRAME, C.byref(ptr), C.byref(nbytes)))

            class _Iface:
                pass
            o = _Iface()
            o.__cuda_array_interface__ = {"shape": (self.n_envs, self.cam_res, self.cam_res), "typestr": "<f4" if self.dtype == "f32" else "<f8",
                                          "data": (int(ptr.value), False), "version": 2, "strides": None}
            self._frame_view = _torch().as_tensor(o, device=self.device)
        return self._frame_view

    def run_integrator(self, i0: int, n_steps: int, gain=None):
        """On-device closed loop of MAIN/integrator_oopao_razor.py:66-91: ``action = gainCL * obs`` fused into
        the step epilogue; returns the last (obs, reward, strehl).  ``reset_soft()`` (or a previous step) must
        have produced the current observation."""
        g = float(self.gainCL if gain is None else gain)
        # the loop runs in place on the observation: on private copies, never on tensors step() / reset_soft() have handed out
        self._obs = self._obs.clone()
        self._reward, self._strehl = _torch().empty_like(self._reward), _torch().empty_like(self._strehl)
        L.check(self._shard.lib.aoenv_run_integrator(
            self._shard.h, int(i0), int(n_steps), g, C.c_void_p(self._obs.data_ptr()), None,
            C.c_void_p(self._reward.data_ptr()), C.c_void_p(self._strehl.data_ptr()), C.c_void_p(self._stream())))
        return self._obs, self._reward, self._strehl

    # -- checkpoint / resume (env state = screens, sub-pixel accumulators, ring RNG, dm coefs) --------
    def get_state(self) -> dict:
        """Everything the next ``step`` depends on, as host arrays: restoring it with ``set_state`` (same geometry, same
        n_envs) continues the episode bit for bit."""
        sh, p, at = self._shard, self.param, self._atm_tables
        st = self._stream()
        return {
            "screen": self._download_screens(),
            "buff": None if self._per_env_clock else sh.get_buff(p.nLayer).copy(),
            "clock_env": sh.get_clock_env(p.nLayer, self.n_envs) if self._per_env_clock else None,
            "wind_env": self._wind_env,
            "mt": sh.download(L.B_MT_STATE, (p.nLayer, self.n_envs, 625), st, dtype=np.uint32),
            "coefs": sh.download(L.B_COEFS, (self.n_envs, self.nValidAct), st),
            "dm_prev": sh.download(L.B_DM_PREV, (self.n_envs, self.nValidAct), st),
            "signal": sh.download(L.B_SIGNAL, (self.n_envs, self.nSignal), st),
            "counters": sh.download(L.B_COUNTERS, (4,), st, dtype=np.uint32),
            "obs": self._obs.detach().cpu().numpy().copy(),
            "windSpeed": list(p.windSpeed), "windDirection": list(p.windDirection),
        }

    def set_state(self, state: dict):
        sh, p = self._shard, self.param
        st = self._stream()
        p.windSpeed, p.windDirection = list(state["windSpeed"]), list(state["windDirection"])
        if state.get("clock_env") is not None:                       # per-env clocks: ratios and accumulators of every env
            clk = np.asarray(state["clock_env"])
            self._shard.set_wind_env(clk[..., :2], False, st)
            self._per_env_clock = True
            self._wind_env = state.get("wind_env")
            sh.upload_state(L.B_SCREEN, self._flat_screens(state["screen"]), st)
            sh.set_clock_env(clk)
        else:
            if self._per_env_clock:
                raise ValueError("this env runs per-env clocks; the state was saved from a shared-clock env")
            self._push_wind(reset=False)
            sh.upload_state(L.B_SCREEN, self._flat_screens(state["screen"]), st)
            sh.set_buff(state["buff"])
        sh.upload_state(L.B_MT_STATE, state["mt"], st, dtype=np.uint32)
        sh.upload_state(L.B_COEFS, state["coefs"], st)
        sh.upload_state(L.B_DM_PREV, state.get("dm_prev", state["coefs"]), st)
        sh.upload_state(L.B_SIGNAL, state["signal"], st)
        sh.upload_state(L.B_COUNTERS, state["counters"], st, dtype=np.uint32)
        self._obs = _torch().as_tensor(state["obs"]).to(device=self.device, dtype=self.tdtype).clone()   # (never into a handed-out tensor)

    def _download_screens(self):
        """layer.mapShift of every env: [nLayer, n_envs, S, S], or -- layers on grids of their own -- a list of [n_envs, S_l, S_l]."""
        sh, p, at = self._shard, self.param, self._atm_tables
        if at.uniform:
            return sh.download(L.B_SCREEN, (p.nLayer, self.n_envs, at.S, at.S), self._stream())
        sizes = [self.n_envs * t.S * t.S for t in at.layers]
        flat = sh.download(L.B_SCREEN, (sum(sizes),), self._stream())
        out, o = [], 0
        for t, n in zip(at.layers, sizes):
            out.append(flat[o:o + n].reshape(self.n_envs, t.S, t.S))
            o += n
        return out

    @staticmethod
    def _flat_screens(scr):
        return scr if isinstance(scr, np.ndarray) else np.concatenate([np.asarray(x).reshape(-1) for x in scr])

    def accumulate_returns(self, tensor):
        """Attach a device tensor [n_envs] (env dtype) to which every step adds its reward (None detaches): the
        episode return the trainers sum on the host (MAIN/PO4AO/mbrl.py:64-89), kept on the device."""
        if tensor is not None:
            if tuple(tensor.shape) != (self.n_envs,) or tensor.dtype != self.tdtype or not tensor.is_cuda or not tensor.is_contiguous():
                raise ValueError(f"the return accumulator must be a contiguous cuda {self.tdtype} tensor of shape ({self.n_envs},)")
        self._returns = tensor
        L.check(self._shard.lib.aoenv_set_return_accumulator(
            self._shard.h, None if tensor is None else C.c_void_p(tensor.data_ptr())))

    def calculate_strehl_AVG(self):
        """MAIN/OOPAOEnv/OOPAOEnv.py:538-546: (mean, std) over the episode, then clears the list."""
        torch = _torch()
        sr = torch.stack(self.SR) if len(self.SR) else torch.zeros((1, self.n_envs), device=self.device)
        avg, std = sr.mean(dim=0), sr.std(dim=0, unbiased=False)
        self.SR = []
        if self.n_envs == 1:
            return float(avg[0]), float(std[0])
        return avg, std

    @property
    def dm_prev(self):
        """The leaky integrator's state (MAIN/OOPAOEnv/OOPAOEnv.py:314, 508-509): ``step`` computes
        ``dm.coefs = dm_prev * leak + action`` and stores it back.  ``env.dm.coefs = 0`` does not clear it (the trainers' episode
        prologue leaves it as the previous episode ended, as in the reference); ``env.dm_prev = 0`` does."""
        return self._fetch(L.B_DM_PREV, (self.nValidAct,))

    @dm_prev.setter
    def dm_prev(self, val):
        v = np.zeros((self.n_envs, self.nValidAct)) if np.isscalar(val) and val == 0 else np.asarray(val, dtype=np.float64)
        if v.shape == (self.nValidAct,):
            v = np.broadcast_to(v, (self.n_envs, self.nValidAct))
        if v.shape != (self.n_envs, self.nValidAct):
            raise ValueError(f"dm_prev must be 0 or have shape ({self.nValidAct},) or ({self.n_envs}, {self.nValidAct})")
        self._shard.upload_state(L.B_DM_PREV, v, self._stream())

    def render4plot(self, current_i):
        """MAIN/OOPAOEnv/OOPAOEnv.py:473-482: short-exposure PSF of the current residual (``tel.computePSF(4)``) and the
        long-exposure one, ``LE_PSF = mean(log10(PSF))`` over the frames with ``current_i > 15`` -- a running sum on the device
        instead of the reference's growing list.  Returns (LE_PSF, log10(PSF)); leading env dimension unless n_envs == 1."""
        torch = _torch()
        self.tel.computePSF(4)
        se = torch.log10(torch.as_tensor(self.tel.PSF, device=self.device)) if self.output == "numpy" else torch.log10(self.tel.PSF)
        if current_i > 15:
            self._se_sum = se.clone() if getattr(self, "_se_sum", None) is None else self._se_sum + se
            self._se_n = getattr(self, "_se_n", 0) + 1
            self.LE_PSF = self._se_sum / self._se_n
            if self.output == "numpy":
                self.LE_PSF = self.LE_PSF.double().cpu().numpy()
        return self.LE_PSF, (se.double().cpu().numpy() if self.output == "numpy" else se)

    render = render4plot                      # the reference's render() is render4plot() plus a matplotlib window

    def get_strehl(self):
        return float(self._strehl[0]) if self.n_envs == 1 else self._strehl.clone()

    def get_slopes(self):
        return self.wfs.signal

    @property
    def total(self):
        t = self._shard.download(L.B_TOTAL, (int(self.param.nLoop), self.n_envs), self._stream())
        return t[:, 0] if self.n_envs == 1 else t

    @property
    def residual(self):
        t = self._shard.download(L.B_RESIDUAL, (int(self.param.nLoop), self.n_envs), self._stream())
        return t[:, 0] if self.n_envs == 1 else t

    def sample_noise(self, sigma, use_torch=False):
        """Exploration noise F @ (sigma N(0,1)^A) as an actuator image (MAIN/OOPAOEnv/OOPAOEnv.py:566-570)."""
        torch = _torch()
        if self.n_envs == 1:
            noise = self.F @ (sigma * np.random.normal(0, 1, size=(int(self.nValidAct),)))
            return self.vec_to_img(torch.tensor(noise).float().to(self.device), use_torch)
        z = sigma * torch.randn((self.n_envs, self.nValidAct), device=self.device, dtype=self.tdtype)
        return self.vec_to_img(z @ self._F_t.T, True)

    def vec_to_img(self, action_vec, use_torch=False):
        """MAIN/OOPAOEnv/OOPAOEnv.py:572-581, plus a leading batch dimension."""
        torch = _torch()
        if torch.is_tensor(action_vec) or use_torch:
            v = torch.as_tensor(action_vec)
            img = torch.zeros(v.shape[:-1] + (self.nActuator, self.nActuator), dtype=v.dtype if v.is_floating_point() else torch.float32,
                              device=v.device)
            img[..., self._xv_t.to(v.device), self._yv_t.to(v.device)] = v.to(img.dtype)
            return img.float() if use_torch and v.dim() == 1 else img
        v = np.asarray(action_vec)
        img = np.zeros(v.shape[:-1] + (self.nActuator, self.nActuator))
        img[..., self.xvalid, self.yvalid] = v
        return img

    def img_to_vec(self, action):
        """MAIN/OOPAOEnv/OOPAOEnv.py:583-590 (2-D, batched 3-D and the 4-D network layout)."""
        if _torch().is_tensor(action):
            return action[..., self._xv_t.to(action.device), self._yv_t.to(action.device)]
        return np.asarray(action)[..., self.xvalid, self.yvalid]

    def close(self):
        if self._shard is not None:
            self._shard.close()
            self._shard = None


class OOPAO(BatchedAOEnv):
    """Name-compatible single-env flavour: ``from rlao_amd.env import OOPAO`` in place of
    ``from OOPAOEnv.OOPAOEnv import OOPAO`` (MAIN/PO4AO/mbrl.py:15) -- NumPy/float returns for the stock wrappers."""

    def __init__(self, **kw):
        kw.setdefault("n_envs", 1)
        kw.setdefault("output", "numpy")
        super().__init__(**kw)


def namespace_from_yaml(path: str) -> SimpleNamespace:
    """read_yaml_file + SimpleNamespace of MAIN/ML_stuff/dataset_tools.py:73 (safe loader)."""
    import yaml
    with open(path) as f:
        return SimpleNamespace(**yaml.safe_load(f))
