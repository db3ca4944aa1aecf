"""Host-side (NumPy float64) constants of the batched AO environment.

Everything here runs once per geometry at ``set_params`` time and produces the tables that
``libaoenv`` consumes (include/aoenv.h, ``enum AoConst``).  The per-step physics never runs here: the
wave-front-sensor calibration itself (reference slopes, slope units, interaction matrix) is measured
on the GPU by the same HIP kernels the loop uses, in float64 (rlao_amd/env.py).

Reference parity citations:  OOPAO/ = drl4ao/AO_OOPAO/OOPAO/ ,  MAIN/ = drl4ao/MAIN_CODE/ .
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from types import SimpleNamespace
from typing import Sequence

import numpy as np

ATM_WAVELENGTH = 500e-9           # OOPAO/Atmosphere.py:134
R0_DEF = 0.15                     # OOPAO/Atmosphere.py:124

# (wavelength, bandwidth, zero point) -- OOPAO/Source.py:170-232
PHOTOMETRY = {
    "U": (0.360e-6, 0.070e-6, 1.96e12), "B": (0.440e-6, 0.100e-6, 5.38e12), "V0": (0.500e-6, 0.090e-6, 3.64e12),
    "V": (0.550e-6, 0.090e-6, 3.31e12), "R": (0.640e-6, 0.150e-6, 4.01e12), "R2": (0.650e-6, 0.300e-6, 7.9e12),
    "R3": (0.600e-6, 0.300e-6, 8.56e12), "R4": (0.670e-6, 0.300e-6, 7.66e12), "I": (0.790e-6, 0.150e-6, 2.69e12),
    "I1": (0.700e-6, 0.033e-6, 0.67e12), "I2": (0.750e-6, 0.033e-6, 0.62e12), "I3": (0.800e-6, 0.033e-6, 0.58e12),
    "I4": (0.700e-6, 0.100e-6, 2.02e12), "I5": (0.850e-6, 0.100e-6, 1.67e12), "I6": (1.000e-6, 0.100e-6, 1.42e12),
    "I7": (0.850e-6, 0.300e-6, 5.00e12), "I8": (0.750e-6, 0.100e-6, 1.89e12), "I9": (0.850e-6, 0.300e-6, 5.00e12),
    "I10": (0.900e-6, 0.300e-6, 4.72e12), "J": (1.215e-6, 0.260e-6, 1.90e12), "J2": (1.550e-6, 0.260e-6, 1.49e12),
    "H": (1.654e-6, 0.290e-6, 1.05e12), "Kp": (2.1245e-6, 0.351e-6, 0.62e12), "Ks": (2.157e-6, 0.320e-6, 0.55e12),
    "K": (2.179e-6, 0.410e-6, 0.70e12), "K0": (2.000e-6, 0.410e-6, 0.76e12), "K1": (2.400e-6, 0.410e-6, 0.64e12),
    "L": (3.547e-6, 0.570e-6, 2.5e11), "M": (4.769e-6, 0.450e-6, 8.4e10), "Na": (0.589e-6, 0, 3.3e12),
    "EOS": (1.064e-6, 0, 3.3e12), "IR1310": (1.310e-6, 0, 2e12),
}


# ------------------------------------------------------------------------------------------------------
# Parameters: the keys of MAIN/Conf/parameterFile_oopao_parser.py:19-79 (+ the Razor spelling fractionalR0)
# ------------------------------------------------------------------------------------------------------
@dataclass
class AOParams:
    diameter: float = 8.0
    nSubaperture: int = 20
    nPixelPerSubap: int = 6
    samplingTime: float = 1 / 500
    centralObstruction: float = 0.0
    magnitude: float = 8.0
    opticalBand: str = "I"
    mechanicalCoupling: float = 0.35
    r0: float = 0.13
    L0: float = 30.0
    fractionalR0: Sequence[float] = (1.0,)
    windSpeed: Sequence[float] = (10.0,)
    windDirection: Sequence[float] = (72.0,)
    altitude: Sequence[float] = (0.0,)
    nLoop: int = 10000
    gainCL: float = 0.5
    leak: float = 0.99
    lightThreshold: float = None         # SH lightRatio 0.5 (MAIN/OOPAOEnv/OOPAOEnvRazor.py:236); Pyramid 0.1 (parameterFile:67)
    modulation: float = 0.0              # Pyramid modulation radius in lambda/D (papyrus_config.yaml:17)
    n_pix_separation: int = 4            # parameterFile_oopao_parser.py:65
    psfCentering: bool = True            # Pyramid default used by OOPAOEnv.set_params (the call does not pass it)
    postProcessing: str = "slopesMaps_incidence_flux"
    threshold_cog: float = 0.01          # OOPAO/ShackHartmann.py:42
    nModes: int = 50                     # Zernike modes kept from the M2C (MAIN/OOPAOEnv/OOPAOEnv.py:260)
    nMeasurements: int = 6               # MAIN/OOPAOEnv/OOPAOEnv.py:285
    fov: float = 0.0
    extra: dict = field(default_factory=dict)

    @property
    def resolution(self) -> int:
        return int(self.nSubaperture * self.nPixelPerSubap)

    @property
    def nActuator(self) -> int:
        return int(self.nSubaperture + 1)

    @property
    def nLayer(self) -> int:
        return len(self.fractionalR0)


def params_from_args(args=None, **overrides) -> AOParams:
    """Accepts the YAML namespace / dict the reference passes to ``set_params`` (MAIN/PO4AO/mbrl.py:22-31)."""
    src = {}
    if args is not None:
        src = dict(vars(args)) if isinstance(args, SimpleNamespace) or hasattr(args, "__dict__") else dict(args)
    src.update(overrides)
    if "fractionnalR0" in src and "fractionalR0" not in src:       # [sic] Papyrus spelling
        src["fractionalR0"] = src.pop("fractionnalR0")
    p = AOParams()
    known = set(AOParams.__dataclass_fields__) - {"extra"}
    for k, v in src.items():
        if k in known:
            setattr(p, k, v)
        else:
            p.extra[k] = v
    for k in ("fractionalR0", "windSpeed", "windDirection", "altitude"):
        setattr(p, k, [float(x) for x in getattr(p, k)])
    n = p.nLayer
    if not (len(p.windSpeed) == len(p.windDirection) == len(p.altitude) == n):
        raise ValueError("fractionalR0, windSpeed, windDirection and altitude must have one entry per layer")
    return p


# ------------------------------------------------------------------------------------------------------
# Telescope / source
# ------------------------------------------------------------------------------------------------------
def telescope_pupil(resolution: int, central_obstruction: float = 0.0) -> np.ndarray:
    """Circular pupil mask (OOPAO/Telescope.py:164-180)."""
    x = np.linspace(-resolution / 2, resolution / 2, resolution)
    r2 = x[None, :] ** 2 + x[:, None] ** 2
    rim = (resolution + 1) / 2
    return (r2 < rim ** 2) & (r2 >= (central_obstruction * rim) ** 2)


def source(band: str, magnitude: float):
    """(wavelength, nPhoton) of a natural guide star (OOPAO/Source.py:100-109)."""
    if band not in PHOTOMETRY:
        raise ValueError(f"unknown optical band {band!r}")
    wl, _, zp = PHOTOMETRY[band]
    return wl, (zp / 368) * 10 ** (-0.4 * magnitude)


# ------------------------------------------------------------------------------------------------------
# Atmosphere tables
# ------------------------------------------------------------------------------------------------------
def _vk_covariance(za: np.ndarray, zb: np.ndarray, L0: float) -> np.ndarray:
    """von Karman phase covariance at r0_def (OOPAO/phaseStats.py:70-133)."""
    from scipy.special import kv
    rho = np.abs(za[:, None] - zb[None, :])
    ratio = (L0 / R0_DEF) ** (5.0 / 3)
    g = math.gamma
    head = (24.0 * g(6.0 / 5) / 5) ** (5.0 / 6)
    cst = head * (g(11.0 / 6) / ((2.0 ** (5.0 / 6)) * np.pi ** (8.0 / 3))) * ratio
    out = np.ones(rho.shape) * head * (g(11.0 / 6) * g(5.0 / 6) / (2 * np.pi ** (8.0 / 3))) * ratio
    nz = rho != 0
    u = 2 * np.pi * rho[nz] / L0
    out[nz] = cst * u ** (5.0 / 6) * kv(5.0 / 6, u)
    return out


class LayerTables:
    """Ring geometry and operators of ONE layer grid of N x N pixels (OOPAO/Atmosphere.py:216-298)."""

    def __init__(self, N: int, R: int, D: float, L0: float):
        self.N = N
        self.S = S = N + 2
        self.layer_D = N * D / R                               # :219
        self.L0 = L0
        ring = np.zeros((S, S), bool)
        ring[0, :] = ring[-1, :] = ring[:, 0] = ring[:, -1] = True
        inner = ~ring
        inner[3:-3, 3:-3] = False
        self.outer_mask, self.inner_mask = ring, inner
        self.outer_idx = np.flatnonzero(ring).astype(np.int32)   # boolean-mask (row-major) order, :291, :309
        self.inner_idx = np.flatnonzero(inner).astype(np.int32)
        self.n_outer, self.n_inner = self.outer_idx.size, self.inner_idx.size
        axis = np.linspace(0, N + 1, N + 2) * self.layer_D / (N - 1)          # :271
        u, v = np.meshgrid(axis, axis)
        zin = (u + 1j * v)[inner]
        zout = (u + 1j * v)[ring]
        self._zz = _vk_covariance(zin, zin, L0)
        self._zx = _vk_covariance(zin, zout, L0)
        self._xx = _vk_covariance(zout, zout, L0)
        self._zz_inv = np.linalg.pinv(self._zz)
        self.foot = N // 2 - R // 2                             # first row / column of the on-axis R x R footprint (:226-232)

    def set_r0(self, r0: float):
        """A = ZXt^T ZZt^-1, B = chol(XXt - A ZXt) at the requested r0 (:284-286, :554-557)."""
        s = (R0_DEF / r0) ** (5.0 / 3)
        self.A = np.matmul((self._zx * s).T, self._zz_inv / s)
        self.B = np.linalg.cholesky(self._xx * s - np.matmul(self.A, self._zx * s))
        self.AB = np.ascontiguousarray(np.concatenate([self.A, self.B], axis=1))


class AtmosphereTables:
    """The layers' grids and ring operators (OOPAO/Atmosphere.py:192-298).  With fov = 0, or with every layer on the ground, all
    layers share the (R + 4)^2 grid; with a field of view (the reference env's telescope: fov = 1 arcsec, MAIN/OOPAOEnv/OOPAOEnv.py:129)
    a layer at altitude h has ceil(R / D (D + 2 tan(fov / 2) h)) + 4 pixels across and operators of its own (:216-218).  ``layers[l]``
    are the tables of layer l; the attributes N, S, A, B, AB, inner_idx ... are those of layer 0 (every layer's when ``uniform``)."""

    def __init__(self, p: AOParams):
        R, D = p.resolution, p.diameter
        fov_rad = float(p.fov) / 206265.0                       # arcsec -> rad as OOPAO/Telescope.py does
        grids = {}
        self.layers = []
        for h in p.altitude:
            d_fov = D + 2 * np.tan(fov_rad / 2) * h
            N = int(np.ceil((R / D) * d_fov)) + 4
            if N not in grids:
                grids[N] = LayerTables(N, R, D, p.L0)
            self.layers.append(grids[N])
        if not self.layers:
            self.layers = [LayerTables(R + 4, R, D, p.L0)]
        self._grids = list(grids.values()) or self.layers
        self.uniform = len({t.N for t in self.layers}) == 1
        self.layer_res = [t.N for t in self.layers]
        g0 = self.layers[0]
        self.N, self.S, self.layer_D = g0.N, g0.S, g0.layer_D
        self.ps_loop = g0.layer_D / g0.N                        # :351 (the pixel size D / R, the same for every layer)
        self.outer_mask, self.inner_mask = g0.outer_mask, g0.inner_mask
        self.outer_idx, self.inner_idx = g0.outer_idx, g0.inner_idx
        self.n_outer, self.n_inner = g0.n_outer, g0.n_inner
        self.weights = np.sqrt(np.asarray(p.fractionalR0, float))
        self.set_r0(p.r0)

    def set_r0(self, r0: float):
        for t in self._grids:
            t.set_r0(r0)
        g0 = self.layers[0]
        self.A, self.B, self.AB = g0.A, g0.B, g0.AB

    def wind_ratio(self, speeds, directions, dt):
        """pixels per frame along (x, y) for each layer (:209-210, :352-363)."""
        out = np.zeros((len(speeds), 2))
        for l, (ws, wd) in enumerate(zip(speeds, directions)):
            vy = ws * np.cos(np.deg2rad(wd))
            vx = ws * np.sin(np.deg2rad(wd))
            out[l] = [vx * dt / self.ps_loop, vy * dt / self.ps_loop]
        return out


# ------------------------------------------------------------------------------------------------------
# Deformable mirror
# ------------------------------------------------------------------------------------------------------
class DMTables:
    """Cartesian Fried-geometry DM with Gaussian influence functions and zero mis-registration
    (OOPAO/DeformableMirror.py:286-305 valid actuators, :494-514 influence model).  With no rotation /
    anamorphosis the influence function of actuator (iy, ix) is gy[:, iy] (x) gx[:, ix]."""

    def __init__(self, p: AOParams, pitch: float | None = None, n_subap: int | None = None):
        R, D = p.resolution, p.diameter
        ns = p.nSubaperture if n_subap is None else int(n_subap)      # a second DM has its own actuator pitch
        self.nAct = nAct = ns + 1
        self.pitch = D / nAct if pitch is None else pitch        # MAIN/OOPAOEnv/OOPAOEnv.py:228
        x = np.linspace(-D / 2, D / 2, nAct)
        X, Y = np.meshgrid(x, x)
        rad = np.sqrt(X.reshape(-1) ** 2 + Y.reshape(-1) ** 2)
        self.validAct = (rad > (p.centralObstruction * D / 2 - 0.5 * self.pitch)) & (rad <= (D / 2 + 0.7533 * self.pitch))
        self.act_idx = np.flatnonzero(self.validAct).astype(np.int32)       # iy * nAct + ix
        self.nValidAct = int(self.act_idx.size)
        self.dm_mask = self.validAct.reshape(nAct, nAct)
        self.xvalid, self.yvalid = np.nonzero(self.dm_mask)                 # MAIN/OOPAOEnv/OOPAOEnv.py:231-232
        centre = R / 2 + x * R / D
        width = (R / ns) / np.sqrt(2 * np.log(1.0 / p.mechanicalCoupling))
        pix = np.linspace(0, 1, R) * R
        self.gx = np.exp(-((pix[:, None] - centre[None, :]) ** 2) / (2 * width ** 2))     # [R, nAct]
        self.gy = self.gx.copy()
        self._R = R
        self._centre, self._a, self._pix = centre, 1.0 / (2 * width ** 2), pix

    def dense_modes(self) -> np.ndarray:
        """dm.modes [R*R, nValidAct], evaluated with the reference's own expression (:506-511)."""
        XX, YY = np.meshgrid(self._pix, self._pix)
        x0 = self._centre[self.act_idx % self.nAct]
        y0 = self._centre[self.act_idx // self.nAct]
        a = self._a
        return np.exp(-(a * (XX.reshape(-1, 1) - x0[None, :]) ** 2 + a * (YY.reshape(-1, 1) - y0[None, :]) ** 2))


class CompositeDM:
    """Two deformable mirrors chained in the beam, ``tel*dm1*dm2*wfs`` (OOPAO/Telescope.py:533-544: every DM adds its OPD;
    with fov = 0 an altitude-conjugated DM has the ground DM's grid, DeformableMirror.py:388-389).  Presented to the library
    as ONE separable DM: command vector [dm1 | dm2], factors Gy = [gy1 | gy2], Gx = [gx1 | gx2], and an actuator "image" of
    side nAct1 + nAct2 that holds dm1's grid in its top-left block and dm2's in the bottom-right one.  The off-diagonal blocks
    hold no actuator, so Gy C Gx^T = gy1 C1 gx1^T + gy2 C2 gx2^T: the sum of the two mirrors' surfaces, through the same
    kernels (fused step kernel included, nAct1 + nAct2 <= 32) as a single mirror."""

    def __init__(self, p: AOParams, n_subap_2: int):
        self.dm1, self.dm2 = DMTables(p), DMTables(p, n_subap=n_subap_2)
        n1, n2 = self.dm1.nAct, self.dm2.nAct
        self.nAct = n1 + n2
        mask = np.zeros((self.nAct, self.nAct), bool)
        mask[:n1, :n1] = self.dm1.dm_mask
        mask[n1:, n1:] = self.dm2.dm_mask
        self.dm_mask = mask
        self.validAct = mask.reshape(-1)
        self.act_idx = np.flatnonzero(self.validAct).astype(np.int32)
        self.nValidAct = int(self.act_idx.size)
        self.xvalid, self.yvalid = np.nonzero(mask)
        self.gx = np.hstack([self.dm1.gx, self.dm2.gx])
        self.gy = np.hstack([self.dm1.gy, self.dm2.gy])

    def dense_modes(self) -> np.ndarray:
        return np.hstack([self.dm1.dense_modes(), self.dm2.dense_modes()])


# ------------------------------------------------------------------------------------------------------
# Shack-Hartmann geometry
# ------------------------------------------------------------------------------------------------------
class SHTables:
    """Valid-lenslet selection and field amplitude (OOPAO/ShackHartmann.py:154-236, 327-338)."""

    def __init__(self, p: AOParams, pupil: np.ndarray, n_photon: float):
        R, ns = p.resolution, p.nSubaperture
        self.p = px = R // ns
        self.n = 2 * px
        self.cam_res = ns * px
        self.flux_map = pupil.astype(float) * n_photon * p.samplingTime * (p.diameter / R) ** 2   # OOPAO/Source.py:151
        # lenslet k = i*ns + j sees flux_map.T[j*px:(j+1)*px, i*px:(i+1)*px]  (:331-335)
        per = self.flux_map.T.reshape(ns, px, ns, px).sum(axis=(1, 3)).T.reshape(-1)
        self.valid_1d = per >= (0.5 if p.lightThreshold is None else p.lightThreshold) * per.max()
        self.valid_2d = self.valid_1d.reshape(ns, ns)
        self.subap_idx = np.flatnonzero(self.valid_1d).astype(np.int32)
        self.nValid = int(self.subap_idx.size)
        self.nSignal = 2 * self.nValid
        self.amp = np.sqrt(self.flux_map)


# ------------------------------------------------------------------------------------------------------
# Pyramid geometry
# ------------------------------------------------------------------------------------------------------
class PyramidTables:
    """Sizes, focal-plane mask, modulation path and quadrant geometry of the Pyramid WFS
    (OOPAO/Pyramid.py:251-297 sizes, :368-405 mask, :941-976 modulation, :774-790 quadrants)."""

    def __init__(self, p: AOParams, pupil: np.ndarray, n_photon: float, psf_centering: bool = True,
                 n_pix_separation: int = 4, n_pix_edge: int = 2, calib_modulation: float = 50,
                 post_processing: str = "slopesMaps_incidence_flux"):
        R, ns = p.resolution, p.nSubaperture
        if (R / ns) % 2 != 0:
            raise ValueError("The resolution should be an even number and be a multiple of 2**i where i>=2")   # :210-211
        if post_processing not in ("slopesMaps", "slopesMaps_incidence_flux"):
            raise NotImplementedError("only the slopes-maps post-processings are built (full-frame modes are out of scope)")
        self.R, self.nSubap = R, ns
        self.psf_centering = bool(psf_centering)
        self.norm_valid = 1 if post_processing == "slopesMaps" else 0
        self.n_pix_separation, self.n_pix_edge = n_pix_separation, n_pix_edge
        self.nRes = int((ns * 2 + n_pix_separation + n_pix_edge * 2) * R / ns)
        self.cam_res = round(ns * (self.nRes / R))
        self.calib_modulation = R / 2 - 1 if calib_modulation >= R / 2 else calib_modulation
        self.pupil_f = pupil.astype(float)
        self.flux_map = self.pupil_f * n_photon * p.samplingTime * (p.diameter / R) ** 2
        tip, tilt = np.meshgrid(np.linspace(-np.pi, np.pi, R), np.linspace(-np.pi, np.pi, R))
        self.Tip, self.Tilt = tip * self.pupil_f, tilt * self.pupil_f
        ne = int(np.round((n_pix_separation / ns) * R / (R / ns) / 2))
        c = int(np.round(self.cam_res / 2))
        self.q_lo, self.q_hi = c - ne - int(np.ceil(ns)), c + ne
        self.m = self._phase_mask()
        mk = np.complex64(np.exp(1j * self.m))                       # complex64, as the reference stores it (:323)
        self.mask_pairs = np.stack([mk.real.astype(np.float64), mk.imag.astype(np.float64)], axis=-1)

    def _phase_mask(self) -> np.ndarray:
        n_tot, ns, sep = self.nRes, self.nSubap, self.n_pix_separation
        norma = (ns + sep) * (self.R / ns)
        m = np.zeros([n_tot, n_tot])
        h = n_tot // 2
        if self.psf_centering:                                        # mask centred on 4 pixels
            lim = np.pi / 4 - np.pi / 4 / h
            a, b = np.meshgrid(np.linspace(-lim, lim, h), np.linspace(-lim, lim, h))
            m[:h, :h] = a * norma + b * norma
            m[:h, -h:] = -a * norma + b * norma
            m[-h:, -h:] = -a * norma + -b * norma
            m[-h:, :h] = a * norma + -b * norma
        else:                                                         # mask centred on 1 pixel
            d_pix = (np.pi / 4) / (n_tot / 2)
            lp, lm = np.pi / 4, np.pi / 4 - 2 * d_pix
            t1, u1 = np.meshgrid(np.linspace(-lp, lp, h + 1), np.linspace(-lp, lp, h + 1))
            t2, u2 = np.meshgrid(np.linspace(-lp, lp, h + 1), np.linspace(-lm, lm, h - 1))
            t3, u3 = np.meshgrid(np.linspace(-lm, lm, h - 1), np.linspace(-lm, lm, h - 1))
            t4, u4 = np.meshgrid(np.linspace(-lm, lm, h - 1), np.linspace(-lp, lp, h + 1))
            m[:h + 1, :h + 1] = t1 * norma + u1 * norma
            m[:h + 1, -h + 1:] = -t4 * norma + u4 * norma
            m[-h + 1:, -h + 1:] = -t3 * norma + -u3 * norma
            m[-h + 1:, :h + 1] = t2 * norma + -u2 * norma
        return -m

    def modulation_table(self, modulation: float):
        """(nTheta, tt[nTheta, R, R]) -- the tip/tilt phases are float32 in the reference (:964-970)."""
        if modulation == 0:
            return 1, None
        n_theta = 4 * int(0 + np.ceil(np.pi * 2 * modulation / 4))
        theta = np.linspace(0, 2 * np.pi, n_theta, endpoint=False)
        buf = np.zeros([n_theta, self.R, self.R]).astype(np.float32)
        for i in range(n_theta):
            buf[i] = (modulation * np.cos(theta[i]) * self.Tip + modulation * np.sin(theta[i]) * self.Tilt) * self.pupil_f
        return n_theta, buf.astype(np.float64)

    def amplitude(self, n_theta: int) -> np.ndarray:
        return np.sqrt(self.flux_map / n_theta) * self.pupil_f       # :520

    def quadrant_sum(self, frame: np.ndarray) -> np.ndarray:
        lo, hi, p = self.q_lo, self.q_hi, self.nSubap
        return (frame[lo:lo + p, lo:lo + p] + frame[lo:lo + p, hi:hi + p] + frame[hi:hi + p, hi:hi + p]
                + frame[hi:hi + p, lo:lo + p])


def tip_ramp(R: int) -> np.ndarray:
    """Unit tip used for the slope-unit calibration (OOPAO/ShackHartmann.py:286-290).  The reference
    normalises with ``np.std(Tip[tel.pupil])`` where tel.pupil is an *integer* 0/1 array, i.e. the std of
    whole rows 0 / 1 of the ramp = the std of one row; kept as is."""
    tip, _ = np.meshgrid(np.linspace(0, np.pi, R, endpoint=False), np.linspace(0, np.pi, R, endpoint=False))
    return tip * (1 / np.std(tip[0]))


# ------------------------------------------------------------------------------------------------------
# Modal basis and reconstructor
# ------------------------------------------------------------------------------------------------------
def _noll(j: int):
    n = int((-1.0 + np.sqrt(8 * (j - 1) + 1)) / 2.0)
    k = n % 2
    m = int(((j - (n * (n + 1)) / 2.0) + k) / 2.0) * 2 - k
    if m != 0:
        m *= 1 if j % 2 == 0 else -1
    return n, m


def _radial(n: int, m: int, r: np.ndarray) -> np.ndarray:
    out = np.zeros(r.shape)
    for i in range(0, int((n - m) / 2) + 1):
        out += r ** (n - 2.0 * i) * (((-1) ** i) * math.factorial(n - i)) / (
            math.factorial(i) * math.factorial(int(0.5 * (n + m) - i)) * math.factorial(int(0.5 * (n - m) - i)))
    return out


def zernike_basis(pupil: np.ndarray, D: float, n_modes: int) -> np.ndarray:
    """Noll Zernike modes 2..n_modes+1 on the pupil pixels, mean-removed, unit std (OOPAO/Zernike.py:26-66)."""
    R = pupil.shape[0]
    X, Y = np.where(pupil > 0)
    X = (X - (R + R % 2 - 1) / 2) / R * D
    Y = (Y - (R + R % 2 - 1) / 2) / R * D
    rr = np.sqrt(X ** 2 + Y ** 2)
    rr = rr / rr.max()
    th = np.arctan2(Y, X)
    out = np.zeros((X.size, n_modes))
    for i in range(1, n_modes + 1):
        n, m = _noll(i + 1)
        if m == 0:
            Z = np.sqrt(n + 1) * _radial(n, 0, rr)
        elif m > 0:
            Z = np.sqrt(2 * (n + 1)) * _radial(n, m, rr) * np.cos(m * th)
        else:
            Z = np.sqrt(2 * (n + 1)) * _radial(n, -m, rr) * np.sin(-m * th)
        Z = Z - Z.mean()
        out[:, i - 1] = Z * (1 / np.std(Z))
    return out


def zernike_m2c(dm: DMTables, pupil: np.ndarray, D: float, n_modes: int) -> np.ndarray:
    """M2C = pinv(dm.modes[pupil]) @ Z.modes  (MAIN/OOPAOEnv/OOPAOEnv.py:258, OOPAOEnvRazor.py:261).
    Beyond ~2000 actuators (ELT: 5209 actuators x 181 k pupil pixels) the SVD behind ``pinv`` takes hours on the host:
    the same least-squares problem is then solved on the GPU in float64 (one-off set-up work, not the step path)."""
    modes = dm.dense_modes()[pupil.reshape(-1)]
    Z = zernike_basis(pupil, D, n_modes)
    if modes.shape[1] <= 2000:
        return np.linalg.pinv(modes) @ Z
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    A_ = torch.as_tensor(modes, device=dev)
    # normal equations through a Cholesky factor of the (well conditioned: Gaussian influence functions, coupling 0.35) Gram matrix
    G = A_.T @ A_
    rhs = A_.T @ torch.as_tensor(Z, device=dev)
    del A_
    return torch.linalg.solve(G, rhs).cpu().numpy()


def svd_inverse(D: np.ndarray) -> np.ndarray:
    """calib.M = V^T S^-1 U^T  (OOPAO/calibration/CalibrationVault.py:19-30)."""
    U, s, Vt = np.linalg.svd(D, full_matrices=False)
    return Vt.T @ np.diag(1 / s) @ U.T


def reconstructor_from_imat(imat: np.ndarray, m2c: np.ndarray, return_factors: bool = False):
    """(reconstructor, F) of MAIN/OOPAOEnv/OOPAOEnv.py:295, 381-383 [+ the modal command matrix calib.M]."""
    M = svd_inverse(imat @ m2c)
    if return_factors:
        return m2c @ M, m2c @ np.linalg.pinv(m2c), M
    return m2c @ M, m2c @ np.linalg.pinv(m2c)
