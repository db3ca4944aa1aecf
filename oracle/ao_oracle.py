"""CPU oracle for the adaptive-optics env-step hot path (TEST INFRASTRUCTURE ONLY).

This file is a plain NumPy float64 restatement of the algorithm that the reference
(artiom-matvei/RLAO: drl4ao on top of the vendored OOPAO simulator) executes behind
``env.step()``.  It is the *checker* for the HIP product path in ``rlao_amd/``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it.  Nothing under ``rlao_amd/`` imports or calls it; the product path fails loudly
when the HIP library is missing.

Pinning status
--------------
* Every stage except the sub-pixel image warp is pinned against the reference itself:
  ``oracle/make_goldens.py`` imports the reference's own ``Telescope / Source / Atmosphere /
  DeformableMirror / ShackHartmann / Pyramid / InteractionMatrix / CalibrationVault`` in the
  build container and writes the vectors in ``tests/golden/``; ``tests/test_oracle_golden.py``
  checks this file against them.
* ``warp_translate`` restates ``skimage.transform.warp(order=3, mode='constant', cval=0,
  clip=True)`` for a pure translation (scikit-image 0.18.3, pinned in the reference's
  ``AO_OOPAO/requirements.txt:20``).  scikit-image is a third-party dependency that is absent
  from /root/reference and from this image, and no reference file holds an output of it:
  **parity unpinned at that one stage** (property tests guard the frozen spec).

Path aliases in citations:  OOPAO/ = drl4ao/AO_OOPAO/OOPAO/ ,  MAIN/ = drl4ao/MAIN_CODE/ .
"""
from __future__ import annotations

import math

import numpy as np
from numpy.random import RandomState

TWO_PI = 2.0 * np.pi

# --------------------------------------------------------------------------------------
# Source photometry                                            (OOPAO/Source.py:164-242)
# --------------------------------------------------------------------------------------
_PHOTOMETRY = {            # band: (wavelength [m], bandwidth [m], zero point)
    "U": (0.360e-6, 0.070e-6, 1.96e12), "B": (0.440e-6, 0.100e-6, 5.38e12),
    "V0": (0.500e-6, 0.090e-6, 3.64e12), "V": (0.550e-6, 0.090e-6, 3.31e12),
    "R": (0.640e-6, 0.150e-6, 4.01e12), "I": (0.790e-6, 0.150e-6, 2.69e12),
    "J": (1.215e-6, 0.260e-6, 1.90e12), "H": (1.654e-6, 0.290e-6, 1.05e12),
    "K": (2.179e-6, 0.410e-6, 0.70e12),
}


def source_photometry(band: str, magnitude: float):
    """wavelength [m] and nPhoton [ph/m2/s]  (OOPAO/Source.py:100-109)."""
    wl, _bw, zp = _PHOTOMETRY[band]
    zero_point = zp / 368
    return wl, zero_point * 10 ** (-0.4 * magnitude)


# --------------------------------------------------------------------------------------
# Telescope pupil                                            (OOPAO/Telescope.py:164-180)
# --------------------------------------------------------------------------------------
def make_pupil(resolution: int, central_obstruction: float = 0.0) -> np.ndarray:
    d = resolution + 1
    x = np.linspace(-resolution / 2, resolution / 2, resolution)
    xx, yy = np.meshgrid(x, x)
    circle = xx ** 2 + yy ** 2
    obs = circle >= (central_obstruction * d / 2) ** 2
    return (circle < (d / 2) ** 2) & obs


# --------------------------------------------------------------------------------------
# von Karman statistics                                    (OOPAO/phaseStats.py:70-133)
# --------------------------------------------------------------------------------------
def covariance_matrix(z1: np.ndarray, z2: np.ndarray, L0: float, r0: float) -> np.ndarray:
    """Phase covariance between complex coordinate lists z1, z2 (makeCovarianceMatrix)."""
    from scipy.special import kv
    rho = np.abs(z1[:, None] - z2[None, :])                  # bsxfunMinus, tools.py:194-198
    ratio = (L0 / r0) ** (5.0 / 3)
    g65, g116, g56 = math.gamma(6.0 / 5), math.gamma(11.0 / 6), math.gamma(5.0 / 6)
    cst = (24.0 * g65 / 5) ** (5.0 / 6) * (g116 / (2.0 ** (5.0 / 6) * np.pi ** (8.0 / 3))) * ratio
    out = np.ones(rho.shape) * ((24.0 * g65 / 5) ** (5.0 / 6)) * (g116 * g56 / (2 * np.pi ** (8.0 / 3))) * ratio
    nz = rho != 0
    u = TWO_PI * rho[nz] / L0
    out[nz] = cst * u ** (5.0 / 6) * kv(5.0 / 6, u)
    return out


def ft_phase_screen(r0: float, L0: float, n: int, delta: float, seed: int, l0: float = 1e-10):
    """FFT phase screen, rad @ 500 nm  (OOPAO/phaseStats.py:190-235)."""
    rs = RandomState(seed)
    del_f = 1.0 / (n * delta)
    fx = np.arange(-n / 2.0, n / 2.0) * del_f
    fx, fy = np.meshgrid(fx, fx)
    f = np.sqrt(fx ** 2 + fy ** 2)
    fm = 5.92 / l0 / TWO_PI
    f0 = 1.0 / L0
    psd = 0.023 * r0 ** (-5.0 / 3.0) * np.exp(-1 * ((f / fm) ** 2)) / ((f ** 2 + f0 ** 2) ** (11.0 / 6))
    psd[int(n / 2), int(n / 2)] = 0
    cn = (rs.normal(size=(n, n)) + 1j * rs.normal(size=(n, n))) * np.sqrt(psd) * del_f
    return np.fft.fftshift(np.fft.fft2(np.fft.fftshift(cn))).real      # ift2, :170-187


def ft_sh_phase_screen(r0: float, L0: float, n: int, delta: float, seed: int, l0: float = 1e-10):
    """FFT screen + 3 sub-harmonic grids (OOPAO/phaseStats.py:243-318).

    Quirks kept: the sub-harmonic RandomState is re-created from the same seed as the
    high-frequency screen (:268,:272) and only i,j in {0,1} of each 3x3 grid are summed (:306-309).
    """
    rs = RandomState(seed)
    D = n * delta
    hi = ft_phase_screen(r0, L0, n, delta, seed, l0)
    coords = np.arange(-n / 2, n / 2) * delta
    x, y = np.meshgrid(coords, coords)
    lo = np.zeros(hi.shape, dtype=complex)
    for p in range(1, 4):
        del_f = 1 / (3 ** p * D)
        fx = np.arange(-1, 2) * del_f
        fx, fy = np.meshgrid(fx, fx)
        f = np.sqrt(fx ** 2 + fy ** 2)
        fm = 5.92 / l0 / TWO_PI
        f0 = 1.0 / L0
        psd = 0.023 * r0 ** (-5.0 / 3) * np.exp(-1 * (f / fm) ** 2) / ((f ** 2 + f0 ** 2) ** (11.0 / 6))
        psd[1, 1] = 0
        cn = (rs.normal(size=(3, 3)) + 1j * rs.normal(size=(3, 3))) * np.sqrt(psd) * del_f
        sh = np.zeros((n, n), dtype=complex)
        for i in range(0, 2):
            for j in range(0, 2):
                sh += cn[i, j] * np.exp(1j * TWO_PI * (fx[i, j] * x + fy[i, j] * y))
        lo = lo + sh
    lo = lo.real - lo.real.mean()
    return lo + hi


# --------------------------------------------------------------------------------------
# Image translation = skimage.transform.warp(img, SimilarityTransform(translation).inverse, order=3)
# call sites: OOPAO/tools/tools.py:210-217  <-  OOPAO/Atmosphere.py:304-305, 406-407
# FROZEN SPEC (scikit-image 0.18.3 `_warp_fast` + `_clip_warp_output`), parity unpinned:
#   out[r, c] = bicubic(img, r - ty, c - tx)   with translation = (tx, ty), x = column
#   bicubic   = separable Catmull-Rom on the 4x4 neighbourhood, rows first then columns,
#               cubic(x; f0..f3) = f1 + 0.5 x (f2 - f0 + x (2 f0 - 5 f1 + 4 f2 - f3 + x (3 (f1 - f2) + f3 - f0)))
#   pixels outside the image read as cval = 0 (mode='constant')
#   result clipped to [img.min(), img.max()]; if 0 is outside that range, outputs that are
#   exactly 0 are kept at 0 (preserve_cval).
# --------------------------------------------------------------------------------------
def _cubic(x, f0, f1, f2, f3):
    return f1 + 0.5 * x * (f2 - f0 + x * (2.0 * f0 - 5.0 * f1 + 4.0 * f2 - f3 + x * (3.0 * (f1 - f2) + f3 - f0)))


def warp_translate(img: np.ndarray, tx: float, ty: float) -> np.ndarray:
    rows, cols = img.shape
    rr = np.arange(rows, dtype=np.float64) - ty
    cc = np.arange(cols, dtype=np.float64) - tx
    r0 = np.floor(rr).astype(np.int64)
    c0 = np.floor(cc).astype(np.int64)
    xr = rr - r0
    xc = cc - c0
    pad = np.zeros((rows + 8, cols + 8), dtype=np.float64)       # zero apron == cval outside
    pad[4:-4, 4:-4] = img
    ri = np.clip(r0 - 1 + 4, 0, rows + 4)                         # |t| < rows is assumed; clip is a guard
    ci = np.clip(c0 - 1 + 4, 0, cols + 4)
    # interpolate along columns (x) for the 4 source rows, then along rows (y)
    fr = []
    for pr in range(4):
        rsel = pad[np.clip(ri + pr, 0, rows + 7)]
        f = [rsel[:, np.clip(ci + pc, 0, cols + 7)] for pc in range(4)]
        fr.append(_cubic(xc[None, :], f[0], f[1], f[2], f[3]))
    out = _cubic(xr[:, None], fr[0], fr[1], fr[2], fr[3])
    lo, hi = img.min(), img.max()
    keep0 = None
    if not (lo <= 0.0 <= hi):
        keep0 = out == 0.0
    out = np.clip(out, lo, hi)
    if keep0 is not None:
        out[keep0] = 0.0
    return out


# --------------------------------------------------------------------------------------
# Atmosphere                                                       (OOPAO/Atmosphere.py)
# --------------------------------------------------------------------------------------
class LayerGeometry:
    """Shared per-geometry constants of one turbulence layer (buildLayer :192-298)."""

    def __init__(self, tel_resolution: int, tel_D: float, L0: float, r0_def: float = 0.15,
                 altitude: float = 0.0, fov_rad: float = 0.0):
        self.D_fov = tel_D + 2 * np.tan(fov_rad / 2) * altitude
        self.resolution_fov = int(np.ceil((tel_resolution / tel_D) * self.D_fov))
        self.resolution = n = self.resolution_fov + 4
        self.D = n * tel_D / tel_resolution
        self.d0 = self.D / n
        self.nPixel = int(1 + np.round(self.D / self.d0))
        outer = np.ones((n + 2, n + 2))
        outer[1:-1, 1:-1] = 0
        inner = np.ones((n + 2, n + 2)) - outer
        inner[3:-3, 3:-3] = 0
        self.outerMask = outer != 0
        self.innerMask = inner != 0
        self.innerMaskCrop = self.innerMask[1:-1, 1:-1]
        l = np.linspace(0, n + 1, n + 2) * self.D / (n - 1)
        u, v = np.meshgrid(l, l)
        self.innerZ = u[self.innerMask] + 1j * v[self.innerMask]
        self.outerZ = u[self.outerMask] + 1j * v[self.outerMask]
        self.L0 = L0
        self.r0_def = r0_def
        self._cov = None
        # on-axis source: centred R x R footprint (:226-232)
        c = n // 2
        h = tel_resolution // 2
        self.foot = (slice(c - h, c + h), slice(c - h, c + h))

    def covariances(self):
        if self._cov is None:
            ZZt = covariance_matrix(self.innerZ, self.innerZ, self.L0, self.r0_def)
            ZXt = covariance_matrix(self.innerZ, self.outerZ, self.L0, self.r0_def)
            XXt = covariance_matrix(self.outerZ, self.outerZ, self.L0, self.r0_def)
            self._cov = (ZZt, ZXt, XXt, np.linalg.pinv(ZZt))
        return self._cov

    def AB(self, r0: float):
        """A = ZXt_r0^T ZZt_inv_r0 ; B = chol(XXt_r0 - A ZXt_r0)   (:284-286, :554-557)."""
        ZZt, ZXt, XXt, ZZt_inv = self.covariances()
        s = (self.r0_def / r0) ** (5.0 / 3)
        A = np.matmul((ZXt * s).T, ZZt_inv / s)
        B = np.linalg.cholesky(XXt * s - np.matmul(A, ZXt * s))
        return A, B


class OracleLayer:
    def __init__(self, geom: LayerGeometry, A, B, wind_speed, wind_dir_deg, frac_r0):
        self.g = geom
        self.A, self.B = A, B
        self.frac = frac_r0
        self.set_wind(wind_speed, wind_dir_deg)
        n = geom.resolution
        self.mapShift = np.zeros((n + 2, n + 2))
        self.phase = np.zeros((n, n))
        self.rs = RandomState(0)
        self.notDoneOnce = True
        self.ratio = np.zeros(2)
        self.buff = np.zeros(2)

    def set_wind(self, speed, direction_deg):
        self.speed, self.direction = speed, direction_deg
        self.vY = speed * np.cos(np.deg2rad(direction_deg))      # :209-210
        self.vX = speed * np.sin(np.deg2rad(direction_deg))

    def _ring(self, interior):
        Z = interior[self.g.innerMaskCrop]
        return self.A @ Z + self.B @ self.rs.normal(size=self.B.shape[1])

    def new_screen(self, r0, L0, seed, i_layer):
        """generateNewPhaseScreen, per layer (:565-588)."""
        g = self.g
        self.phase = ft_sh_phase_screen(r0, L0, g.resolution, g.D / g.resolution, seed + i_layer)
        self.rs = RandomState(seed + i_layer * 1000)
        X = self._ring(self.phase)
        self.mapShift[g.outerMask] = X
        self.mapShift[~g.outerMask] = self.phase.reshape(-1)
        self.notDoneOnce = True

    def add_row(self, step):
        """One-pixel integer shift + regeneration of the whole outer ring (:301-311)."""
        g = self.g
        shifted = warp_translate(self.mapShift, step[0], step[1])[1:-1, 1:-1]
        X = self._ring(shifted)
        self.mapShift[g.outerMask] = X
        self.mapShift[~g.outerMask] = shifted.reshape(-1)
        return shifted

    def update(self, dt, ps_loop):
        """updateLayer (:350-407)."""
        if self.vX == 0 and self.vY == 0:
            return
        if self.notDoneOnce:
            self.notDoneOnce = False
            self.ratio = np.array([self.vX * dt / ps_loop, self.vY * dt / ps_loop])
            self.buff = np.zeros(2)
        ratio = self.ratio
        tmp = np.abs(ratio)
        tmp[np.isinf(tmp)] = 0
        nscr = tmp.astype(int)
        for _ in range(nscr.min()):
            self.phase = self.add_row(np.ones(2) * np.sign(ratio))
        for _ in range(nscr.max() - nscr.min()):
            step = np.ones(2) * np.sign(ratio)
            step[np.where(nscr == nscr.min())] = 0
            self.phase = self.add_row(step)
        sub = (np.abs(ratio) % 1) * np.sign(ratio)
        self.buff = self.buff + sub
        if np.abs(self.buff[0]) >= 1 or np.abs(self.buff[1]) >= 1:
            step = 1 * np.sign(self.buff)
            step[np.where(np.abs(self.buff) < 1)] = 0
            self.phase = self.add_row(step)
        self.buff = (np.abs(self.buff) % 1) * np.sign(self.buff)
        self.phase = warp_translate(self.mapShift, self.buff[0], self.buff[1])[1:-1, 1:-1]


class OracleAtmosphere:
    """Frozen-flow multi-layer atmosphere, on-axis NGS, fov = 0 (OOPAO/Atmosphere.py)."""

    wavelength = 500e-9

    def __init__(self, tel_resolution, tel_D, dt, pupil, r0, L0, windSpeed, fractionalR0,
                 windDirection, altitude, fov_rad=0.0, geom_AB=None):
        """geom_AB = (LayerGeometry, A, B): operators computed by the caller (fov = 0; the ELT-size ones take minutes)."""
        self.R, self.D, self.dt, self.pupil = tel_resolution, tel_D, dt, pupil
        self.r0, self.L0 = r0, L0
        self.fractionalR0 = list(fractionalR0)
        self.nLayer = len(self.fractionalR0)
        self.layers = []
        geom = None
        if geom_AB is not None and fov_rad == 0:
            geom, A, B = geom_AB
        for i in range(self.nLayer):
            if geom is None or fov_rad != 0:
                geom = LayerGeometry(tel_resolution, tel_D, L0, altitude=altitude[i], fov_rad=fov_rad)
                A, B = geom.AB(r0)
            self.layers.append(OracleLayer(geom, A, B, windSpeed[i], windDirection[i], self.fractionalR0[i]))
        self.OPD = None
        self.OPD_no_pupil = None

    def set_wind_speed(self, speeds):
        """windSpeed setter: recomputes ratio in place, keeps buff (:829-847)."""
        for lay, s in zip(self.layers, speeds):
            lay.set_wind(s, lay.direction)
            if not lay.notDoneOnce:
                ps = lay.g.D / lay.g.resolution
                lay.ratio = np.array([lay.vX * self.dt / ps, lay.vY * self.dt / ps])

    def _collect(self):
        sup = np.zeros((self.R, self.R))
        for lay in self.layers:
            sup += lay.phase[lay.g.foot] * np.sqrt(lay.frac)                # :439-450
        self.OPD_no_pupil = sup * self.wavelength / 2 / np.pi               # :474-477
        self.OPD = self.OPD_no_pupil * self.pupil

    def generate_new_phase_screen(self, seed):
        for i, lay in enumerate(self.layers):
            lay.new_screen(self.r0, self.L0, seed, i)
        self._collect()

    def update(self):
        for lay in self.layers:
            lay.update(self.dt, lay.g.D / lay.g.resolution)
        self._collect()


# --------------------------------------------------------------------------------------
# Deformable mirror                                          (OOPAO/DeformableMirror.py)
# --------------------------------------------------------------------------------------
def dm_geometry(resolution, D, n_subap, mech_coupling=0.35, pitch=None, central_obstruction=0.0, dense=True):
    """Fried-geometry Gaussian DM: valid mask, separable factors and dense IF matrix.

    valid-actuator rule :300-305; IF model :494-514 (zero mis-registration => separable).
    Returns dict(validAct[nAct^2] bool, gx[R,nAct], gy[R,nAct], modes[R^2, nValid]).
    """
    nAct = n_subap + 1
    if pitch is None:
        pitch = D / n_subap
    x = np.linspace(-D / 2, D / 2, nAct)
    X, Y = np.meshgrid(x, x)
    xIF0, yIF0 = X.reshape(-1), Y.reshape(-1)
    r = np.sqrt(xIF0 ** 2 + yIF0 ** 2)
    valid = (r > (central_obstruction * D / 2 - 0.5 * pitch)) & (r <= (D / 2 + 0.7533 * pitch))
    u0 = resolution / 2 + x * resolution / D                   # actuator centres on the pixel grid
    c = (resolution / (nAct - 1)) / np.sqrt(2 * np.log(1.0 / mech_coupling))
    px = np.linspace(0, 1, resolution) * resolution
    g = np.exp(-((px[:, None] - u0[None, :]) ** 2) / (2 * c ** 2))          # [R, nAct]
    if not dense:                                              # ELT size: the dense matrix is 9.6 GB
        return dict(nAct=nAct, validAct=valid, gx=g, gy=g.copy(), modes=None)
    XX, YY = np.meshgrid(px, px)
    a = 1.0 / (2 * c ** 2)
    k = np.nonzero(valid)[0]
    x0 = u0[k % nAct]
    y0 = u0[k // nAct]
    modes = np.exp(-(a * (XX.reshape(-1, 1) - x0[None, :]) ** 2 + a * (YY.reshape(-1, 1) - y0[None, :]) ** 2))
    return dict(nAct=nAct, validAct=valid, gx=g, gy=g.copy(), modes=modes)


# --------------------------------------------------------------------------------------
# Shack-Hartmann (diffractive, single wavefront)            (OOPAO/ShackHartmann.py)
# --------------------------------------------------------------------------------------
class OracleSH:
    def __init__(self, n_subap, resolution, D, pupil, wavelength, flux_map, light_ratio=0.5,
                 threshold_cog=0.01):
        self.nSubap, self.R, self.pupil = n_subap, resolution, pupil
        self.wavelength = wavelength
        self.p = p = resolution // n_subap                      # n_pix_subap (:154)
        self.n = n = 2 * p                                      # zero-padded lenslet (:161)
        self.thr = threshold_cog
        self.cam_res = n_subap * p
        xx, yy = np.meshgrid(np.arange(n, dtype=float), np.arange(n, dtype=float))
        self.phasor = np.exp(-(1j * np.pi * (n + 1) / n) * (xx + yy))       # :208-209
        self.flux_tiles = self._tiles(flux_map)                 # initialize_flux :327-338
        pps = self.flux_tiles.sum(axis=(1, 2))
        self.valid_1d = pps >= light_ratio * pps.max()          # :227-229
        self.valid_2d = self.valid_1d.reshape(n_subap, n_subap)
        self.nValid = int(self.valid_1d.sum())
        self.nSignal = 2 * self.nValid
        self.valid_slopes_maps = np.concatenate((self.valid_2d, self.valid_2d))
        self.vx, self.vy = np.where(self.valid_2d)
        self.SX = np.zeros((n_subap, n_subap))
        self.SY = np.zeros((n_subap, n_subap))
        self.reference_slopes_maps = np.zeros((2 * n_subap, n_subap))
        self.slopes_units = 1.0
        self.frame = np.zeros((self.cam_res, self.cam_res))
        self.cam = None                                         # wfs.cam: a Detector, or None = ideal (set after the calibration)
        self._calibrate(D)

    def _tiles(self, img):
        """tile k = i*nSubap + j holds img.T[6j:6j+6, 6i:6i+6]   (get_lenslet_em_field :340-347)."""
        s, p = self.nSubap, self.p
        t = img.T.reshape(s, p, s, p)                           # [j, a, i, b]
        return np.ascontiguousarray(t.transpose(2, 0, 1, 3)).reshape(s * s, p, p)

    def spots(self, phase):
        """|FFT2|^2/n^2 of the zero-padded lenslet fields, 2x2 binned: [nSubap^2, p, p] (:539-565)."""
        p, n = self.p, self.n
        em = np.zeros((self.nSubap ** 2, n, n), dtype=complex)
        lo = n // 2 - p // 2
        em[:, lo:lo + p, lo:lo + p] = np.sqrt(self.flux_tiles) * np.exp(1j * self._tiles(phase))
        em *= self.phasor[None]
        I = np.abs(np.fft.fft2(em, axes=(1, 2)) / n) ** 2
        return I.reshape(-1, p, 2, p, 2).sum(axis=(2, 4))

    def measure(self, phase, group_max=None):
        I = self.spots(phase)[self.valid_1d]
        s, p = self.nSubap, self.p
        frame = np.zeros((self.cam_res, self.cam_res))
        blocks = np.zeros((s * s, p, p))
        blocks[self.valid_1d] = I
        frame[:] = blocks.reshape(s, s, p, p).transpose(0, 2, 1, 3).reshape(s * p, s * p)   # :349-353
        if self.cam is not None:                                # self*self.cam then split_camera_frame (:576, :355-362)
            frame = self.cam.integrate(frame)
            I = frame.reshape(s, p, s, p).transpose(0, 2, 1, 3).reshape(s * s, p, p)[self.valid_1d]
        self.frame = frame                                      # noise-free detector == identity
        im = I.copy()
        mx = im.max() if group_max is None else group_max
        self.last_max = im.max()
        im[im < self.thr * mx] = 0                              # centroid :314-324
        norma = im.sum(axis=(1, 2))
        u = np.arange(p, dtype=float)
        with np.errstate(invalid="ignore", divide="ignore"):
            c0 = (im * u[None, :, None]).sum(axis=(1, 2)) / norma
            c1 = (im * u[None, None, :]).sum(axis=(1, 2)) / norma
        c0[~np.isfinite(c0)] = 0                                # :583-592
        c1[~np.isfinite(c1)] = 0
        self.SX[self.vx, self.vy] = c0
        self.SY[self.vx, self.vy] = c1
        s2d = np.concatenate((self.SX, self.SY)) - self.reference_slopes_maps
        s2d[~self.valid_slopes_maps] = 0
        self.signal_2D = s2d / self.slopes_units                # :598-601
        self.signal = self.signal_2D[self.valid_slopes_maps]
        return self.signal

    def _calibrate(self, D):
        """initialize_wfs :254-312 (reference slopes, then slope units from a 5-point tip ramp)."""
        R = self.R
        self.measure(np.zeros((R, R)))
        self.reference_slopes_maps = self.signal_2D.copy()
        tip, _ = np.meshgrid(np.linspace(0, np.pi, R, endpoint=False), np.linspace(0, np.pi, R, endpoint=False))
        # quirk kept: the reference indexes Tip with the *integer* 0/1 pupil array (fancy indexing of
        # rows 0 and 1, Telescope.py:392 + ShackHartmann.py:290), so the normalisation is the std of
        # one full row of the ramp, not the std over the pupil pixels.
        tip = tip * (1 / np.std(tip[self.pupil.astype(int)]))
        amp = 10e-9
        mean_slope = np.zeros(5)
        for i in range(5):
            opd = self.pupil * tip * (i - 2) * amp
            self.measure(opd * TWO_PI / self.wavelength)
            mean_slope[i] = np.mean(self.signal[:self.nValid])
        pfit = np.polyfit(np.linspace(-2, 2, 5) * amp, mean_slope, deg=1)
        self.slopes_units = np.abs(pfit[0]) * (self.wavelength / 2 / np.pi)
        self.tip_unit = tip


# --------------------------------------------------------------------------------------
# Pyramid WFS (single wavefront)                                   (OOPAO/Pyramid.py)
# --------------------------------------------------------------------------------------
class OraclePyramid:
    """Fourier-filtering pyramid sensor: zero-padded FFT -> 4-facet phase mask -> inverse FFT -> |.|^2, summed over
    the modulation points, binned to the camera, quadrant slopes maps.  Quirks kept: the mask is stored as
    complex64 (:323) and the modulation tip/tilt buffer as float32 (:964-970)."""

    def __init__(self, n_subap, resolution, pupil, wavelength, flux_map, modulation=0.0, light_ratio=0.1,
                 n_pix_separation=4, n_pix_edge=2, psf_centering=True, post_processing="slopesMaps_incidence_flux",
                 calib_modulation=50):
        self.nSubap, self.R, self.pupil = n_subap, resolution, pupil
        self.pupil_f = pupil.astype(float)
        self.flux_map = flux_map
        self.psfCentering = psf_centering
        self.postProcessing = post_processing
        self.n_pix_separation, self.n_pix_edge = n_pix_separation, n_pix_edge
        R = resolution
        self.nRes = int((n_subap * 2 + n_pix_separation + n_pix_edge * 2) * R / n_subap)          # :251
        self.zeroPaddingFactor = self.nRes / R
        self.cam_res = round(n_subap * self.zeroPaddingFactor)                                      # :255
        self.center = self.nRes // 2
        self.calibModulation = R / 2 - 1 if calib_modulation >= R / 2 else calib_modulation        # :259-262
        tip, tilt = np.meshgrid(np.linspace(-np.pi, np.pi, R), np.linspace(-np.pi, np.pi, R))       # :288-290
        self.Tilt = tilt * self.pupil_f
        self.Tip = tip * self.pupil_f
        n = self.nRes
        xx, yy = np.meshgrid(np.linspace(0, n - 1, n), np.linspace(0, n - 1, n))
        self.phasor = np.exp(-(1j * np.pi * (n + 1) / n) * (xx + yy))                               # :293-294
        self.m = self.phase_mask()
        self.mask = np.complex64(np.exp(1j * self.m))                                               # :323
        self.referenceSignal_2D = 0
        self.slopesUnits = 1
        self.isInitialized = False
        # valid-pixel selection at the (large) calibration modulation, then reference slopes at the user modulation
        self.set_modulation(self.calibModulation)
        self.measure(np.zeros((R, R)), process=False)                                               # :408-415
        I1, I2, I3, I4 = (self.grab_quadrant(k) for k in (1, 2, 3, 4))
        self.I4Q = I1 + I2 + I3 + I4
        self.validI4Q = self.I4Q >= light_ratio * self.I4Q.max()                                     # :426-430
        self.validSignal = np.concatenate((self.validI4Q, self.validI4Q))
        self.nSignal = int(np.sum(self.validSignal))
        self.isInitialized = True
        self.set_modulation(modulation)
        # wfs_calibration :454-466: OPD = pupil (1 m of piston inside the pupil)
        self.measure(self.pupil_f * TWO_PI / wavelength, process=False)
        self.referenceSignal_2D, _ = self.signal_processing()
        self.measure(np.zeros((R, R)))                                                              # :313-314

    def phase_mask(self):
        """get_phase_mask :368-405 with sx = sy = 0."""
        n_tot = self.nRes
        norma = (self.nSubap + self.n_pix_separation) * (self.R / self.nSubap)
        m = np.zeros([n_tot, n_tot])
        h = n_tot // 2
        if self.psfCentering:
            lim = np.pi / 4
            d_pix = np.pi / 4 / h
            lim = lim - d_pix
            Tip, Tilt = np.meshgrid(np.linspace(-lim, lim, h), np.linspace(-lim, lim, h))
            m[:h, :h] = Tip * norma + Tilt * norma
            m[:h, -h:] = -Tip * norma + Tilt * norma
            m[-h:, -h:] = -Tip * norma + -Tilt * norma
            m[-h:, :h] = Tip * norma + -Tilt * norma
        else:
            d_pix = (np.pi / 4) / (n_tot / 2)
            lim_p = np.pi / 4
            lim_m = np.pi / 4 - 2 * d_pix
            Tip_1, Tilt_1 = np.meshgrid(np.linspace(-lim_p, lim_p, h + 1), np.linspace(-lim_p, lim_p, h + 1))
            Tip_2, Tilt_2 = np.meshgrid(np.linspace(-lim_p, lim_p, h + 1), np.linspace(-lim_m, lim_m, h - 1))
            Tip_3, Tilt_3 = np.meshgrid(np.linspace(-lim_m, lim_m, h - 1), np.linspace(-lim_m, lim_m, h - 1))
            Tip_4, Tilt_4 = np.meshgrid(np.linspace(-lim_m, lim_m, h - 1), np.linspace(-lim_p, lim_p, h + 1))
            m[:h + 1, :h + 1] = Tip_1 * norma + Tilt_1 * norma
            m[:h + 1, -h + 1:] = -Tip_4 * norma + Tilt_4 * norma
            m[-h + 1:, -h + 1:] = -Tip_3 * norma + -Tilt_3 * norma
            m[-h + 1:, :h + 1] = Tip_2 * norma + -Tilt_2 * norma
        return -m

    def set_modulation(self, val):
        """modulation setter :941-976."""
        self.modulation = val
        if val != 0:
            perimeter = np.pi * 2 * val
            self.nTheta = 4 * int(0 + np.ceil(perimeter / 4))
            theta = np.linspace(0, 2 * np.pi, self.nTheta, endpoint=False)
            buf = np.zeros([self.nTheta, self.R, self.R]).astype(np.float32)
            for i in range(self.nTheta):
                buf[i] = (val * np.cos(theta[i]) * self.Tip + val * np.sin(theta[i]) * self.Tilt) * self.pupil_f
            self.tt_buffer = buf
        else:
            self.nTheta = 1
            self.tt_buffer = None

    def transform(self, phase):
        """pyramid_transform :469-504."""
        n, R, c = self.nRes, self.R, self.center
        support = np.zeros((n, n), dtype=complex)
        support[c - R // 2:c + R // 2, c - R // 2:c + R // 2] = self.maskAmplitude * np.exp(1j * phase)
        if self.psfCentering:
            ft = np.fft.fft2(support * self.phasor)
        else:
            ft = np.fft.fftshift(np.fft.fft2(support))
        return np.abs(np.fft.ifft2(ft * self.mask)) ** 2

    def measure(self, phase, process=True):
        self.maskAmplitude = np.sqrt(self.flux_map / self.nTheta) * self.pupil_f                    # :520
        if self.modulation == 0:
            frame = self.transform(phase)
        else:
            frame = np.zeros((self.nRes, self.nRes))
            for i in range(self.nTheta):
                frame = frame + self.transform(phase + self.tt_buffer[i])
        self.pyramidFrame = frame
        b = self.nRes // self.cam_res
        self.frame = frame.reshape(self.cam_res, b, self.cam_res, b).sum(-1).sum(1)                 # set_binning, :999
        if process and self.isInitialized:
            self.signal_2D, self.signal = self.signal_processing()
            return self.signal

    def grab_quadrant(self, n):
        """grabQuadrant :774-790 (binning 1, no rooftop)."""
        ne = int(np.round((self.n_pix_separation / self.nSubap) * self.R / (self.R / self.nSubap) / 2))
        c = int(np.round(self.cam_res / 2))
        p = int(np.ceil(self.nSubap))
        f = self.frame
        if n == 3:
            return f[ne + c:ne + c + p, ne + c:ne + c + p]
        if n == 4:
            return f[ne + c:ne + c + p, -ne + c - p:-ne + c]
        if n == 1:
            return f[-ne + c - p:-ne + c, -ne + c - p:-ne + c]
        return f[-ne + c - p:-ne + c, ne + c:ne + c + p]

    def signal_processing(self):
        """signalProcessing :682-725."""
        I1, I2, I3, I4 = (self.grab_quadrant(k) * self.validI4Q for k in (1, 2, 3, 4))
        if self.postProcessing == "slopesMaps":
            norma = np.mean((I1 + I2 + I3 + I4)[self.validI4Q])
        else:
            norma = np.float64(self.frame.mean())
        Sx = I1 - I2 + I4 - I3
        Sy = I1 - I4 + I2 - I3
        maps = (np.concatenate((Sx, Sy)) / norma - self.referenceSignal_2D) * self.slopesUnits
        return maps, maps[np.where(self.validSignal == 1)]



# --------------------------------------------------------------------------------------
# Science-path PSF                                          (OOPAO/Telescope.py:258-357)
# --------------------------------------------------------------------------------------
def telescope_psf(pupil: np.ndarray, flux_map: np.ndarray, phase: np.ndarray, zero_padding: int = 2) -> np.ndarray:
    """tel.computePSF(zeroPaddingFactor) with no detector and no spatial filter (img_resolution = zeroPaddingFactor *
    resolution): PropagateField (:296-351) of E = pupil * sqrt(src.fluxMap) * exp(i phase).  Kept quirk: the parity rule
    ``if oversampling % 2 != img_resolution % 2: oversampling += 1`` (:303-305) turns the default oversampling 1 into 2 for
    every even image size, so the transform runs at N = 2 * zeroPaddingFactor * resolution and the PSF is its 2 x 2
    sum-binned |.|^2.  The phasor is complex64 in the reference (:316)."""
    R = pupil.shape[0]
    img_resolution = zero_padding * R
    oversampling = 1
    if zero_padding * oversampling < 2:
        oversampling = int(np.ceil(2.0 / zero_padding))                                      # :299-300
    if oversampling % 2 != img_resolution % 2:
        oversampling += 1                                                                    # :303-305
    img_size = int(np.ceil(img_resolution * oversampling))
    N = int(np.fix(zero_padding * oversampling * R))
    pad = int(np.ceil((N - R) / 2))                                                          # :308
    amp = pupil * np.sqrt(flux_map)                                                          # :278 (pupilReflectivity = 1)
    sup = np.pad(amp * np.exp(1j * phase), ((pad, pad), (pad, pad)), constant_values=0)      # :310-311
    N = sup.shape[0]
    xx, yy = np.meshgrid(np.linspace(0, N - 1, N), np.linspace(0, N - 1, N), copy=False)
    phasor = np.exp(-1j * np.pi / N * (xx + yy) * (1 - img_resolution % 2)).astype(np.complex64)   # :315-316
    emf = np.fft.fftshift(1 / N * np.fft.fft2(np.fft.ifftshift(sup * phasor)))               # :319
    shift_pix = 0 if N % 2 == img_size % 2 else (1 if N % 2 == 0 else -1)                    # :322-328
    lo = int(np.ceil(N / 2) - img_size // 2 + (1 - N % 2) - 1)
    hi = int(np.ceil(N / 2) + img_size // 2 + shift_pix)                                     # :333-336
    emf = emf[lo:hi, lo:hi]
    psf = np.abs(emf) ** 2
    if oversampling != 1:                                                                    # :341-343, tools.set_binning (sum)
        m = psf.shape[0] // oversampling
        psf = psf.reshape(m, oversampling, m, oversampling).sum(-1).sum(1)
    return psf


# --------------------------------------------------------------------------------------
# Detector                                                   (OOPAO/Detector.py:178-301)
# --------------------------------------------------------------------------------------
class Detector:
    """WFS camera: photon noise, quantum efficiency, dark shot noise, saturation, read-out noise, ADC.

    ``integrate`` + ``readout`` of the reference for one frame per read-out (integrationTime equal to the loop's
    samplingTime or None), binning 1, no background map.  The reference seeds its four RandomStates from the wall
    clock (Detector.py:127-130), so a noisy frame is reproducible only in distribution; here the seed is explicit.
    With every noise source off the path is deterministic (QE, saturation clip, ADC truncation) and bit-comparable."""

    def __init__(self, photonNoise=False, readoutNoise=0.0, QE=1.0, darkCurrent=0.0, integrationTime=None, FWC=None,
                 bits=None, gain=1, sensor="CCD", seed=0):
        if sensor not in ("EMCCD", "CCD", "CMOS"):
            raise ValueError("Sensor must be 'EMCCD', 'CCD', or 'CMOS'")                   # :40-41
        self.photonNoise, self.readoutNoise, self.QE = photonNoise, readoutNoise, QE
        self.darkCurrent, self.integrationTime, self.FWC = darkCurrent, integrationTime, FWC
        self.bits, self.gain, self.sensor = bits, gain, sensor
        self.rs_photon, self.rs_readout, self.rs_dark = RandomState(seed), RandomState(seed + 1), RandomState(seed + 2)

    def integrate(self, frame: np.ndarray) -> np.ndarray:
        frame = np.array(frame, dtype=np.float64)
        if self.photonNoise:                                                               # :204-206, :285-286
            frame = self.rs_photon.poisson(frame).astype(np.float64)
        frame = frame * self.QE                                                            # :178-180
        if self.darkCurrent != 0:                                                          # :224-229, :235-236
            frame = frame + self.rs_dark.poisson(np.ones(frame.shape) * (self.darkCurrent * self.integrationTime))
        if self.FWC is not None:                                                           # :183-187
            frame = np.clip(frame, 0, self.FWC)
        if self.sensor == "EMCCD":                                                         # :243-244
            frame = frame * self.gain
        if self.readoutNoise != 0:                                                         # :218-221
            frame = frame + np.round(self.rs_readout.randn(*frame.shape) * self.readoutNoise).astype(int)
        if self.sensor in ("CCD", "CMOS"):                                                 # :258-259
            frame = frame * self.gain
        if self.bits is not None:                                                          # :190-201
            if self.FWC is None:
                frame = (frame / frame.max() * 2 ** self.bits).astype(np.int64)
            else:
                frame = (frame / self.FWC * (2 ** self.bits - 1)).astype(np.int64)          # truncation toward zero
                frame = np.clip(frame, frame.min(), 2 ** self.bits - 1)
            frame = frame.astype(np.float64)
        return frame

# --------------------------------------------------------------------------------------
# Zernike basis (Noll), as the reference builds it          (OOPAO/Zernike.py:26-66)
# aotools 1.0.6 (third party, absent) supplies zernIndex / zernikeRadialFunc: restated here.
# --------------------------------------------------------------------------------------
def noll_to_nm(j: int):
    n = int((-1.0 + np.sqrt(8 * (j - 1) + 1)) / 2.0)
    p = j - (n * (n + 1)) / 2.0
    k = n % 2
    m = int((p + k) / 2.0) * 2 - k
    if m != 0:
        if j % 2 == 0:
            s = 1
        else:
            s = -1
        m *= s
    return n, m


def zernike_radial(n: int, m: int, r: np.ndarray):
    out = np.zeros(r.shape)
    for i in range(0, int((n - m) / 2) + 1):
        out += r ** (n - 2.0 * i) * (((-1) ** i) * math.factorial(n - i)) / (
            math.factorial(i) * math.factorial(int(0.5 * (n + m) - i)) * math.factorial(int(0.5 * (n - m) - i)))
    return out


def zernike_modes(pupil: np.ndarray, D: float, n_modes: int) -> np.ndarray:
    """[pixelArea, n_modes], piston excluded, each mode mean-removed and unit-std in the pupil."""
    R = pupil.shape[0]
    X, Y = np.where(pupil > 0)
    X = (X - (R + R % 2 - 1) / 2) / R * D
    Y = (Y - (R + R % 2 - 1) / 2) / R * D
    rr = np.sqrt(X ** 2 + Y ** 2)
    rr = rr / rr.max()
    th = np.arctan2(Y, X)
    out = np.zeros((X.size, n_modes))
    for i in range(1, n_modes + 1):
        n, m = noll_to_nm(i + 1)
        if m == 0:
            Z = np.sqrt(n + 1) * zernike_radial(n, 0, rr)
        elif m > 0:
            Z = np.sqrt(2 * (n + 1)) * zernike_radial(n, m, rr) * np.cos(m * th)
        else:
            m = abs(m)
            Z = np.sqrt(2 * (n + 1)) * zernike_radial(n, m, rr) * np.sin(m * th)
        Z = Z - Z.mean()
        Z = Z * (1 / np.std(Z))
        out[:, i - 1] = Z
    return out


# --------------------------------------------------------------------------------------
# Calibration                     (OOPAO/calibration/InteractionMatrix.py:13-135, CalibrationVault.py:19-30)
# --------------------------------------------------------------------------------------
def calibration_vault_M(D: np.ndarray) -> np.ndarray:
    U, s, V = np.linalg.svd(D, full_matrices=False)
    return V.T @ np.diag(1 / s) @ U.T


# --------------------------------------------------------------------------------------
# The environment (Papyrus-style OOPAO gym env with a Shack-Hartmann WFS)
#   MAIN/OOPAOEnv/OOPAOEnv.py:93-385 (set_params), :485-536 (step), :82-86 (reset_soft)
# --------------------------------------------------------------------------------------
class OracleEnv:
    def __init__(self, resolution=120, diameter=8.0, n_subap=20, dt=1 / 500, band="I", magnitude=8.0,
                 r0=0.13, L0=30.0, windSpeed=(10.0,), windDirection=(72.0,), fractionalR0=(1.0,),
                 altitude=(0.0,), mech_coupling=0.35, m2c=None, n_modes=50, light_ratio=None,
                 threshold_cog=0.01, nLoop=10000, leak=0.99, gainCL=0.5, n_meas=6, wfs_type="sh", modulation=0.0,
                 psf_centering=True, second_dm_nsub=None, modal_cm=None, dm_dense=True, geom_AB=None, fov_arcsec=0.0):
        self.R, self.D, self.dt = resolution, diameter, dt
        self.leak, self.gainCL = leak, gainCL
        self.pupil = make_pupil(resolution)
        self.wavelength, self.nPhoton = source_photometry(band, magnitude)
        self.flux_map = self.pupil.astype(float) * self.nPhoton * dt * (diameter / resolution) ** 2
        self.atm = OracleAtmosphere(resolution, diameter, dt, self.pupil, r0, L0, windSpeed, fractionalR0,
                                    windDirection, altitude, geom_AB=geom_AB,
                                    fov_rad=fov_arcsec / 206265.0)    # tel.fov_rad = fov / 206265 (OOPAO/Telescope.py)
        self.nActuator = n_subap + 1
        # dm_dense = False (ELT size: dm.modes would be 9.6 GB): the DM surface through the separable factors,
        # OPD = gy C gx^T with C the command image -- equal to modes @ coefs to rounding (tests/test_oracle_golden.py)
        dm = dm_geometry(resolution, diameter, n_subap, mech_coupling, pitch=diameter / self.nActuator, dense=dm_dense)
        self.dm_modes = dm["modes"]
        self.dm_valid_flat = np.flatnonzero(dm["validAct"])
        if not dm_dense and (second_dm_nsub is not None or m2c is None):
            raise ValueError("dm_dense=False: one DM, and the mode-to-command matrix handed over")
        self.gx, self.gy = dm["gx"], dm["gy"]
        self.dm_mask = dm["validAct"].reshape(self.nActuator, self.nActuator)
        self.xvalid, self.yvalid = np.nonzero(self.dm_mask)
        self.nValidAct = int(dm["validAct"].sum())
        self.dm_blocks = [(0, self.nValidAct)]                   # every DM is calibrated by its own InteractionMatrix call
        if second_dm_nsub is not None:
            # a second DM chained behind the first, tel*dm1*dm2 (Telescope.py:533-544: every DM adds its OPD): one stacked
            # command vector [dm1 | dm2]; the actuator "image" holds both grids, dm1 top-left and dm2 bottom-right
            n1, n2 = self.nActuator, second_dm_nsub + 1
            dm2 = dm_geometry(resolution, diameter, second_dm_nsub, mech_coupling, pitch=diameter / n2)
            self.dm_modes = np.hstack([dm["modes"], dm2["modes"]])
            self.nActuator = n1 + n2
            mask = np.zeros((n1 + n2, n1 + n2), bool)
            mask[:n1, :n1] = dm["validAct"].reshape(n1, n1)
            mask[n1:, n1:] = dm2["validAct"].reshape(n2, n2)
            self.dm_mask = mask
            self.xvalid, self.yvalid = np.nonzero(mask)
            self.dm_blocks = [(0, self.nValidAct), (self.nValidAct, int(mask.sum()))]
            self.nValidAct = int(mask.sum())
            self.gx = self.gy = None
        self.wfs_type = wfs_type
        if wfs_type == "sh":
            self.wfs = OracleSH(n_subap, resolution, diameter, self.pupil, self.wavelength, self.flux_map,
                                0.5 if light_ratio is None else light_ratio, threshold_cog)
        else:                                                    # MAIN/OOPAOEnv/OOPAOEnv.py:239-246
            self.wfs = OraclePyramid(n_subap, resolution, self.pupil, self.wavelength, self.flux_map,
                                     modulation=modulation, light_ratio=0.1 if light_ratio is None else light_ratio,
                                     psf_centering=psf_centering)
        if m2c is None:
            Z = zernike_modes(self.pupil, diameter, n_modes)
            m2c = np.linalg.pinv(self.dm_modes[self.pupil.reshape(-1)]) @ Z
        self.M2C = m2c[:, :n_modes]
        self.n_meas = n_meas
        if modal_cm is None:
            self.imat = self.interaction_matrix(self.wavelength / 16, n_meas)
            Mmod = calibration_vault_M(self.imat @ self.M2C)
        else:                                                   # large geometries: the modal command matrix calib.M is handed
            self.imat = None                                    # over (the tests pin it, and a few pokes, separately)
            Mmod = np.asarray(modal_cm, dtype=float)
        self.modal_cm = Mmod
        self.reconstructor = self.M2C @ Mmod                    # OOPAOEnv.py:381
        self.F = self.M2C @ np.linalg.pinv(self.M2C)            # :383
        self.coefs = np.zeros(self.nValidAct)
        self.dm_prev = np.zeros(self.nValidAct)                 # OOPAOEnv.py:314: set once here, then only by step()
        self.total = np.zeros(nLoop)
        self.residual = np.zeros(nLoop)
        self.SR = []
        self.tel_OPD = np.zeros((self.R, self.R))
        self.measure()                                          # flat wavefront: ngs*tel*dm*wfs, :312-315
        self.atm.generate_new_phase_screen(10)                  # :320  (tel.OPD <- atm.OPD, no measurement)
        self.tel_OPD = self.atm.OPD.copy()

    # -- helpers --------------------------------------------------------------------
    def dm_opd(self, coefs):
        if self.dm_modes is None:                               # separable form of the same product (dm_dense=False)
            n1 = self.nActuator
            C = np.zeros(n1 * n1)
            C[self.dm_valid_flat] = coefs
            return self.gy @ C.reshape(n1, n1) @ self.gx.T
        return (self.dm_modes @ coefs).reshape(self.R, self.R)  # DeformableMirror.py:556

    def vec_to_img(self, v):
        img = np.zeros((self.nActuator, self.nActuator))
        img[self.xvalid, self.yvalid] = v
        return img

    def img_to_vec(self, a):
        return a[self.xvalid, self.yvalid]

    def poke_signal(self, a, stroke):
        """wfs.signal / stroke of one poked actuator on its own (Pyramid, or an SH measurement that shares no threshold)."""
        cf = np.zeros(self.nValidAct)
        cf[a] = stroke
        return self.wfs.measure(self.dm_opd(cf) * self.pupil * TWO_PI / self.wavelength) / stroke

    def interaction_matrix(self, stroke, n_meas):
        """Zonal push-only matrix D = signal(poke)/stroke (InteractionMatrix.py:13-135, single_pass), one call per DM: the
        pokes of a call are measured in batches of n_meas that share the centroid-threshold maximum (multi-wavefront
        branch, ShackHartmann.py:605-672); the last batch of a call holds its n % n_meas last actuators (:72-78)."""
        nA = self.nValidAct
        D = np.zeros((self.wfs.nSignal, nA))
        if self.wfs_type != "sh":                                # Pyramid: the batched measurements are independent
            for a in range(nA):
                D[:, a] = self.poke_signal(a, stroke)
            return D
        for lo, hi in self.dm_blocks:
            n = hi - lo
            n_cycle = int(np.ceil(n / n_meas))
            n_extra = n % n_meas
            for c in range(n_cycle):
                if c == n_cycle - 1 and n_extra != 0:
                    idx = list(range(hi - n_extra, hi))
                else:
                    idx = list(range(lo + c * n_meas, lo + (c + 1) * n_meas))
                phases = []
                for a in idx:
                    cf = np.zeros(nA)
                    cf[a] = stroke
                    phases.append(self.dm_opd(cf) * TWO_PI / self.wavelength)   # phase_no_pupil
                gmax = max(self.wfs.spots(ph)[self.wfs.valid_1d].max() for ph in phases)
                for a, ph in zip(idx, phases):
                    D[:, a] = self.wfs.measure(ph, group_max=gmax) / stroke
        return D

    def measure(self):
        self.phase = self.tel_OPD * TWO_PI / self.wavelength    # Telescope.py:404-412
        return self.wfs.measure(self.phase)

    # -- reference-visible surface -------------------------------------------------
    def new_episode(self, seed):
        """mbrl.py:49-52: new screens, flat DM, one WFS measurement."""
        self.atm.generate_new_phase_screen(seed)
        self.coefs = np.zeros(self.nValidAct)
        self.tel_OPD = (self.atm.OPD_no_pupil + self.dm_opd(self.coefs)) * self.pupil
        self.measure()

    def reset_soft(self):
        return self.vec_to_img(-self.reconstructor @ self.wfs.signal) * 1e6

    def step(self, i, action):
        a = self.img_to_vec(np.asarray(action)) * 1e-6
        self.atm.update()
        self.total[i] = np.std(self.atm.OPD[self.pupil]) * 1e9
        self.tel_OPD = (self.atm.OPD_no_pupil + self.dm_opd(self.coefs)) * self.pupil
        self.measure()
        self.coefs = self.dm_prev * self.leak + a               # OOPAOEnv.py:508-509: dm_prev, not dm.coefs
        self.dm_prev = self.coefs.copy()
        obs = self.vec_to_img(-self.reconstructor @ self.wfs.signal) * 1e6
        self.residual[i] = np.std(self.tel_OPD[self.pupil]) * 1e9
        strehl = np.exp(-np.var(self.phase[self.pupil]))
        self.SR.append(strehl)
        return obs, self.wfs.frame, -np.linalg.norm(obs), strehl, False, {"strehl": strehl}
