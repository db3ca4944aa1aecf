"""Import the reference's OOPAO physics classes in the BUILD CONTAINER (test infrastructure only).

Used only by ``oracle/make_goldens.py`` (and ad-hoc validation here).  /root/reference does not
exist on the GPU box, so nothing in tests/, bench.py or smoke() imports this module at run time.

The reference's hot-path modules import a few third-party packages at module import time that
are absent from this image (ordinary ModuleNotFoundError, not a permission denial):

* ``jsonpickle``, ``astropy.io.fits``, ``aotools``, ``gym`` -- never executed on the step path.
  They get inert ``types.ModuleType`` placeholders so that the ``import`` statements succeed.
* ``numpy.math`` was removed in NumPy 2 (the reference pins numpy 1.23.4); it is aliased to ``math``.
* ``skimage.transform`` -- ``SimilarityTransform(translation=...)`` and ``warp(order=3)`` ARE executed
  (OOPAO/tools/tools.py:210-217).  scikit-image is third-party code that is not part of the reference;
  its documented behaviour is restated in ``oracle/ao_oracle.py::warp_translate`` and wired in here.
  That single stage is therefore NOT pinned by the reference ("parity unpinned");
  everything around it (ring extrusion A.Z + B.xi, RNG streams, footprint crop, scaling, DM, WFS,
  calibration) is executed by the reference's own code.
"""
from __future__ import annotations

import contextlib
import io
import math
import os
import sys
import types

import numpy as np

REF_ROOT = "/root/reference/drl4ao"
OOPAO_PARENT = os.path.join(REF_ROOT, "AO_OOPAO")


class _Translation:
    """Stand-in for skimage.transform.SimilarityTransform restricted to translations."""

    def __init__(self, translation=(0, 0), matrix=None):
        if matrix is not None:
            self.params = np.array(matrix, dtype=float)
        else:
            self.params = np.eye(3)
            self.params[0, 2] = translation[0]
            self.params[1, 2] = translation[1]

    @property
    def inverse(self):
        return _Translation(matrix=np.linalg.inv(self.params))


def _warp(image, inverse_map, order=3, **_kw):
    from oracle.ao_oracle import warp_translate
    assert order == 3
    m = inverse_map.params                       # output (x, y) -> input (x, y)
    assert np.allclose(m[:2, :2], np.eye(2))
    return warp_translate(np.asarray(image, dtype=float), -m[0, 2], -m[1, 2])


def install_stubs():
    if not hasattr(np, "math"):
        np.math = math
    for name in ("jsonpickle", "aotools", "gym"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    gym = sys.modules["gym"]
    if not hasattr(gym, "Env"):
        gym.Env = type("Env", (), {})
        gym.Wrapper = type("Wrapper", (), {"__init__": lambda self, env: None})
    if "astropy" not in sys.modules:
        ap = types.ModuleType("astropy")
        apio = types.ModuleType("astropy.io")
        fits = types.ModuleType("astropy.io.fits")
        ap.io = apio
        apio.fits = fits
        sys.modules.update({"astropy": ap, "astropy.io": apio, "astropy.io.fits": fits})
    if "skimage" not in sys.modules:
        sk = types.ModuleType("skimage")
        skt = types.ModuleType("skimage.transform")
        skt.SimilarityTransform = _Translation
        skt.warp = _warp
        sk.transform = skt
        sys.modules.update({"skimage": sk, "skimage.transform": skt})


@contextlib.contextmanager
def quiet():
    """The reference prints banners and property tables on every constructor."""
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield


def load():
    """Returns a namespace with the reference classes used on the hot path."""
    import matplotlib
    matplotlib.use("Agg")
    install_stubs()
    if OOPAO_PARENT not in sys.path:
        sys.path.insert(0, OOPAO_PARENT)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if repo not in sys.path:
        sys.path.insert(0, repo)
    with quiet():
        from OOPAO.Telescope import Telescope
        from OOPAO.Source import Source
        from OOPAO.Atmosphere import Atmosphere
        from OOPAO.DeformableMirror import DeformableMirror
        from OOPAO.ShackHartmann import ShackHartmann
        from OOPAO.Pyramid import Pyramid
        from OOPAO.Detector import Detector
        from OOPAO.calibration.InteractionMatrix import InteractionMatrix
        from OOPAO.calibration.CalibrationVault import CalibrationVault
    return types.SimpleNamespace(Telescope=Telescope, Source=Source, Atmosphere=Atmosphere,
                                 DeformableMirror=DeformableMirror, ShackHartmann=ShackHartmann,
                                 Pyramid=Pyramid, Detector=Detector, InteractionMatrix=InteractionMatrix,
                                 CalibrationVault=CalibrationVault)
