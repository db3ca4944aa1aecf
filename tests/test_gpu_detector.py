"""GPU: the WFS camera model (rlao_amd/csrc/detector.hpp) against the oracle's restatement of OOPAO/Detector.py.

The reference seeds its noise generators from the wall clock (Detector.py:127-130), so noisy frames are comparable in
distribution only; the deterministic part (QE, saturation, ADC) is compared count for count."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SMALL = dict(diameter=3.2, nSubaperture=8, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
             fractionalR0=[1.0], altitude=[0.0], nModes=20, nLoop=64, magnitude=6.0)


def _env(n, **kw):
    from rlao_amd.env import BatchedAOEnv
    env = BatchedAOEnv(n_envs=n, device=0, dtype="f32", **kw)
    env.set_params(SMALL, camera="ideal", wfs_type="shackhartmann")
    env.env_seed_stride = 0                       # the same atmosphere in every env: frames differ by noise only
    env.generate_new_phase_screen(7)
    env.dm.coefs = 0
    return env


def _frames(env):
    from rlao_amd import _lib as L
    env.measure()
    return env._shard.download(L.B_FRAME, (env.n_envs, env.cam_res, env.cam_res)).astype(np.float64)


def test_adc_and_saturation_are_exact():
    from oracle import ao_oracle as O             # checker only
    env = _env(2)
    ideal = _frames(env)
    fwc = float(ideal.max()) * 0.4                 # QE 0.56: the brightest spots saturate
    env.wfs.cam.QE, env.wfs.cam.FWC, env.wfs.cam.bits, env.wfs.cam.sensor = 0.56, fwc, 10, "CMOS"
    got = _frames(env)
    want = O.Detector(QE=0.56, FWC=fwc, bits=10, sensor="CMOS").integrate(ideal)
    diff = np.abs(got - want)
    assert diff.max() <= 1                          # float32 vs float64 only at a truncation boundary ...
    assert (diff > 0).mean() < 2e-3                 # ... on a handful of pixels
    assert got.max() == 1023 and (got == np.floor(got)).all()
    env.close()


def test_photon_noise_is_poisson_and_reproducible():
    env = _env(256)
    ideal = _frames(env)[0]
    env.wfs.cam.photonNoise = True
    a = _frames(env)
    lit = ideal > 0.05 * ideal.max()
    mean, var = a.mean(axis=0), a.var(axis=0)
    # per-pixel mean and variance over 256 draws: both equal lambda for a Poisson variable
    z = (mean[lit] - ideal[lit]) / np.sqrt(ideal[lit] / 256)
    assert abs(z.mean()) < 0.2 and 0.85 < z.std() < 1.15
    ratio = var[lit] / ideal[lit]
    assert abs(ratio.mean() - 1) < 0.03
    assert (a == np.floor(a)).all() and a.min() >= 0
    # faint pixels too: mean of the whole frame
    assert abs(a.mean() - ideal.mean()) < 5 * np.sqrt(ideal.mean() / a.size) + 1e-3 * ideal.mean()
    # every measurement is a new frame of the noise stream; a new env with the same seed replays the same stream
    b = _frames(env)
    assert (a != b).any()
    env2 = _env(256)
    env2.wfs.cam.photonNoise = True
    np.testing.assert_array_equal(_frames(env2), a)
    env2.close()
    # the stream of an env does not depend on where it sits in a batch
    env3 = _env(4, env_index_offset=100)
    env3.wfs.cam.photonNoise = True
    np.testing.assert_array_equal(_frames(env3), a[100:104])
    env3.close()
    env.close()


def test_photon_counts_follow_the_poisson_law_in_every_brightness_class():
    """Distribution test of the photon draw, per brightness class: for X ~ Poisson(lam) and V ~ U(0, 1) independent,
    F(X - 1) + V p(X) is uniform on (0, 1) (randomised probability-integral transform), whatever lam each pixel has.  Pooled over
    the pixels of a class x 256 envs x 4 frames: chi-square of a 64-bin histogram and the Kolmogorov-Smirnov distance.  Every class
    goes through the alias sampler of poisson_alias.hpp (below 32 photons: fine table + remainder; above: + the coarse table),
    exact to the 2^-23 quantisation of its thresholds, so no approximation error is allowed for.  (tests/test_poisson_alias.py
    drives the sampler directly, incl. the PTRS hand-over above 1024 photons.)"""
    from scipy import stats
    env = _env(256)
    ideal = _frames(env)[0]
    env.wfs.cam.photonNoise = True
    a = np.stack([_frames(env) for _ in range(4)])               # [4, 256, cam, cam]
    env.close()
    rs = np.random.RandomState(3)
    tested = []
    for lo, hi in ((0.02, 1.0), (1.0, 10.0), (10.0, 30.0), (30.0, 1e9)):
        sel = (ideal >= lo) & (ideal < hi)
        if sel.sum() < 8:
            continue
        lam = ideal[sel]
        x = a[:, :, sel].reshape(-1, lam.size)
        u = stats.poisson.cdf(x - 1, lam) + rs.uniform(size=x.shape) * stats.poisson.pmf(x, lam)
        n = u.size
        h = np.histogram(u, bins=64, range=(0, 1))[0]
        chi2 = float(((h - n / 64) ** 2 / (n / 64)).sum())
        ks = float(stats.kstest(u.ravel(), "uniform").statistic) * np.sqrt(n)
        assert chi2 < stats.chi2.ppf(1 - 1e-6, 63), (lo, hi, chi2)          # 63 dof: 99.9999 % point = 137
        assert ks < 2.2, (lo, hi, ks)                                       # P(sqrt(n) D > 2.2) = 1.2e-4
        tested.append((lo, hi, int(sel.sum()), chi2, ks))
    assert any(hi <= 10 for _, hi, *_ in tested) and any(lo >= 10 for lo, *_ in tested), tested


def test_razor_camera_moments_match_oracle():
    """Razor settings (MAIN/OOPAOEnv/OOPAOEnvRazor.py:243-250, 333): photon + dark + read-out noise, QE, FWC, 10 bits."""
    from oracle import ao_oracle as O             # checker only
    env = _env(256)
    ideal = _frames(env)[0]
    cam = env.wfs.cam
    cam.sensor, cam.FWC, cam.bits, cam.QE, cam.darkCurrent, cam.integrationTime = "CMOS", 10000, 10, 0.56, 5, 1 / 500
    cam.photonNoise, cam.readoutNoise = True, 14
    got = _frames(env)
    det = O.Detector(photonNoise=True, readoutNoise=14, QE=0.56, darkCurrent=5, integrationTime=1 / 500, FWC=10000, bits=10,
                     sensor="CMOS", seed=3)
    want = np.stack([det.integrate(ideal) for _ in range(256)])
    # frame-wide and bright-pixel moments of the two samples agree within their sampling error
    for sel in (np.ones_like(ideal, bool), ideal > 0.3 * ideal.max()):
        g, w = got[:, sel], want[:, sel]
        se = np.sqrt((g.var() + w.var()) / g.size)
        assert abs(g.mean() - w.mean()) < 6 * se + 1e-3
        assert abs(g.std() - w.std()) < 0.05 * w.std() + 1e-3
    # and the loop still closes on the noisy frames (slopes finite, Strehl sane)
    obs = env.reset_soft()
    for i in range(5):
        obs, frame, rew, sr, done, info = env.step(i, 0.4 * obs)
    assert np.isfinite(obs.cpu().numpy()).all() and float(sr.min()) >= 0
    env.close()


@pytest.mark.parametrize("photon_only", [False, True])
def test_fused_step_kernel_and_separate_kernels_draw_the_same_noise(photon_only):
    """The camera inside the fused step kernel and the stand-alone detector kernel index the same Philox streams: with the
    same seed the two paths give the same noisy frames (up to a float32 last-bit difference in the photon count fed to the
    Poisson inversion, which may flip a rare draw) and the same closed loop."""
    import torch
    from rlao_amd import _lib as L
    outs = []
    for fused in (1, 0):
        env = _env(8)
        L.check(env._shard.lib.aoenv_set_option(env._shard.h, L.OPT_FUSED_STEP, fused))
        cam = env.wfs.cam
        if photon_only:                     # the reference envs' default camera (OOPAOEnv.py:379)
            cam.photonNoise = True
        else:
            cam.sensor, cam.FWC, cam.bits, cam.QE, cam.darkCurrent, cam.integrationTime = "CMOS", 10000, 10, 0.56, 5, 1 / 500
            cam.photonNoise, cam.readoutNoise = True, 14
        env.measure()
        obs = env.reset_soft()
        frames = []
        for i in range(6):
            obs, frame, rew, sr, done, info = env.step(i, 0.4 * obs)
            frames.append(frame.cpu().numpy().astype(np.float64))
        torch.cuda.synchronize()
        outs.append((np.stack(frames), obs.cpu().numpy()))
        env.close()
    (fa, oa), (fb, ob) = outs
    assert (fa[0] != fb[0]).mean() < 2e-3           # first step: identical inputs
    assert np.abs(fa[0] - fb[0]).max() <= 3
    assert (fa[0] > 0).any()
    if not photon_only:
        assert fa[0, 0, :6, :6].std() > 0     # corner lenslet (not valid, no light): dark + read-out noise
    assert np.abs(oa - ob).max() < 0.05 * np.abs(ob).max() + 1e-3

