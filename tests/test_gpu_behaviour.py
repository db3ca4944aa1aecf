"""GPU: properties of the HIP path that the goldens do not cover -- implementation switches agree, batching is
invariant and bitwise reproducible at the full BASELINE size, the C ABI reports errors instead of crashing,
the single-env flavour returns the reference's types, and random (non-integrator) actions match the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

C2 = dict(diameter=8.0, nSubaperture=20, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
          fractionalR0=[1.0], altitude=[0.0], nModes=50, nLoop=64)
SMALL = dict(diameter=3.2, nSubaperture=8, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
             fractionalR0=[1.0], altitude=[0.0], nModes=20, nLoop=64)


def _run(env, steps, seed, gain=0.5):
    import torch
    env.generate_new_phase_screen(seed)
    env.dm.coefs = 0
    env.dm_prev = 0                                             # a fresh episode of a fresh env (the prologue alone keeps dm_prev)
    env.measure()
    obs = env.reset_soft()
    out = []
    for i in range(steps):
        obs, frame, rew, sr, done, info = env.step(i, gain * obs)
        out.append((obs.clone(), frame.clone(), rew.clone(), sr.clone()))
    torch.cuda.synchronize()
    return out


def test_switches_agree_small():
    """Specialised kernels (register-resident SH, MFMA contractions, hardware trig) vs the generic ones."""
    from rlao_amd import _lib as L
    from rlao_amd.env import BatchedAOEnv
    ref = None
    for opts in [dict(), {L.OPT_RING_LOOKAHEAD: 0}, {L.OPT_DEFER_RING: 0}, {L.OPT_FUSED_STEP: 0}, {L.OPT_FUSED_STEP: 0, L.OPT_COEFS_IMAGE: 1},
                 {L.OPT_FUSED_STEP: 0, L.OPT_COEFS_IMAGE: 1, L.OPT_MFMA_GEMM: 0}, {L.OPT_FUSED_STEP: 0, L.OPT_FAST_TRIG: 0}, {L.OPT_FAST_WFS: 0}, {L.OPT_MFMA_GEMM: 0}, {L.OPT_FAST_TRIG: 0}, {L.OPT_FUSED_TAIL: 0},
                 {L.OPT_FUSED_TAIL: 0, L.OPT_MFMA_GEMM: 0},
                 {L.OPT_FUSED_TAIL: 0, L.OPT_FACTORED_RECON: 0}, {L.OPT_FUSED_TAIL: 0, L.OPT_FACTORED_RECON: 0, L.OPT_MFMA_GEMM: 0},
                 {L.OPT_FAST_WFS: 0, L.OPT_MFMA_GEMM: 0, L.OPT_FAST_TRIG: 0, L.OPT_FUSED_TAIL: 0}]:
        env = BatchedAOEnv(n_envs=3, device=0, dtype="f32")
        env.set_params(SMALL, camera="ideal", wfs_type="shackhartmann")
        for k, v in opts.items():
            L.check(env._shard.lib.aoenv_set_option(env._shard.h, k, v))
        out = _run(env, 12, 5)
        env.close()
        if ref is None:
            ref = out
            continue
        if opts == {L.OPT_RING_LOOKAHEAD: 0}:                 # the ring computed ahead on the side stream IS the ring computed in place
            import torch
            for x, y in zip(out, ref):
                assert all(torch.equal(p, q) for p, q in zip(x, y))
        for (o, f, r, s), (o0, f0, r0, s0) in zip(out, ref):
            np.testing.assert_allclose(o.cpu().numpy(), o0.cpu().numpy(), atol=2e-5)
            np.testing.assert_allclose(f.cpu().numpy(), f0.cpu().numpy(), atol=2e-5 * float(f0.max()))
            np.testing.assert_allclose(s.cpu().numpy(), s0.cpu().numpy(), atol=1e-5)
            np.testing.assert_allclose(r.cpu().numpy(), r0.cpu().numpy(), rtol=1e-4, atol=2e-5)


def test_large_dm_row_products_agree():
    """A DM of 37 x 37 actuators (more than 32 across: the KS = 32 register blocking of the matrix-core kernels, R = 216 > 128:
    the separate phase / spots kernels).  Gy.C once per env in MFMA operand layout (k_dm_rows, AOENV_OPT_COEFS_IMAGE, the
    default above 1024 actuators) vs every tile forming its own rows vs the float64 shard, in a closed loop."""
    from rlao_amd import _lib as L
    from rlao_amd.env import BatchedAOEnv
    geo = dict(C2, nSubaperture=36, nPixelPerSubap=6, nModes=60)
    outs = {}
    for name, dtype, opt in (("rows", "f32", 1), ("tiles", "f32", 0), ("f64", "f64", 0)):
        env = BatchedAOEnv(n_envs=2, device=0, dtype=dtype)
        env.set_params(geo, camera="ideal", wfs_type="shackhartmann")
        assert env.nActuator == 37 and env.R == 216
        L.check(env._shard.lib.aoenv_set_option(env._shard.h, L.OPT_COEFS_IMAGE, opt))
        outs[name] = _run(env, 6, 11)
        env.close()
    for (o, f, r, s), (o1, f1, r1, s1), (o2, f2, r2, s2) in zip(outs["rows"], outs["tiles"], outs["f64"]):
        scale = float(o2.abs().max())
        np.testing.assert_allclose(o.cpu().numpy(), o1.cpu().numpy(), atol=2e-5 * scale)
        np.testing.assert_allclose(o.cpu().numpy(), o2.cpu().numpy(), atol=1e-2 * scale)     # float32 slopes through the reconstructor
        np.testing.assert_allclose(f.cpu().numpy(), f1.cpu().numpy(), atol=2e-5 * float(f1.max()))
        np.testing.assert_allclose(f.cpu().numpy(), f2.cpu().numpy(), atol=2e-3 * float(f2.max()))
        np.testing.assert_allclose(s.cpu().numpy(), s1.cpu().numpy(), atol=1e-5)
        np.testing.assert_allclose(s.cpu().numpy(), s2.cpu().numpy(), atol=1e-3)


def test_full_size_batch_invariance_and_determinism():
    """256 envs of the BASELINE geometry: identical seeds give bitwise identical envs wherever they sit in the
    batch, a second run reproduces every bit (no atomics-order dependence), different seeds differ."""
    import torch
    from rlao_amd.env import BatchedAOEnv
    env = BatchedAOEnv(n_envs=256, device=0, dtype="f32", env_seed_stride=0)     # every env: seed 17
    env.set_params(C2, camera="ideal", wfs_type="shackhartmann")
    a = _run(env, 8, 17)
    b = _run(env, 8, 17)
    for (o, f, r, s), (o2, f2, r2, s2) in zip(a, b):
        assert torch.equal(o, o2) and torch.equal(f, f2) and torch.equal(r, r2) and torch.equal(s, s2)
    o, f, r, s = a[-1]
    assert torch.equal(o, o[:1].expand_as(o)) and torch.equal(f, f[:1].expand_as(f))
    assert torch.equal(r, r[:1].expand_as(r)) and torch.equal(s, s[:1].expand_as(s))
    env.env_seed_stride = 1
    c = _run(env, 8, 17)
    assert torch.equal(c[-1][0][0], o[0])                       # env 0 still has seed 17
    assert not torch.equal(c[-1][0][1], o[1])                   # env 1 now has seed 18
    assert float(c[-1][3].std()) > 0
    env.close()


def test_run_integrator_equals_stepping():
    import torch
    from rlao_amd.env import BatchedAOEnv
    env = BatchedAOEnv(n_envs=4, device=0, dtype="f32")
    env.set_params(SMALL, camera="ideal", wfs_type="shackhartmann", gainCL=0.4)
    a = _run(env, 10, 3, gain=0.4)
    total_stepped, residual_stepped = env.total[:10].copy(), env.residual[:10].copy()
    env.generate_new_phase_screen(3)
    env.dm.coefs = 0
    env.dm_prev = 0
    env.measure()
    env.reset_soft()
    obs, rew, sr = env.run_integrator(0, 10)
    torch.cuda.synchronize()
    np.testing.assert_allclose(obs.cpu().numpy(), a[-1][0].cpu().numpy(), atol=1e-6)
    np.testing.assert_allclose(sr.cpu().numpy(), a[-1][3].cpu().numpy(), atol=1e-6)
    # the telemetry of the on-device loop against the stepped run's (env.total / env.residual, OOPAOEnv.py:497, 522)
    np.testing.assert_allclose(env.total[:10], total_stepped, rtol=1e-6)
    np.testing.assert_allclose(env.residual[:10], residual_stepped, rtol=1e-6)
    assert (total_stepped > 0).all() and (residual_stepped > 0).all()
    # the on-device episode return (aoenv_set_return_accumulator) is the sum of the step rewards
    ret = torch.zeros(4, device=env.device, dtype=env.tdtype)
    env.generate_new_phase_screen(3)
    env.dm.coefs = 0
    env.dm_prev = 0
    env.measure()
    env.reset_soft()
    env.accumulate_returns(ret)
    env.run_integrator(0, 10)
    env.accumulate_returns(None)
    env.run_integrator(10, 2)                                   # detached: not counted
    torch.cuda.synchronize()
    want = sum(step[2].double() for step in a)
    np.testing.assert_allclose(ret.cpu().numpy(), want.cpu().numpy(), rtol=2e-6)
    with pytest.raises(ValueError):
        env.accumulate_returns(torch.zeros(3, device=env.device))
    env.close()


def test_random_actions_match_oracle_c2():
    """Three envs of the BASELINE geometry, arbitrary bounded actions (not the integrator), against the oracle."""
    import torch
    from oracle import ao_oracle as O
    from rlao_amd.env import BatchedAOEnv
    env = BatchedAOEnv(n_envs=3, device=0, dtype="f32", env_seed_stride=100)
    env.set_params(C2, camera="ideal", wfs_type="shackhartmann")
    env.generate_new_phase_screen(23)
    env.dm.coefs = 0
    env.measure()
    obs = env.reset_soft().cpu().numpy()
    rs = np.random.RandomState(9)
    orcs = []
    for k in range(3):
        o = O.OracleEnv(resolution=120, diameter=8.0, n_subap=20, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
                        fractionalR0=[1.0], altitude=[0.0], m2c=env.M2C_CL, n_modes=50)
        o.new_episode(23 + 100 * k)
        np.testing.assert_allclose(obs[k], o.reset_soft(), atol=3e-5)
        orcs.append(o)
    for i in range(6):
        act = (0.4 * obs + 0.05 * rs.randn(*obs.shape)).astype(np.float32)
        act *= env.dm_mask[None]
        got, frame, rew, sr, _, _ = env.step(i, torch.as_tensor(act))
        obs = got.cpu().numpy()
        for k, o in enumerate(orcs):
            oo, of, orw, osr, _, _ = o.step(i, act[k])
            np.testing.assert_allclose(obs[k], oo, atol=3e-5)
            np.testing.assert_allclose(float(rew[k]), orw, rtol=1e-4)
            np.testing.assert_allclose(float(sr[k]), osr, atol=1e-5)
            np.testing.assert_allclose(frame[k].cpu().numpy(), of, atol=5e-5 * of.max())
    env.close()


def test_single_env_flavour_returns_reference_types(golden_dir):
    """OOPAO(): numpy / float returns so that drl4ao's own TorchWrapper and run() loops work unchanged."""
    import torch
    from rlao_amd.env import OOPAO
    from rlao_amd.wrappers import TorchWrapper
    g = np.load(os.path.join(golden_dir, "small_sh.npz"))
    env = OOPAO(device=0, dtype="f64")
    env.set_params_file("Conf.parameterFile_oopao_parser", "AO_OOPAO")
    env.set_params(SMALL, camera="ideal", wfs_type="shackhartmann", m2c=g["m2c"])
    env.atm.generateNewPhaseScreen(17)
    env.dm.coefs = 0
    env.tel * env.dm * env.wfs
    obs = env.reset_soft()
    assert isinstance(obs, np.ndarray) and obs.shape == (9, 9) and obs.dtype == np.float64
    np.testing.assert_allclose(obs, g["s17_obs0"], atol=1e-6)
    nxt, wfsf, reward, strehl, done, info = env.step(0, g["s17_actions"][0])
    assert isinstance(nxt, np.ndarray) and isinstance(wfsf, np.ndarray) and isinstance(reward, float)
    assert isinstance(strehl, float) and done is False and set(info) == {"strehl"}
    np.testing.assert_allclose(nxt, g["s17_obs"][0], atol=1e-6)
    np.testing.assert_allclose(reward, g["s17_reward"][0], rtol=1e-6)
    assert env.wfs.signal.shape == (env.nSignal,) and env.wfs.cam.frame.shape == (48, 48)
    assert env.dm.coefs.shape == (env.nValidAct,)
    np.testing.assert_allclose(env.total[0], g["s17_total"][0], atol=1e-4)
    mean, std = env.calculate_strehl_AVG()
    assert abs(mean - g["s17_strehl"][0]) < 1e-7 and std == 0.0 and env.SR == []
    # wrappers + helpers
    w = TorchWrapper(env)
    o2, _, r2, s2, d2, info2 = w.step(1, torch.as_tensor(g["s17_actions"][1]))
    assert o2.dtype == torch.float32 and o2.device.type == "cpu" and info2[0][0] == "strehl"
    np.testing.assert_allclose(o2.numpy(), g["s17_obs"][1], atol=1e-5)
    img = env.vec_to_img(np.arange(env.nValidAct, dtype=float))
    np.testing.assert_array_equal(env.img_to_vec(img), np.arange(env.nValidAct))
    assert env.img_to_vec(torch.zeros(2, 5, 9, 9)).shape == (2, 5, env.nValidAct)
    np.random.seed(5)
    n1 = env.sample_noise(0.2, use_torch=True)
    np.random.seed(5)
    want = env.F @ (0.2 * np.random.normal(0, 1, size=(env.nValidAct,)))
    np.testing.assert_allclose(env.img_to_vec(n1).cpu().numpy(), want, rtol=1e-5, atol=1e-7)
    env.atm.windSpeed = [20.0]
    env.atm.r0 = 0.1
    env.step(2, np.zeros((9, 9)))
    env.close()


def test_c_abi_reports_errors():
    from rlao_amd import _lib as L
    from rlao_amd.env import BatchedAOEnv
    lib = L.load()
    env = BatchedAOEnv(n_envs=2, device=0, dtype="f32")
    env.set_params(SMALL, camera="ideal", wfs_type="shackhartmann")
    h = env._shard.h
    assert lib.aoenv_step(h, 10 ** 6, C.c_void_p(env._obs.data_ptr()), C.c_void_p(env._obs.data_ptr()), None, None, None, None) != 0
    assert b"frame index" in lib.aoenv_last_error()
    assert lib.aoenv_step(h, 0, None, None, None, None, None, None) != 0
    bad = np.zeros(3)
    assert lib.aoenv_upload(h, L.C_RECON, bad.ctypes.data_as(C.c_void_p), bad.nbytes) != 0
    assert b"expected" in lib.aoenv_last_error()
    assert lib.aoenv_upload(h, 999, bad.ctypes.data_as(C.c_void_p), bad.nbytes) != 0
    assert lib.aoenv_set_option(h, 12345, 1) != 0
    assert lib.aoenv_download(h, L.B_SIGNAL, bad.ctypes.data_as(C.c_void_p), 8, None) != 0
    idx = np.full(env._atm_tables.n_inner, 10 ** 6, dtype=np.int32)
    assert lib.aoenv_upload(h, L.C_INNER_IDX, idx.ctypes.data_as(C.c_void_p), idx.nbytes) != 0
    with pytest.raises(ValueError):
        env.step(0, np.zeros((5, 5)))
    # the env still works after the rejected calls
    env.step(0, np.zeros((2, 9, 9), dtype=np.float32))
    env.close()


@pytest.mark.parametrize("dtype,n_layer", [("f64", 1), ("f32", 3)])
def test_device_phase_screens_match_oracle(dtype, n_layer):
    """aoenv_new_screens_device (MT19937 normals, float64 FFT + sub-harmonics on the GPU) against the oracle's
    restatement of ft_sh_phase_screen (OOPAO/phaseStats.py:243-318), for every env and layer of the shard; and the
    NumPy upload path gives the same loop."""
    from oracle import ao_oracle as O                       # checker only
    from rlao_amd import _lib as L
    from rlao_amd.env import BatchedAOEnv
    geo = dict(SMALL, windSpeed=[10.0, 7.0, 12.0][:n_layer], windDirection=[72.0, 0.0, 144.0][:n_layer],
               fractionalR0=[[1.0], None, [0.6, 0.25, 0.15]][n_layer - 1], altitude=[0.0] * n_layer)
    env = BatchedAOEnv(n_envs=5, device=0, dtype=dtype)
    env.set_params(geo, camera="ideal", wfs_type="shackhartmann")
    env.env_seed_stride = 7
    env.generate_new_phase_screen(41)
    at = env._atm_tables
    S, N = at.S, at.N
    maps = env._shard.download(L.B_SCREEN, (n_layer, 5, S, S))
    tol = 1e-9 if dtype == "f64" else 2e-5
    for e, seed in enumerate(env.env_seeds(41)):
        for l in range(n_layer):
            want = O.ft_sh_phase_screen(geo["r0"], geo["L0"], N, at.layer_D / N, int(seed) + l)
            np.testing.assert_allclose(maps[l, e, 1:-1, 1:-1], want, rtol=0, atol=tol)
    env.dm.coefs = 0
    env.measure()
    obs_dev = env.reset_soft().cpu().numpy()
    host = np.stack([np.stack([O.ft_sh_phase_screen(geo["r0"], geo["L0"], N, at.layer_D / N, int(seed) + l) for l in range(n_layer)])
                     for seed in env.env_seeds(41)])
    env.generate_new_phase_screen(41, screens=host)           # the same screens made by the oracle on the host and uploaded
    maps_host = env._shard.download(L.B_SCREEN, (n_layer, 5, S, S))
    np.testing.assert_allclose(maps, maps_host, rtol=0, atol=tol * 10)       # the ring X = A Z + B xi amplifies 1e-12
    env.dm.coefs = 0
    env.measure()
    np.testing.assert_allclose(env.reset_soft().cpu().numpy(), obs_dev, atol=1e-6 if dtype == "f64" else 3e-5)
    env.close()


@pytest.mark.parametrize("n_sub", [3, 7, 9])
def test_pyramid_prime_radix_fft_matches_oracle(n_sub):
    """Pyramid sizes whose padded FFT length nRes = 6 (2 nSub + 8) has a prime factor 7 / 11 / 13 (the BASELINE 40x40
    Pyramid has nRes = 528 = 16.3.11): the register-resident prime-radix stage against the oracle's NumPy FFT path,
    float64 shard, both mask centrings, with and without modulation."""
    from oracle import ao_oracle as O                       # checker only
    from rlao_amd import _lib as L
    from rlao_amd.env import BatchedAOEnv
    D = 0.4 * n_sub
    geo = dict(diameter=D, nSubaperture=n_sub, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
               fractionalR0=[1.0], altitude=[0.0], nModes=4, nLoop=16)
    for centering, mod in ((True, 0.0), (False, 2.0)):
        env = BatchedAOEnv(n_envs=2, device=0, dtype="f64")
        env.set_params(dict(geo, psfCentering=centering, modulation=mod), camera="ideal", wfs_type="pyramid")
        ref = O.OracleEnv(resolution=6 * n_sub, diameter=D, n_subap=n_sub, r0=0.13, L0=30.0, windSpeed=[10.0],
                          windDirection=[72.0], fractionalR0=[1.0], altitude=[0.0], m2c=env.M2C_CL, n_modes=4,
                          wfs_type="pyramid", modulation=mod, psf_centering=centering)
        assert env._pyr_tables.nRes % {3: 7, 7: 11, 9: 13}[n_sub] == 0
        env.generate_new_phase_screen(5)
        env.dm.coefs = 0
        env.measure()
        ref.new_episode(5)
        frame = env._shard.download(L.B_FRAME, (2, env.cam_res, env.cam_res))[0]
        sig = env._shard.download(L.B_SIGNAL, (2, env.nSignal))[0]
        np.testing.assert_allclose(frame, ref.wfs.frame, rtol=0, atol=1e-9 * ref.wfs.frame.max())
        np.testing.assert_allclose(sig, ref.wfs.signal, rtol=0, atol=1e-8)
        env.close()


@pytest.mark.parametrize("noise", [False, True])
def test_checkpoint_resume_is_bitwise(noise):
    """get_state() / set_state(): an episode resumed from a snapshot (taken mid-way, between two ring extrusions, with the
    screens stored as shifted tori) continues bit for bit -- on the same env and on a freshly built one."""
    import torch
    from rlao_amd.env import BatchedAOEnv

    def make():
        env = BatchedAOEnv(n_envs=3, device=0, dtype="f32")
        env.set_params(dict(SMALL, windSpeed=[35.0]), camera="ideal", wfs_type="shackhartmann")       # a crossing every ~1.1 steps
        if noise:
            env.wfs.cam.photonNoise, env.wfs.cam.readoutNoise = True, 3
        return env

    env = make()
    env.generate_new_phase_screen(11)
    env.dm.coefs = 0
    env.measure()
    env.reset_soft()
    env.run_integrator(0, 7)
    snap = env.get_state()
    obs_a, rew_a, sr_a = (t.clone() for t in env.run_integrator(7, 9))
    maps_a = env.get_state()["screen"]
    # same env, rewound
    env.set_state(snap)
    obs_b, rew_b, sr_b = (t.clone() for t in env.run_integrator(7, 9))
    assert torch.equal(obs_a, obs_b) and torch.equal(rew_a, rew_b) and torch.equal(sr_a, sr_b)
    np.testing.assert_array_equal(env.get_state()["screen"], maps_a)
    env.close()
    # a new env of the same geometry picks the episode up
    env2 = make()
    env2.set_state(snap)
    obs_c, rew_c, sr_c = env2.run_integrator(7, 9)
    assert torch.equal(obs_a, obs_c) and torch.equal(rew_a, rew_c) and torch.equal(sr_a, sr_c)
    np.testing.assert_array_equal(env2.get_state()["screen"], maps_a)
    env2.close()


@pytest.mark.parametrize("dtype,tol", [("f64", 2e-7), ("f32", 2e-5)])
def test_science_psf_matches_oracle(dtype, tol):
    """env.tel.computePSF(zp) (aoenv_compute_psf: zero-padded pupil FFT, 2 x 2 binned) against the oracle's restatement of
    Telescope.computePSF, on the residual phase of a stepped loop; float64 differs from the reference only by its
    complex64 phasor (6e-8 relative)."""
    import torch
    from oracle import ao_oracle as O                       # checker only
    from rlao_amd import _lib as L
    from rlao_amd.env import BatchedAOEnv
    env = BatchedAOEnv(n_envs=3, device=0, dtype=dtype)
    env.set_params(SMALL, camera="ideal", wfs_type="shackhartmann")
    out = _run(env, 4, 9)
    phase = env._shard.download(L.B_PHASE, (3, env.R, env.R)).astype(np.float64)
    flux = env._sh_tables.flux_map
    for zp in (2, 3):
        psf = env.tel.computePSF(zp)
        torch.cuda.synchronize()
        assert tuple(psf.shape) == (3, zp * env.R, zp * env.R)
        for e in range(3):
            want = O.telescope_psf(env.pupil.astype(float), flux, phase[e], zp)
            np.testing.assert_allclose(psf[e].double().cpu().numpy(), want, rtol=0, atol=tol * want.max())
    env.close()


def test_history_env_on_the_device():
    """HistoryEnv over the real batched env: reset() + steps, history stays on the GPU and matches plain stepping."""
    import torch
    from rlao_amd.env import BatchedAOEnv
    from rlao_amd.wrappers import HistoryEnv
    env = BatchedAOEnv(n_envs=4, device=0, dtype="f32")
    env.set_params(SMALL, camera="ideal", wfs_type="shackhartmann")
    h = HistoryEnv(env, n_history=5, delay=1)
    hist, info = h.reset(seed=3)
    assert hist.is_cuda and tuple(hist.shape) == (4, 5, env.nActuator, env.nActuator)
    first = hist[:, 0].clone()
    acts = []
    for k in range(3):
        a = 0.4 * hist[:, 0]
        acts.append(a.clone())
        hist, rew, term, trunc, info = h.step(a)
    # the same episode by hand
    env.atm.generateNewPhaseScreen(3)
    env.dm.coefs = 0
    env.dm_prev = 0                                             # (the prologue alone keeps the previous episode's last command)
    env.tel * env.dm * env.wfs
    obs = env.reset_soft()
    assert torch.equal(obs, first)
    want = [obs.clone()]
    for k in range(3):
        obs, _, _, sr, _, _ = env.step(k, acts[k])
        want.append(obs.clone())
    for j in range(4):
        assert torch.equal(hist[:, j], want[3 - j])
    assert torch.equal(hist[:, 4], torch.zeros_like(first)) and torch.equal(rew, sr)
    env.close()


def test_two_dms_match_reference_and_oracle(golden_dir):
    """BASELINE configs[4] hook: two chained DMs (tel*dm1*dm2*wfs).  (a) float64 measurement of the reference's recorded
    atmosphere OPD + commands on both mirrors against the reference's own signal (tests/golden/two_dm.npz); (b) a closed
    loop over both mirrors against the oracle."""
    import torch
    from oracle import ao_oracle as O                       # checker only
    from rlao_amd import _lib as L
    from rlao_amd.env import BatchedAOEnv
    g = np.load(os.path.join(golden_dir, "two_dm.npz"))
    ns2 = int(g["cfg_ns2"])
    env = BatchedAOEnv(n_envs=3, device=0, dtype="f64")
    env.set_params(SMALL, camera="ideal", wfs_type="shackhartmann", second_dm=dict(nSubaperture=ns2))
    A1, A2 = g["coefs1"].shape[1], g["coefs2"].shape[1]
    assert env.nValidAct == A1 + A2 and env.nActuator == 9 + ns2 + 1
    env._shard.set_atm_opd(g["opd_atm"].reshape(3, -1))
    env._shard.set_coefs(np.concatenate([g["coefs1"], g["coefs2"]], axis=1))
    env.measure()
    sig = env._shard.download(L.B_SIGNAL, (3, env.nSignal))
    np.testing.assert_allclose(sig, g["signal"], rtol=0, atol=1e-8)
    opd = env._shard.download(L.B_PHASE, (3, env.R, env.R)) * (env.src_wavelength / (2 * np.pi))
    np.testing.assert_allclose(opd, g["opd"], rtol=0, atol=1e-15)
    env.close()
    for dtype, tol in (("f64", 1e-6), ("f32", 5e-5)):
        env = BatchedAOEnv(n_envs=2, device=0, dtype=dtype)
        env.set_params(SMALL, camera="ideal", wfs_type="shackhartmann", second_dm=dict(nSubaperture=ns2))
        ref = O.OracleEnv(resolution=48, diameter=3.2, n_subap=8, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
                          fractionalR0=[1.0], altitude=[0.0], m2c=env.M2C_CL, n_modes=20, second_dm_nsub=ns2)
        env.generate_new_phase_screen(13)
        env.dm.coefs = 0
        env.measure()
        obs = env.reset_soft()
        ref.new_episode(13)
        o_obs = ref.reset_soft()
        np.testing.assert_allclose(obs[0].cpu().numpy(), o_obs, atol=tol)
        for i in range(6):
            act = (0.5 * obs).float()
            obs, frame, rew, sr, done, info = env.step(i, act)
            o_obs, _, o_rew, o_sr, _, _ = ref.step(i, act[0].cpu().numpy())
            np.testing.assert_allclose(obs[0].cpu().numpy(), o_obs, atol=tol)
            assert abs(float(sr[0]) - o_sr) < tol
        assert float(torch.abs(env.dm.coefs if torch.is_tensor(env.dm.coefs) else torch.as_tensor(env.dm.coefs)).max()) > 0
        env.close()


def test_long_closed_loop_tracks_the_oracle():
    """300 closed-loop steps (about 100 ring extrusions, every one drawing from the MT19937 stream) of the float32 fused
    step kernel against the float64 oracle: the loop is contractive, so the float32 error stays at its one-step level."""
    from oracle import ao_oracle as O                       # checker only
    from rlao_amd.env import BatchedAOEnv
    env = BatchedAOEnv(n_envs=2, device=0, dtype="f32")
    env.set_params(dict(SMALL, nLoop=400), camera="ideal", wfs_type="shackhartmann")
    ref = O.OracleEnv(resolution=48, diameter=3.2, n_subap=8, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
                      fractionalR0=[1.0], altitude=[0.0], m2c=env.M2C_CL, n_modes=20, nLoop=400)
    env.generate_new_phase_screen(29)
    env.dm.coefs = 0
    env.measure()
    obs = env.reset_soft()
    ref.new_episode(29)
    o_obs = ref.reset_soft()
    worst = 0.0
    for i in range(300):
        act = (0.5 * obs).float()
        obs, frame, rew, sr, done, info = env.step(i, act)
        o_obs, _, o_rew, o_sr, _, _ = ref.step(i, act[0].cpu().numpy())
        worst = max(worst, float(np.abs(obs[0].cpu().numpy() - o_obs).max()))
        assert abs(float(sr[0]) - o_sr) < 2e-5
    assert worst < 5e-5, worst
    np.testing.assert_allclose(env.residual[:300, 0], ref.residual[:300], atol=5e-3)
    np.testing.assert_allclose(env.total[:300, 0], ref.total[:300], atol=5e-3)
    env.close()


def test_frame_view_is_the_copied_frame_without_the_copy():
    """return_frame="view": step() hands out a tensor aliasing the library's frame buffer (no device copy per step); same values as
    the copied frame, overwritten by the next step; return_frame=False hands out None."""
    import torch
    from rlao_amd.env import BatchedAOEnv
    outs = {}
    for mode in (True, "view", False):
        env = BatchedAOEnv(n_envs=3, device=0, dtype="f32", return_frame=mode)
        env.set_params(SMALL, camera="papyrus", wfs_type="shackhartmann")
        env.generate_new_phase_screen(2)
        env.dm.coefs = 0
        env.dm_prev = 0
        env.measure()
        obs = env.reset_soft()
        frames = []
        for i in range(3):
            obs, fr, _, _, _, _ = env.step(i, 0.5 * obs)
            frames.append(fr)
        torch.cuda.synchronize()
        outs[mode] = (obs.clone(), frames, None if frames[-1] is None else frames[-1].clone())
        env.close()
    assert outs[False][1] == [None, None, None]
    assert torch.equal(outs[True][0], outs["view"][0]) and torch.equal(outs[True][0], outs[False][0])
    assert torch.equal(outs[True][2], outs["view"][2])                          # the last frame: the same
    assert outs["view"][1][0].data_ptr() == outs["view"][1][2].data_ptr()       # one buffer, overwritten ...
    assert outs[True][1][0].data_ptr() != outs[True][1][2].data_ptr()           # ... against a new tensor per step
    assert not torch.equal(outs[True][1][0], outs[True][1][2])


def test_fused_step_query_tells_which_path_runs():
    """env.fused_step (aoenv_fused_step_active): the one-kernel step inside its envelope, the batched kernels outside it."""
    from rlao_amd import _lib as L
    from rlao_amd.env import BatchedAOEnv
    for dtype, wfs, geo, want in (("f32", "shackhartmann", SMALL, True), ("f64", "shackhartmann", SMALL, False),
                                  ("f32", "pyramid", dict(SMALL, modulation=0.0), False),
                                  ("f32", "shackhartmann", dict(SMALL, nModes=60), False)):       # > 52 modes: outside the envelope
        env = BatchedAOEnv(n_envs=2, device=0, dtype=dtype)
        env.set_params(geo, camera="ideal", wfs_type=wfs)
        assert env.fused_step is want, (dtype, wfs)
        if want:
            L.check(env._shard.lib.aoenv_set_option(env._shard.h, L.OPT_FUSED_STEP, 0))
            assert env.fused_step is False
        env.close()


def test_tensors_handed_out_by_step_survive_the_next_episode():
    """step() hands out new tensors (a replay buffer may keep them, as it keeps the reference's arrays): the episode prologue
    (reset_soft), the on-device loop (run_integrator) and a restored checkpoint must not write into them."""
    import torch
    from rlao_amd.env import BatchedAOEnv
    env = BatchedAOEnv(n_envs=3, device=0, dtype="f32")
    env.set_params(SMALL, camera="ideal", wfs_type="shackhartmann", gainCL=0.4)
    env.generate_new_phase_screen(3)
    obs0 = env.reset_soft()
    obs, frame, rew, sr, done, info = env.step(0, 0.4 * obs0)
    kept = [t.clone() for t in (obs0, obs, frame, rew, sr, env.SR[-1])]
    state = env.get_state()
    env.generate_new_phase_screen(4)
    env.dm.coefs = 0
    env.measure()
    obs1 = env.reset_soft()
    env.run_integrator(0, 5)
    o2, _, r2, s2, _, _ = env.step(5, 0.1 * obs1)
    env.set_state(state)
    env.run_integrator(1, 3)
    torch.cuda.synchronize()
    for t, k in zip((obs0, obs, frame, rew, sr, env.SR[0]), kept):
        assert torch.equal(t, k)
    assert not torch.equal(o2, obs) and o2.data_ptr() != obs.data_ptr()
    env.close()


@pytest.mark.parametrize("coefs_image", [0, 1])
def test_phase_kernel_with_16_byte_accesses_is_bit_identical(coefs_image):
    """k_phase_mfma4 (lane = one row, four consecutive pixels: float4 tile loads, taps, pupil flags and phase stores) against
    k_phase_mfma (dword accesses; forced through the diagnostic switch 99 = 256): the same taps in the same order and the same k order
    of the DM product, so every output bit agrees -- three layers with winds that move the torus origins (float4s that straddle the
    wrap), both forms of the Gy C operand (computed in the kernel / handed over by k_dm_rows)."""
    import torch
    from rlao_amd import _lib as L
    from rlao_amd.env import BatchedAOEnv
    geo = dict(SMALL, windSpeed=[40.0, 25.0, 33.0], windDirection=[72.0, 200.0, 310.0], fractionalR0=[0.6, 0.3, 0.1],
               altitude=[0.0, 0.0, 0.0])
    outs = []
    for old in (0, 1):
        env = BatchedAOEnv(n_envs=3, device=0, dtype="f32")
        env.set_params(geo, camera="ideal", wfs_type="shackhartmann", gainCL=0.4)
        L.check(env._shard.lib.aoenv_set_option(env._shard.h, L.OPT_FUSED_STEP, 0))
        L.check(env._shard.lib.aoenv_set_option(env._shard.h, L.OPT_COEFS_IMAGE, coefs_image))
        if old:
            L.check(env._shard.lib.aoenv_set_option(env._shard.h, 99, 256))
        rec = _run(env, 12, 5, gain=0.4)
        phase = env._shard.download(L.B_PHASE, (3, env.R, env.R))
        opd = env._shard.download(L.B_OPD_ATM, (3, env.R, env.R))
        outs.append((rec, phase, opd, env.total[:12].copy(), env.residual[:12].copy()))
        env.close()
    (ra, pa, oa, ta, sa), (rb, pb, ob, tb, sb) = outs
    for x, y in zip(ra, rb):
        assert all(torch.equal(p, q) for p, q in zip(x, y))
    assert np.array_equal(pa, pb) and np.array_equal(oa, ob) and np.array_equal(ta, tb) and np.array_equal(sa, sb)
    assert np.abs(pa).max() > 0
