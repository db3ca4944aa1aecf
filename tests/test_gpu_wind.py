"""GPU: per-env winds (aoenv_set_wind_env -- every env of a shard its own wind speed / direction per layer, clocks advanced on the
device).  The reference changes the wind between runs (MAIN/integrator_oopao_razor.py:41-44 through the setters of
OOPAO/Atmosphere.py:829-873); batched, that is one wind per env.  The checker is the shared host clock (pinned to the reference by
the golden replays) and the oracle: an env stepped by its own clock must be BIT-IDENTICAL to a shard stepped with that wind."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SMALL = dict(diameter=3.2, nSubaperture=8, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
             fractionalR0=[1.0], altitude=[0.0], nModes=20, nLoop=64)
SMALL3 = dict(SMALL, windSpeed=[10.0, 25.0, 18.0], windDirection=[0.0, 72.0, 200.0], fractionalR0=[0.6, 0.25, 0.15],
              altitude=[0.0, 1000.0, 5000.0])
# pixel 6.7 cm, 500 Hz: 28 m/s = 0.84 px per frame (the per-env clocks take < 1 px per frame and axis)
SPEEDS = np.array([[0.0], [10.0], [17.0], [28.0], [12.0], [24.0]])
DIRS = np.array([[0.0], [72.0], [190.0], [270.0], [-45.0], [135.0]])


def _episode(env, steps, seed, winds=None, gain=0.5):
    """one closed-loop episode; winds = (speed, direction) [n_envs, nLayer] -> per-env clocks"""
    import torch
    env.generate_new_phase_screen(seed)
    if winds is not None:
        env.set_wind_per_env(winds[0], winds[1], reset=True)
    env.dm.coefs = 0
    env.dm_prev = 0
    env.measure()
    obs = env.reset_soft()
    out = []
    for i in range(steps):
        obs, frame, rew, sr, _, _ = env.step(i, gain * obs)
        out.append((obs.clone(), frame.clone(), rew.clone(), sr.clone()))
    torch.cuda.synchronize()
    return out


def _make(n, dtype, geo=SMALL, **kw):
    from rlao_amd.env import BatchedAOEnv
    env = BatchedAOEnv(n_envs=n, device=0, dtype=dtype, env_seed_stride=0, **kw)   # the same screens in every env: only the wind differs
    env.set_params(geo, camera="ideal", wfs_type="shackhartmann")
    return env


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_each_env_equals_a_shard_with_its_wind(dtype):
    """6 envs, 6 winds (calm, slow, fast, along -x, negative angle ...), 24 steps (the fast ones cross ~20 pixels): env e of the
    per-env shard == env 0 of a shared-clock shard whose wind is wind e, bit for bit -- obs, frame, reward, Strehl, and the
    logical screens at the end.  float32 runs the fused step kernel (deferred per-env ring scatter), float64 the batched kernels
    (per-env scatter + range)."""
    import torch
    from rlao_amd import _lib as L
    n = len(SPEEDS)
    env = _make(n, dtype)
    got = _episode(env, 24, 9, winds=(SPEEDS, DIRS))
    scr = env._shard.download(L.B_SCREEN, (1, n, env._atm_tables.S, env._atm_tables.S), env._stream())
    clk = env._shard.get_clock_env(1, n)
    env.close()
    assert np.abs(clk[..., 2:]).max() < 1 and np.abs(clk[0, 3, :2]).max() > 0.8            # 28 m/s: 0.84 px per frame
    for e in range(n):
        ref = _make(1, dtype)
        ref.atm.windSpeed = list(SPEEDS[e])
        ref.atm.windDirection = list(DIRS[e])
        want = _episode(ref, 24, 9)
        rs = ref._shard.download(L.B_SCREEN, (1, 1, ref._atm_tables.S, ref._atm_tables.S), ref._stream())
        rb = ref._shard.get_buff(1)
        ref.close()
        for (o, f, r, s), (o0, f0, r0, s0) in zip(got, want):
            assert torch.equal(o[e], o0[0]) and torch.equal(f[e], f0[0]) and torch.equal(r[e], r0[0]) and torch.equal(s[e], s0[0]), e
        assert np.array_equal(scr[0, e], rs[0, 0]), e
        np.testing.assert_array_equal(clk[0, e, 2:], rb[0])
    # the winds really differ: env 0 (calm) never moves, env 3 does
    assert not torch.equal(got[-1][0][0], got[-1][0][3])


def test_uniform_per_env_clocks_equal_the_shared_clock_three_layers():
    """3 layers, 4 envs with distinct seeds, every env given the shard's own wind through the per-env path == the shared host clock."""
    import torch
    from rlao_amd.env import BatchedAOEnv
    outs = []
    for per_env in (False, True):
        env = BatchedAOEnv(n_envs=4, device=0, dtype="f32")
        env.set_params(SMALL3, camera="ideal", wfs_type="shackhartmann")
        w = (np.tile(SMALL3["windSpeed"], (4, 1)), np.tile(SMALL3["windDirection"], (4, 1))) if per_env else None
        outs.append(_episode(env, 20, 3, winds=w))
        env.close()
    for x, y in zip(*outs):
        assert all(torch.equal(p, q) for p, q in zip(x, y))


def test_per_env_wind_matches_oracle_and_survives_checkpoint_and_new_episode():
    """(a) env 2 (17 m/s at 190 deg) against the NumPy oracle with that wind; (b) get_state / set_state in the middle of an episode
    continues bit for bit; (c) a new episode keeps every env's wind and restarts its clock; (d) wrong shapes / > 1 px per frame are
    refused; the shared-wind setter afterwards gives every env the same wind again."""
    import torch
    from oracle import ao_oracle as O                       # checker only
    from rlao_amd import _lib as L
    n = len(SPEEDS)
    env = _make(n, "f32")
    a = _episode(env, 10, 9, winds=(SPEEDS, DIRS))
    orc = O.OracleEnv(resolution=48, diameter=3.2, n_subap=8, r0=0.13, L0=30.0, windSpeed=list(SPEEDS[2]), windDirection=list(DIRS[2]),
                      fractionalR0=[1.0], altitude=[0.0], m2c=env.M2C_CL, n_modes=20)
    orc.new_episode(9)
    obs_o = orc.reset_soft()
    for i in range(10):
        obs_o, fr_o, rw_o, sr_o, _, _ = orc.step(i, 0.5 * obs_o)
        np.testing.assert_allclose(a[i][0][2].cpu().numpy(), obs_o, atol=3e-5)
        np.testing.assert_allclose(float(a[i][3][2]), sr_o, atol=1e-5)
    # (b) checkpoint in the middle of an episode
    snap = env.get_state()
    obs = a[-1][0]
    cont = []
    for i in range(10, 16):
        obs, frame, rew, sr, _, _ = env.step(i, 0.5 * obs)
        cont.append((obs.clone(), frame.clone(), rew.clone(), sr.clone()))
    env2 = _make(n, "f32")
    env2.generate_new_phase_screen(1)                               # some other state first
    env2.set_state(snap)
    obs = a[-1][0].clone()
    for i in range(10, 16):
        obs, frame, rew, sr, _, _ = env2.step(i, 0.5 * obs)
        x = cont[i - 10]
        assert torch.equal(obs, x[0]) and torch.equal(frame, x[1]) and torch.equal(rew, x[2]) and torch.equal(sr, x[3])
    env2.close()
    # (c) a new episode: the winds are kept, the clocks restart
    b = _episode(env, 10, 9)
    for x, y in zip(a, b):
        assert all(torch.equal(p, q) for p, q in zip(x, y))
    # (d)
    with pytest.raises(ValueError):
        env.set_wind_per_env(SPEEDS[:3], DIRS[:3])
    with pytest.raises(L.AoEnvError, match="< 1"):
        env.set_wind_per_env(SPEEDS * 3, DIRS)
    env.atm.windSpeed = [17.0]
    env.atm.windDirection = [190.0]
    c = _episode(env, 10, 9)
    for x, y in zip(a, c):
        assert torch.equal(x[0][2], y[0][2]) and torch.equal(y[0][0], y[0][2])     # every env now has env 2's wind
    env.close()


@pytest.mark.parametrize("dtype", ["f32", "f64"])
def test_atm_update_alone_moves_the_atmosphere_like_a_step(dtype):
    """atm.update() (OOPAO/Atmosphere.py:439-477) without the rest of env.step(): 15 updates == 15 steps with a zero command, as far
    as the atmosphere goes (atm.OPD_no_pupil bit for bit; the steps run the fused kernel in float32, the update the batched one)."""
    import torch
    from rlao_amd import _lib as L
    a, b = _make(3, dtype), _make(3, dtype)
    L.check(b._shard.lib.aoenv_set_option(b._shard.h, L.OPT_STORE_ATM_OPD, 1))
    for env in (a, b):
        env.env_seed_stride = 1
        env.atm.windSpeed = [25.0]                                  # 0.75 px per frame: ~11 ring extrusions
        env.generate_new_phase_screen(4)
        env.dm.coefs = 0
        env.dm_prev = 0
        env.measure()
    zero = torch.zeros_like(b.reset_soft())
    for i in range(15):
        a.atm.update()
        b.step(i, zero)
    oa, ob = a.atm.OPD_no_pupil, b.atm.OPD_no_pupil
    assert np.array_equal(oa, ob) and np.abs(oa).max() > 0 and not np.array_equal(oa[0], oa[1])
    a.measure()                                                     # tel*dm*wfs after the updates: the frame of the 15th step
    fa = a._shard.download(L.B_FRAME, (3, a.cam_res, a.cam_res))
    fb = b._shard.download(L.B_FRAME, (3, b.cam_res, b.cam_res))
    np.testing.assert_allclose(fa, fb, atol=(0 if dtype == "f64" else 2e-5) * fb.max())   # (float32: batched vs fused kernels)
    a.close()
    b.close()


def test_per_env_winds_under_the_pyramid():
    """The Pyramid runs the batched kernels in float32 (no fused step): 4 envs, 4 winds, each bit-identical to a shard with that wind."""
    import torch
    from rlao_amd.env import BatchedAOEnv
    geo = dict(SMALL, modulation=0.0)
    sp, di = SPEEDS[[0, 2, 3, 5]], DIRS[[0, 2, 3, 5]]
    env = BatchedAOEnv(n_envs=4, device=0, dtype="f32", env_seed_stride=0)
    env.set_params(geo, camera="ideal", wfs_type="pyramid")
    got = _episode(env, 16, 5, winds=(sp, di))
    env.close()
    for e in range(4):
        ref = BatchedAOEnv(n_envs=1, device=0, dtype="f32", env_seed_stride=0)
        ref.set_params(geo, camera="ideal", wfs_type="pyramid")
        ref.atm.windSpeed = list(sp[e])
        ref.atm.windDirection = list(di[e])
        want = _episode(ref, 16, 5)
        ref.close()
        for (o, f, r, s_), (o0, f0, r0, s0) in zip(got, want):
            assert torch.equal(o[e], o0[0]) and torch.equal(f[e], f0[0]) and torch.equal(r[e], r0[0]) and torch.equal(s_[e], s0[0]), e
    assert not torch.equal(got[-1][0][0], got[-1][0][2])


def test_ring_pipeline_survives_wind_changes_and_other_consumers():
    """The operand of a layer's next crossing is prepared ahead (ring pipeline, AOENV_OPT_RING_LOOKAHEAD); whatever invalidates it must
    fall back to preparing in place: wind speed / direction changed in mid-episode (the next crossing may go the other way), a
    stand-alone measurement or atm.update() between steps (another consumer of the deferred ring), a checkpoint restored.  Pipeline
    on == pipeline off, bit for bit, through all of it."""
    import torch
    from rlao_amd import _lib as L
    outs = []
    for pipeline in (1, 0):
        env = _make(3, "f32")
        env.env_seed_stride = 1
        L.check(env._shard.lib.aoenv_set_option(env._shard.h, L.OPT_RING_LOOKAHEAD, pipeline))
        env.atm.windSpeed = [25.0]
        env.generate_new_phase_screen(8)
        env.dm.coefs = 0
        env.dm_prev = 0
        env.measure()
        obs = env.reset_soft()
        log = []
        snap = None
        for i in range(40):
            if i == 7:
                env.atm.windDirection = [250.0]                      # the prepared Z is for the old direction
            if i == 13:
                env.atm.windSpeed = [9.0]
            if i == 17:
                env.measure()                                        # tel*dm*wfs between two steps
            if i == 21:
                env.atm.update()                                     # the atmosphere alone moves a frame
            if i == 26:
                snap = env.get_state()
            if i == 31:
                env.set_state(snap)                                  # back to step 26's state (the loop goes on from there)
                obs = torch.as_tensor(snap["obs"], device=env.device)
            obs, frame, rew, sr, _, _ = env.step(i, 0.5 * obs)
            log.append((obs.clone(), frame.clone(), rew.clone(), sr.clone()))
        torch.cuda.synchronize()
        outs.append(log)
        env.close()
    for i, (x, y) in enumerate(zip(*outs)):
        assert all(torch.equal(p, q) for p, q in zip(x, y)), i
    assert not torch.equal(outs[0][6][0], outs[0][30][0])
