"""CPU: the NumPy oracle against the vectors the reference itself produced (tests/golden/*.npz).

The goldens come from ``oracle/make_goldens.py`` (reference classes executed in the build container).
Tolerances are float64 re-ordering noise only; the replayed actions are the recorded float32 actions, so
closed-loop chaos cannot mask a discrepancy.
"""
import os

import numpy as np
import pytest

from oracle import ao_oracle as O

# tiny_3layer_fov1: layers at 0 / 1000 / 5000 m under the reference env's own telescope (fov = 1 arcsec): grids of 28, 29, 29 pixels
CASES = ["tiny_sh", "tiny_fastwind", "tiny_3layer", "tiny_3layer_fov1", "small_sh", "c2_sh", "tiny_pyr", "tiny_pyr_mod", "papyrus_pyr", "c5_mcao"]


def _env_from_golden(g, **extra):
    kw = dict(extra)
    if "cfg_wfs" in g:
        kw.update(wfs_type="pyramid", modulation=float(g["cfg_modulation"]), psf_centering=bool(g["cfg_centering"]))
    if "cfg_second_nsub" in g:
        kw.update(second_dm_nsub=int(g["cfg_second_nsub"]))
    if "cfg_fov" in g:
        kw.update(fov_arcsec=float(g["cfg_fov"]))
    return O.OracleEnv(resolution=int(g["cfg_R"]), diameter=float(g["cfg_D"]), n_subap=int(g["cfg_nsub"]),
                       r0=float(g["cfg_r0"]), L0=float(g["cfg_L0"]), windSpeed=list(g["cfg_ws"]),
                       windDirection=list(g["cfg_wd"]), fractionalR0=list(g["cfg_frac"]),
                       altitude=list(g["cfg_alt"]), m2c=g["m2c"], n_modes=int(g["cfg_n_modes"]), **kw)


@pytest.fixture(scope="module", params=CASES)
def case(request, golden_dir):
    g = np.load(os.path.join(golden_dir, request.param + ".npz"))
    return request.param, g, _env_from_golden(g)


def test_constants(case):
    name, g, env = case
    assert np.array_equal(env.pupil, g["pupil"])
    assert env.wavelength == float(g["wavelength"]) and env.nPhoton == float(g["nPhoton"])
    assert np.array_equal(env.dm_mask.reshape(-1), g["validAct"])
    assert np.array_equal(env.xvalid, g["xvalid"]) and np.array_equal(env.yvalid, g["yvalid"])
    if "cfg_wfs" in g:
        assert np.array_equal(env.wfs.validI4Q, g["validI4Q"]) and env.wfs.nTheta == int(g["nTheta"])
        np.testing.assert_allclose(env.wfs.referenceSignal_2D, g["referenceSignal_2D"], atol=1e-13)
        m = env.wfs.m if env.R <= 48 else env.wfs.m[::7, ::5]
        np.testing.assert_allclose(m, g["pyr_m"], atol=1e-13)
    else:
        assert np.array_equal(env.wfs.valid_2d, g["valid_subap"])
        np.testing.assert_allclose(env.wfs.reference_slopes_maps, g["reference_slopes_maps"], atol=1e-13)
        np.testing.assert_allclose(env.wfs.slopes_units, float(g["slopes_units"]), rtol=1e-11)
    lay = env.atm.layers[0]
    if "cfg_layer_S" in g:                                      # fov != 0: every layer its own grid and ring operators
        assert [l.mapShift.shape[0] for l in env.atm.layers] == list(g["cfg_layer_S"])
        for i, l in enumerate(env.atm.layers[1:], start=1):
            np.testing.assert_allclose(l.A, g[f"A_l{i}"], atol=1e-12)
            np.testing.assert_allclose(l.B, g[f"B_l{i}"], atol=1e-12)
    if "A" in g:
        np.testing.assert_allclose(lay.A, g["A"], atol=1e-12)
        np.testing.assert_allclose(lay.B, g["B"], atol=1e-12)
        np.testing.assert_allclose(env.dm_modes, g["modes"], atol=1e-15)
    else:
        np.testing.assert_allclose(lay.A @ g["A_probe_in"], g["A_probe_out"], atol=1e-10)
        np.testing.assert_allclose(lay.B @ g["B_probe_in"], g["B_probe_out"], atol=1e-10)
        np.testing.assert_allclose(np.linalg.norm(lay.A), float(g["A_fro"]), rtol=1e-12)
        a1 = len(g["modes_probe_in"])
        np.testing.assert_allclose(env.dm_modes[:, :a1] @ g["modes_probe_in"], g["modes_probe_out"], atol=1e-12)
        if "modes2_probe_in" in g:                              # second (altitude-conjugated, fov = 0) mirror of the pair
            np.testing.assert_allclose(env.dm_modes[:, a1:] @ g["modes2_probe_in"], g["modes2_probe_out"], atol=1e-12)
    # separable DM factors reproduce the dense influence matrix
    if env.gx is not None:
        R = env.R
        sep = np.einsum("yi,xj->yxij", env.gy, env.gx).reshape(R * R, -1)[:, env.dm_mask.reshape(-1)]
        np.testing.assert_allclose(sep, env.dm_modes, atol=5e-16)


def test_calibration(case):
    name, g, env = case
    scale = np.abs(g["imat"]).max()
    np.testing.assert_allclose(env.imat, g["imat"], atol=1e-12 * scale)
    np.testing.assert_allclose(env.reconstructor, g["recon"], atol=1e-9 * np.abs(g["recon"]).max())
    if "F" in g:
        np.testing.assert_allclose(env.F, g["F"], atol=1e-12)


def test_closed_loop_replay(case):
    name, g, env = case
    for seed in g["cfg_seeds"]:
        p = f"s{int(seed)}_"
        env.dm_prev[:] = 0                                      # every recorded episode is the first one of a fresh env
        env.new_episode(int(seed))
        for l, lay in enumerate(env.atm.layers):
            S = lay.mapShift.shape[0]
            np.testing.assert_allclose(lay.mapShift, g[p + "mapShift0"][l][:S, :S], atol=1e-12)
        np.testing.assert_allclose(env.reset_soft(), g[p + "obs0"], atol=1e-11)
        full = {int(s): k for k, s in enumerate(g[p + "full_steps"])}
        acts = g[p + "actions"]
        for i in range(len(acts)):
            obs, frame, rew, sr, done, info = env.step(i, acts[i])
            np.testing.assert_allclose(obs, g[p + "obs"][i], atol=1e-10)
            np.testing.assert_allclose(rew, g[p + "reward"][i], atol=1e-10)
            np.testing.assert_allclose(sr, g[p + "strehl"][i], atol=1e-12)
            np.testing.assert_allclose(env.total[i], g[p + "total"][i], atol=1e-9)
            np.testing.assert_allclose(env.residual[i], g[p + "residual"][i], atol=1e-9)
            np.testing.assert_allclose(env.wfs.signal, g[p + "signal"][i], atol=1e-11)
            np.testing.assert_allclose(env.coefs, g[p + "coefs"][i], atol=1e-18)
            for l, lay in enumerate(env.atm.layers):
                np.testing.assert_allclose(lay.buff, g[p + "buff"][i, l], atol=1e-13)
            if i in full:
                k = full[i]
                np.testing.assert_allclose(env.atm.OPD, g[p + "opd_atm"][k], atol=1e-18)
                np.testing.assert_allclose(env.tel_OPD, g[p + "opd_res"][k], atol=1e-18)
                np.testing.assert_allclose(frame, g[p + "frame"][k], atol=1e-8 * g[p + "frame"][k].max())
                assert done is False


@pytest.mark.parametrize("fixture", ["c3_pyr", "c3_pyr_mod3"])
def test_c3_pyramid_full_size(golden_dir, fixture):
    """(c3_pyr_mod3: the same with modulation 3 lambda/D, nTheta = 20 -- OOPAO/Pyramid.py:589-598, 941-985.)
    BASELINE configs[2] at its real size (8 m, 40x40 Pyramid, R = 240, nRes = 528 = 16 * 3 * 11, 1353 actuators): the oracle
    against the reference's Pyramid / Atmosphere / DeformableMirror.  The 1353-poke interaction matrix is pinned through three
    whole columns; the modal command matrix calib.M is then taken from the fixture (pinv of a 2608 x 50 matrix: checked too)."""
    g = np.load(os.path.join(golden_dir, fixture + ".npz"))
    env = _env_from_golden(g, modal_cm=g["modal_cm"])
    assert env.wfs.nTheta == (20 if fixture.endswith("mod3") else 1)
    assert env.R == 240 and env.wfs.nRes == 528 and env.nValidAct == 1353 and env.wfs.nSignal == 2608
    assert np.array_equal(env.pupil, g["pupil"]) and np.array_equal(env.dm_mask.reshape(-1), g["validAct"])
    assert np.array_equal(env.wfs.validI4Q, g["validI4Q"]) and env.wfs.nTheta == int(g["nTheta"])
    np.testing.assert_allclose(env.wfs.referenceSignal_2D, g["referenceSignal_2D"], atol=1e-13)
    np.testing.assert_allclose(env.wfs.m[::7, ::5], g["pyr_m"], atol=1e-13)
    lay = env.atm.layers[0]
    np.testing.assert_allclose(lay.A @ g["A_probe_in"], g["A_probe_out"], atol=1e-9)
    np.testing.assert_allclose(lay.B @ g["B_probe_in"], g["B_probe_out"], atol=1e-9)
    np.testing.assert_allclose(env.dm_modes @ g["modes_probe_in"], g["modes_probe_out"], atol=1e-12)
    stroke = env.wavelength / 16
    for k, a in enumerate(g["imat_cols_idx"]):
        col = g["imat_cols"][:, k]
        np.testing.assert_allclose(env.poke_signal(int(a), stroke), col, atol=1e-12 * np.abs(g["imat_cols"]).max())
    np.testing.assert_allclose(O.calibration_vault_M(g["modal_imat"]), g["modal_cm"], atol=1e-9 * np.abs(g["modal_cm"]).max())
    p = "s17_"
    env.new_episode(17)
    np.testing.assert_allclose(env.atm.layers[0].mapShift, g[p + "mapShift0"][0], atol=1e-12)
    np.testing.assert_allclose(env.reset_soft(), g[p + "obs0"], atol=1e-10)
    full = {int(s): k for k, s in enumerate(g[p + "full_steps"])}
    for i, act in enumerate(g[p + "actions"]):
        obs, frame, rew, sr, done, info = env.step(i, act)
        np.testing.assert_allclose(env.wfs.signal, g[p + "signal"][i], atol=1e-10)
        np.testing.assert_allclose(obs, g[p + "obs"][i], atol=1e-9)
        np.testing.assert_allclose(rew, g[p + "reward"][i], atol=1e-9)
        np.testing.assert_allclose(sr, g[p + "strehl"][i], atol=1e-12)
        np.testing.assert_allclose(env.residual[i], g[p + "residual"][i], atol=1e-9)
        np.testing.assert_allclose(env.coefs, g[p + "coefs"][i], atol=1e-18)
        if i in full:
            k = full[i]
            np.testing.assert_allclose(env.tel_OPD, g[p + "opd_res"][k], atol=1e-18)
            np.testing.assert_allclose(frame, g[p + "frame"][k], atol=1e-8 * g[p + "frame"][k].max())


def test_c4_elt_shack_hartmann_measurement(golden_dir):
    """BASELINE configs[3] geometry (39 m, 80x80 lenslets, R = 480, 5209 actuators): one tel*dm*wfs of the reference on a fixed
    wave-front against the oracle's DM and Shack-Hartmann."""
    from oracle.make_goldens import c4_test_opd
    g = np.load(os.path.join(golden_dir, "c4_sh.npz"))
    R, ns, D = int(g["cfg_R"]), int(g["cfg_nsub"]), float(g["cfg_D"])
    pupil = O.make_pupil(R)
    wl, n_photon = O.source_photometry("I", 8.0)
    flux = pupil.astype(float) * n_photon * (1 / 500) * (D / R) ** 2
    dm = O.dm_geometry(R, D, ns, 0.35, pitch=D / (ns + 1), dense=False)
    assert np.array_equal(dm["validAct"], g["validAct"])
    wfs = O.OracleSH(ns, R, D, pupil, wl, flux)
    assert np.array_equal(wfs.valid_2d, g["valid_subap"])
    np.testing.assert_allclose(wfs.reference_slopes_maps, g["reference_slopes_maps"], atol=1e-13)
    np.testing.assert_allclose(wfs.slopes_units, float(g["slopes_units"]), rtol=1e-11)
    cimg = np.zeros((ns + 1) ** 2)
    cimg[dm["validAct"]] = g["coefs"]
    dm_opd = dm["gy"] @ cimg.reshape(ns + 1, ns + 1) @ dm["gx"].T           # separable form of modes @ coefs
    opd = (c4_test_opd(R) + dm_opd) * pupil
    np.testing.assert_allclose(opd[::16], g["opd_res_rows"], atol=1e-18)
    sig = wfs.measure(opd * 2 * np.pi / wl)
    np.testing.assert_allclose(sig, g["signal"], atol=1e-10)
    np.testing.assert_allclose(wfs.frame[::8], g["frame_rows"], atol=1e-8 * float(g["frame_max"]))
    np.testing.assert_allclose(wfs.frame.sum(axis=0), g["frame_colsum"], rtol=1e-10)
    np.testing.assert_allclose(wfs.frame.sum(axis=1), g["frame_rowsum"], rtol=1e-10)
    # three measurement groups of the reference's 5209-column zonal InteractionMatrix (pokes of a group share the centroid threshold)
    idx, cols = g["imat_idx"], g["imat_cols"]
    stroke = wl / 16
    groups = [idx[idx // 6 == q] for q in np.unique(idx // 6)]
    assert [len(q) for q in groups] == [6, 6, 1]
    for grp in groups:
        phases = []
        for a in grp:
            ci = np.zeros((ns + 1) ** 2)
            ci[np.flatnonzero(dm["validAct"])[a]] = stroke
            phases.append((dm["gy"] @ ci.reshape(ns + 1, ns + 1) @ dm["gx"].T) * 2 * np.pi / wl)
        gmax = max(wfs.spots(ph)[wfs.valid_1d].max() for ph in phases)
        for a, ph in zip(grp, phases):
            col = cols[:, list(idx).index(a)]
            np.testing.assert_allclose(wfs.measure(ph, group_max=gmax) / stroke, col, atol=1e-9 * np.abs(cols).max())


def test_second_episode_keeps_dm_prev(golden_dir):
    """The trainers' episode prologue (MAIN/PO4AO/mbrl.py:49-55) zeroes dm.coefs but not the env's dm_prev
    (MAIN/OOPAOEnv/OOPAOEnv.py:314, 508-509): step 0 of the next episode applies leak * (last command) + action."""
    g = np.load(os.path.join(golden_dir, "two_episodes.npz"))
    env = _env_from_golden(g)
    for tag, seed in (("e1_", 5), ("e2_", 0)):
        env.new_episode(seed)
        np.testing.assert_allclose(env.reset_soft(), g[tag + "obs0"], atol=1e-11)
        for i, act in enumerate(g[tag + "actions"]):
            obs, _, rew, sr, _, _ = env.step(i, act)
            np.testing.assert_allclose(obs, g[tag + "obs"][i], atol=1e-10)
            np.testing.assert_allclose(env.coefs, g[tag + "coefs"][i], atol=1e-18)
            np.testing.assert_allclose(sr, g[tag + "strehl"][i], atol=1e-12)
        np.testing.assert_allclose(env.dm_prev, g[tag + "dm_prev_end"], atol=1e-18)
    assert np.abs(g["e2_coefs"][0]).max() > 1e-8                  # the carried command is really there


def test_detector_matches_reference(golden_dir):
    """oracle.Detector == the reference's OOPAO/Detector.py (integrate + readout) bit for bit when both draw from the same
    RandomStates: Razor camera (photon + dark + read-out noise, QE, saturation, 10-bit ADC) and three partial settings."""
    from numpy.random import RandomState
    g = np.load(os.path.join(golden_dir, "detector.npz"))
    cases = {
        "razor": dict(photonNoise=True, readoutNoise=14, QE=0.56, darkCurrent=5, integrationTime=1 / 500, FWC=10000, bits=10,
                      sensor="CMOS"),
        "photon_only": dict(photonNoise=True, readoutNoise=0, QE=1, darkCurrent=0, integrationTime=None, FWC=None, bits=None,
                            sensor="CCD"),
        "adc_only": dict(photonNoise=False, readoutNoise=0, QE=0.56, darkCurrent=0, integrationTime=None, FWC=10000, bits=10,
                         sensor="CMOS"),
        "readout_only": dict(photonNoise=False, readoutNoise=3.5, QE=0.9, darkCurrent=0, integrationTime=None, FWC=None,
                             bits=None, sensor="CCD"),
    }
    for name, c in cases.items():
        det = O.Detector(**c)
        det.rs_photon, det.rs_readout, det.rs_dark = RandomState(11), RandomState(12), RandomState(13)
        np.testing.assert_array_equal(det.integrate(g["frame"]), g[name], err_msg=name)


def test_science_psf_matches_reference(golden_dir):
    """oracle.telescope_psf == the reference's Telescope.computePSF (incl. its oversampling-2 quirk for even images)."""
    g = np.load(os.path.join(golden_dir, "psf.npz"))
    for zp in (2, 4):
        got = O.telescope_psf(g["pupil"], g["flux_map"], g["phase"], zp)
        np.testing.assert_allclose(got, g[f"psf_zp{zp}"], rtol=0, atol=1e-12 * g[f"psf_zp{zp}"].max())


def test_two_chained_dms_match_reference(golden_dir):
    """tel*dm1*dm2*wfs of the reference (every DM adds its OPD, Telescope.py:533-544) == the oracle's stacked command vector
    [dm1 | dm2] with the stacked influence matrix: residual OPD and Shack-Hartmann signal over three turbulence steps."""
    g = np.load(os.path.join(golden_dir, "two_dm.npz"))
    env = O.OracleEnv(resolution=48, diameter=3.2, n_subap=8, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
                      fractionalR0=[1.0], altitude=[0.0], n_modes=20, second_dm_nsub=int(g["cfg_ns2"]))
    assert env.nValidAct == g["coefs1"].shape[1] + g["coefs2"].shape[1] and env.nActuator == 9 + 5
    for k in range(3):
        cf = np.concatenate([g["coefs1"][k], g["coefs2"][k]])
        opd = (g["opd_atm"][k] + env.dm_opd(cf)) * env.pupil
        np.testing.assert_allclose(opd, g["opd"][k], rtol=0, atol=1e-18)
        sig = env.wfs.measure(opd * 2 * np.pi / env.wavelength)
        np.testing.assert_allclose(sig, g["signal"][k], rtol=0, atol=1e-10)


def test_oracle_env_with_separable_dm_equals_dense():
    """OracleEnv(dm_dense=False) -- the form the ELT-size closed-loop test uses, where dm.modes would be 9.6 GB -- is the same
    loop as the dense one."""
    kw = dict(resolution=24, diameter=1.6, n_subap=4, n_modes=8)
    dense = O.OracleEnv(**kw)
    sep = O.OracleEnv(m2c=dense.M2C, modal_cm=dense.modal_cm, dm_dense=False, **kw)
    cf = np.random.RandomState(2).normal(size=dense.nValidAct) * 1e-7
    np.testing.assert_allclose(sep.dm_opd(cf), dense.dm_opd(cf), atol=1e-21)
    for e in (dense, sep):
        e.new_episode(5)
    od, os_ = dense.reset_soft(), sep.reset_soft()
    for i in range(6):
        od, fd, rd, sd, _, _ = dense.step(i, 0.5 * od)
        os_, fs, rs_, ss, _, _ = sep.step(i, 0.5 * os_)
        np.testing.assert_allclose(os_, od, atol=1e-9)
        np.testing.assert_allclose(ss, sd, atol=1e-12)
