#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own classes (build container only).

    python oracle/make_goldens.py [--only NAME]

Test infrastructure.  The reference (/root/reference) cannot travel to the GPU box, so the vectors
it produces here are committed as small fixtures together with this script.  The driver below
composes the reference objects in the order of ``MAIN/OOPAOEnv/OOPAOEnv.py:93-385`` (set_params) with
the Shack-Hartmann construction of ``MAIN/OOPAOEnv/OOPAOEnvRazor.py:234-238`` and replays
``OOPAOEnv.py:485-536`` (step) with the reference's objects doing all the arithmetic; nothing here
re-implements physics.  See ``oracle/_ref_loader.py`` for the import shims and for the one stage
(the skimage sub-pixel warp) that is not pinned by the reference.
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
sys.path.insert(0, REPO)

from oracle import _ref_loader as RL  # noqa: E402
from oracle import ao_oracle as O      # noqa: E402   (only: Zernike basis for geometries without an M2C file)

GOLD = os.path.join(REPO, "tests", "golden")
MANUAL_M2C = "/root/reference/drl4ao/MAIN_CODE/OOPAOEnv/manual_m2c.npy"

CASES = {
    # name: geometry + atmosphere + loop
    "tiny_sh": dict(R=24, nsub=4, D=1.6, r0=0.13, L0=30.0, ws=[10.0], wd=[72.0], frac=[1.0], alt=[0.0],
                    n_modes=8, steps=40, seeds=[0, 17], gain=0.5, full_every=1),
    "tiny_fastwind": dict(R=24, nsub=4, D=1.6, r0=0.13, L0=30.0, ws=[50.0], wd=[200.0], frac=[1.0], alt=[0.0],
                          n_modes=8, steps=12, seeds=[3], gain=0.4, full_every=1),
    "tiny_3layer": dict(R=24, nsub=4, D=1.6, r0=0.13, L0=30.0, ws=[10.0, 12.0, 11.0], wd=[0.0, 72.0, 144.0],
                        frac=[0.45 / 0.65, 0.1 / 0.65, 0.1 / 0.65], alt=[0.0, 0.0, 0.0],
                        n_modes=8, steps=30, seeds=[17], gain=0.5, full_every=1),
    # the reference env's own telescope has fov = 1 arcsec (MAIN/OOPAOEnv/OOPAOEnv.py:129): a layer at altitude h then lives on a grid of
    # ceil(R / D (D + 2 tan(fov / 2) h)) + 4 pixels with ring operators of its own (OOPAO/Atmosphere.py:216-218): 28, 29, 29 here
    "tiny_3layer_fov1": dict(R=24, nsub=4, D=1.6, r0=0.13, L0=30.0, ws=[10.0, 12.0, 11.0], wd=[0.0, 72.0, 144.0],
                             frac=[0.45 / 0.65, 0.1 / 0.65, 0.1 / 0.65], alt=[0.0, 1000.0, 5000.0], fov=1.0,
                             n_modes=8, steps=30, seeds=[17], gain=0.5, full_every=1),
    "small_sh": dict(R=48, nsub=8, D=3.2, r0=0.13, L0=30.0, ws=[10.0], wd=[72.0], frac=[1.0], alt=[0.0],
                     n_modes=20, steps=50, seeds=[0, 17], gain=0.5, full_every=10),
    "c2_sh": dict(R=120, nsub=20, D=8.0, r0=0.13, L0=30.0, ws=[10.0], wd=[72.0], frac=[1.0], alt=[0.0],
                  n_modes=50, steps=20, seeds=[17], gain=0.5, full_every=10, m2c_file=True),
    # Pyramid WFS (MAIN/OOPAOEnv/OOPAOEnv.py:239-246): unmodulated / centred mask, and modulated / one-pixel mask
    "tiny_pyr": dict(R=24, nsub=4, D=1.6, r0=0.13, L0=30.0, ws=[10.0], wd=[72.0], frac=[1.0], alt=[0.0],
                     n_modes=8, steps=30, seeds=[0, 17], gain=0.5, full_every=1, wfs="pyr", modulation=0, centering=True),
    "tiny_pyr_mod": dict(R=24, nsub=4, D=1.6, r0=0.13, L0=30.0, ws=[10.0], wd=[72.0], frac=[1.0], alt=[0.0],
                         n_modes=8, steps=20, seeds=[17], gain=0.5, full_every=1, wfs="pyr", modulation=2, centering=False),
    # the reference's own Papyrus configuration (MAIN/Conf/papyrus_config.yaml + parameterFile_oopao_parser.py)
    "papyrus_pyr": dict(R=120, nsub=20, D=1.52, r0=0.25, L0=10.0, ws=[20.0], wd=[72.0], frac=[1.0], alt=[0.0],
                        n_modes=50, steps=12, seeds=[0], gain=0.5, full_every=6, m2c_file=True, wfs="pyr", modulation=0,
                        centering=True),
    # BASELINE.json configs[2] at its real size: 8 m, 40x40 Pyramid (R = 240, nRes = 528 = 16 * 3 * 11), unmodulated, the env's
    # default (centred) mask, 50 Zernike modes as OOPAOEnv.set_params keeps (SURVEY.md 8c "R=240/40 sub Pyr single step")
    "c3_pyr": dict(R=240, nsub=40, D=8.0, r0=0.13, L0=30.0, ws=[10.0], wd=[72.0], frac=[1.0], alt=[0.0],
                   n_modes=50, steps=4, seeds=[17], gain=0.5, full_every=3, wfs="pyr", modulation=0, centering=True),
    # the same geometry with the modulation SURVEY.md 8a A6 / 8d quote (3 lambda/D: nTheta = 20, Pyramid.py:941-985): calibration at
    # that modulation, one measurement + closed-loop steps; every step recorded in full (88 x 88 frame, 2608 signals)
    "c3_pyr_mod3": dict(R=240, nsub=40, D=8.0, r0=0.13, L0=30.0, ws=[10.0], wd=[72.0], frac=[1.0], alt=[0.0],
                        n_modes=50, steps=3, seeds=[17], gain=0.5, full_every=1, wfs="pyr", modulation=3, centering=True),
    # BASELINE.json configs[4] per-env physics: 3 layers (the first three of papyrus_config.yaml's commented profile, Cn2
    # renormalised) + two chained DMs (20x20 and 10x10 pitch, the second one altitude-conjugated at 5000 m:
    # with fov = 0 it has the ground DM's grid, OOPAO/DeformableMirror.py:388-389), tel*dm1*dm2*wfs in closed loop
    "c5_mcao": dict(R=120, nsub=20, D=8.0, r0=0.13, L0=30.0, ws=[10.0, 12.0, 11.0], wd=[0.0, 72.0, 144.0],
                    frac=[0.45 / 0.65, 0.1 / 0.65, 0.1 / 0.65], alt=[0.0, 1000.0, 5000.0],
                    n_modes=50, steps=12, seeds=[17], gain=0.5, full_every=6, second_nsub=10, second_alt=5000.0),
}


def build(ref, c):
    """Reference objects, composed as OOPAOEnv.set_params does (with an SH WFS)."""
    with RL.quiet():
        tel = ref.Telescope(resolution=c["R"], diameter=c["D"], samplingTime=1 / 500, centralObstruction=0,
                            display_optical_path=False, fov=c.get("fov", 0))
        ngs = ref.Source(optBand="I", magnitude=8, coordinates=[0, 0])
        ngs * tel
        atm = ref.Atmosphere(telescope=tel, r0=c["r0"], L0=c["L0"], windSpeed=list(c["ws"]),
                             fractionalR0=list(c["frac"]), windDirection=list(c["wd"]), altitude=list(c["alt"]))
        atm.initializeAtmosphere(tel)
        atm.update()
        nAct = c["nsub"] + 1
        dm = ref.DeformableMirror(telescope=tel, nSubap=c["nsub"], mechCoupling=0.35, coordinates=None,
                                  pitch=tel.D / nAct)
        dm2 = None
        if c.get("second_nsub"):
            n2 = c["second_nsub"] + 1
            dm2 = ref.DeformableMirror(telescope=tel, nSubap=c["second_nsub"], mechCoupling=0.35, coordinates=None,
                                       pitch=tel.D / n2, altitude=c.get("second_alt"))
        tel.isPaired = False
        tel.resetOPD()
        if c.get("wfs", "sh") == "sh":
            wfs = ref.ShackHartmann(telescope=tel, nSubap=c["nsub"], lightRatio=0.5, is_geometric=False,
                                    shannon_sampling=True)
        else:
            wfs = ref.Pyramid(nSubap=c["nsub"], telescope=tel, lightRatio=0.1, modulation=c["modulation"], binning=1,
                              n_pix_separation=4, n_pix_edge=2, postProcessing="slopesMaps_incidence_flux",
                              psfCentering=c["centering"])
        tel * wfs
        if c.get("m2c_file"):
            m2c = np.load(MANUAL_M2C)[:, :c["n_modes"]]                    # OOPAOEnv.py:260
        else:
            # OOPAOEnv.py:258 / OOPAOEnvRazor.py:261 with the Noll basis restated in the oracle (aotools absent);
            # two DMs: the stacked influence matrix [dm1 | dm2]
            Z = O.zernike_modes(tel.pupil > 0, tel.D, c["n_modes"])
            modes = dm.modes if dm2 is None else np.hstack([dm.modes, dm2.modes])
            m2c = np.linalg.pinv(np.squeeze(modes[tel.pupilLogical, :])) @ Z
            del modes
        tel - atm
        # every DM is calibrated by its own InteractionMatrix call (an un-paired telescope takes ONLY the last DM's OPD,
        # OOPAO/DeformableMirror.py:474-476, so chained DMs cannot be poked in one call); D = [D1 | D2]
        Ds = []
        for d in ([dm] if dm2 is None else [dm, dm2]):
            calib = ref.InteractionMatrix(ngs=ngs, atm=atm, tel=tel, dm=d, wfs=wfs, M2C=np.eye(d.nValidAct),
                                          stroke=ngs.wavelength / 16, nMeasurements=6, noise="off", display=False,
                                          single_pass=True)
            Ds.append(np.asarray(calib.D))
        Dfull = np.hstack(Ds)
        vault = ref.CalibrationVault(Dfull @ m2c)
        tel.resetOPD()
        dm.coefs = 0
        if dm2 is not None:
            dm2.coefs = 0
        ngs * tel * dm * wfs
        atm.generateNewPhaseScreen(seed=10)
        tel + atm
    recon = m2c @ vault.M
    F = m2c @ np.linalg.pinv(m2c)
    if dm2 is None:
        dm_mask = np.reshape(dm.validAct, (nAct, nAct))
    else:                                 # actuator image of the pair: dm1's grid top-left, dm2's bottom-right (calib.CompositeDM)
        n2 = dm2.nAct
        dm_mask = np.zeros((nAct + n2, nAct + n2), bool)
        dm_mask[:nAct, :nAct] = np.reshape(dm.validAct, (nAct, nAct))
        dm_mask[nAct:, nAct:] = np.reshape(dm2.validAct, (n2, n2))
        nAct = nAct + n2
    xv, yv = np.nonzero(dm_mask)
    return dict(tel=tel, ngs=ngs, atm=atm, dm=dm, dm2=dm2, wfs=wfs, m2c=m2c, D=Dfull, modal_cm=np.asarray(vault.M), recon=recon, F=F,
                xvalid=xv, yvalid=yv, nAct=nAct, dm_mask=dm_mask)


class _Coefs:
    """dm.coefs of one mirror, or the stacked vector [dm1 | dm2] of two chained ones (bookkeeping only)."""

    def __init__(self, dm, dm2=None):
        self.dm, self.dm2 = dm, dm2
        self.nValidAct = dm.nValidAct + (dm2.nValidAct if dm2 is not None else 0)

    def set(self, v):
        if self.dm2 is None:
            self.dm.coefs = v
        elif np.isscalar(v):
            self.dm.coefs = v
            self.dm2.coefs = v
        else:
            self.dm.coefs = np.asarray(v)[:self.dm.nValidAct]
            self.dm2.coefs = np.asarray(v)[self.dm.nValidAct:]

    def get(self):
        if self.dm2 is None:
            return np.asarray(self.dm.coefs).copy()
        return np.concatenate([np.asarray(self.dm.coefs).reshape(-1), np.asarray(self.dm2.coefs).reshape(-1)])

    def propagate(self, tel, wfs):
        if self.dm2 is None:
            tel * self.dm * wfs
        else:
            tel * self.dm * self.dm2 * wfs                        # paired telescope: every DM adds its OPD (Telescope.py:533-544)


def _stack_maps(atm):
    """layer.mapShift of every layer as one array [nLayer, S, S]; layers on smaller grids (fov != 0: the grid grows with the
    altitude) sit in the top-left corner of an S = max(S_l) frame, zero beyond (cfg_layer_S holds every S_l)."""
    maps = [getattr(atm, f"layer_{i + 1}").mapShift for i in range(atm.nLayer)]
    S = max(m.shape[0] for m in maps)
    out = np.zeros((len(maps), S, S))
    for l, m in enumerate(maps):
        out[l, :m.shape[0], :m.shape[1]] = m
    return out


def run_episode(env, c, seed, dm_prev_in=None):
    """mbrl.py:49-55 prologue + integrator closed loop through OOPAOEnv.step arithmetic.  dm_prev_in: the env's dm_prev as
    the previous episode left it (the prologue's ``dm.coefs = 0`` does not clear it, OOPAOEnv.py:314, 508-509); None = a
    fresh env (dm_prev = 0 from set_params)."""
    tel, atm, wfs = env["tel"], env["atm"], env["wfs"]
    dm = _Coefs(env["dm"], env.get("dm2"))
    nAct, xv, yv, recon = env["nAct"], env["xvalid"], env["yvalid"], env["recon"]
    leak = 0.99

    def vec_to_img(v):
        img = np.zeros((nAct, nAct))
        img[xv, yv] = v
        return img

    with RL.quiet():
        atm.generateNewPhaseScreen(seed)
        dm.set(0)
        dm.propagate(tel, wfs)
    dm_prev = dm.get() if dm_prev_in is None else np.asarray(dm_prev_in).copy()
    obs0 = vec_to_img(-np.matmul(recon, wfs.signal)) * 1e6          # reset_soft
    T = c["steps"]
    R = c["R"]
    out = dict(obs0=obs0, mapShift0=_stack_maps(atm),
               signal0=wfs.signal.copy(),
               actions=np.zeros((T, nAct, nAct), np.float32), obs=np.zeros((T, nAct, nAct)),
               reward=np.zeros(T), strehl=np.zeros(T), total=np.zeros(T), residual=np.zeros(T),
               signal=np.zeros((T, wfs.nSignal)), coefs=np.zeros((T, dm.nValidAct)),
               buff=np.zeros((T, atm.nLayer, 2)), full_steps=[], opd_atm=[], opd_res=[], frame=[], mapShift=[])
    obs = obs0
    for i in range(T):
        action = np.float32(c["gain"] * np.float32(obs))             # TorchWrapper: f32 obs, f32 action
        out["actions"][i] = action
        a = action[xv, yv] * 1e-6                                    # img_to_vec(action)*1e-6   :491
        with RL.quiet():
            atm.update()                                             # :494
        total = np.std(tel.OPD[np.where(tel.pupil > 0)]) * 1e9       # :497
        opd_atm = tel.OPD.copy()
        with RL.quiet():
            dm.propagate(tel, wfs)                                   # :503
            dm.set((dm_prev * leak) + a)                             # :508
        dm_prev = dm.get()
        obs = vec_to_img(-np.matmul(recon, wfs.signal)) * 1e6        # :517-518
        out["obs"][i] = obs
        out["reward"][i] = -1 * np.linalg.norm(obs)                  # :536
        out["strehl"][i] = np.exp(-np.var(tel.src.phase[np.where(tel.pupil == 1)]))   # :554-555
        out["total"][i] = total
        out["residual"][i] = np.std(tel.OPD[np.where(tel.pupil > 0)]) * 1e9           # :522
        out["signal"][i] = wfs.signal
        out["coefs"][i] = dm.get()
        for l in range(atm.nLayer):
            out["buff"][i, l] = getattr(atm, f"layer_{l + 1}").buff
        if i % c["full_every"] == 0 or i == T - 1:
            out["full_steps"].append(i)
            out["opd_atm"].append(opd_atm)
            out["opd_res"].append(tel.OPD.copy())
            out["frame"].append(wfs.cam.frame.copy())
            out["mapShift"].append(_stack_maps(atm))
    for k in ("opd_atm", "opd_res", "frame", "mapShift", "full_steps"):
        out[k] = np.asarray(out[k])
    assert opd_atm.shape == (R, R)
    out["dm_prev_end"] = dm_prev
    return out


def make_case(ref, name):
    c = CASES[name]
    env = build(ref, c)
    tel, atm, dm, wfs = env["tel"], env["atm"], env["dm"], env["wfs"]
    dm2 = env.get("dm2")
    L1 = atm.layer_1
    big = env["D"].size > 400000          # interaction matrix beyond ~3 MB: keep columns / modal products instead
    consts = dict(
        cfg_R=c["R"], cfg_nsub=c["nsub"], cfg_D=c["D"], cfg_r0=c["r0"], cfg_L0=c["L0"],
        cfg_ws=np.array(c["ws"]), cfg_wd=np.array(c["wd"]), cfg_frac=np.array(c["frac"]), cfg_alt=np.array(c["alt"]),
        cfg_gain=c["gain"], cfg_n_modes=c["n_modes"], cfg_seeds=np.array(c["seeds"]),
        pupil=tel.pupil.astype(bool), wavelength=env["ngs"].wavelength, nPhoton=env["ngs"].nPhoton,
        validAct=np.asarray(env["dm_mask"], bool).reshape(-1),
        m2c=env["m2c"], F=env["F"], xvalid=env["xvalid"], yvalid=env["yvalid"])
    if c.get("fov"):
        layers = [getattr(atm, f"layer_{i + 1}") for i in range(atm.nLayer)]
        consts.update(cfg_fov=float(c["fov"]), cfg_layer_S=np.array([l.mapShift.shape[0] for l in layers]))
        for i, l in enumerate(layers[1:], start=1):                   # (layer 0's operators are `A`, `B` below)
            consts.update({f"A_l{i}": l.A, f"B_l{i}": l.B})
    if dm2 is not None:
        consts.update(cfg_second_nsub=c["second_nsub"], cfg_second_alt=float(c.get("second_alt") or 0.0),
                      validAct1=np.asarray(dm.validAct, bool), validAct2=np.asarray(dm2.validAct, bool))
    if not big:
        consts.update(imat=env["D"], recon=env["recon"])
    else:
        # three whole columns (first, middle, last poke), the modal interaction matrix D @ M2C and the modal command
        # matrix calib.M = pinv_svd(D @ M2C): with the M2C above they determine the reconstructor M2C @ calib.M
        cols = np.array([0, env["D"].shape[1] // 2, env["D"].shape[1] - 1])
        rsq = np.random.RandomState(321)
        pa = rsq.randn(env["D"].shape[1], 2)
        consts.update(imat_cols_idx=cols, imat_cols=env["D"][:, cols], imat_probe_in=pa, imat_probe_out=env["D"] @ pa,
                      imat_fro=np.linalg.norm(env["D"]), modal_imat=env["D"] @ env["m2c"], modal_cm=env["modal_cm"])
    if c.get("wfs", "sh") == "sh":
        consts.update(valid_subap=np.asarray(wfs.valid_subapertures, bool),
                      reference_slopes_maps=wfs.reference_slopes_maps, slopes_units=wfs.slopes_units)
    else:
        consts.update(cfg_wfs="pyr", cfg_modulation=c["modulation"], cfg_centering=c["centering"],
                      validI4Q=np.asarray(wfs.validI4Q, bool), referenceSignal_2D=wfs.referenceSignal_2D,
                      pyr_m=wfs.m if c["R"] <= 48 else wfs.m[::7, ::5], nTheta=wfs.nTheta)
    if c["R"] <= 48:
        consts.update(A=L1.A, B=L1.B, modes=dm.modes)
    else:
        # large constants: probes + moments instead of the full arrays
        rs = np.random.RandomState(123)
        pz = rs.randn(L1.A.shape[1])
        px = rs.randn(L1.B.shape[1])
        pc = rs.randn(dm.nValidAct)
        consts.update(A_probe_in=pz, A_probe_out=L1.A @ pz, B_probe_in=px, B_probe_out=L1.B @ px,
                      A_fro=np.linalg.norm(L1.A), B_fro=np.linalg.norm(L1.B), A_shape=np.array(L1.A.shape),
                      modes_probe_in=pc, modes_probe_out=dm.modes @ pc, modes_fro=np.linalg.norm(dm.modes))
        if dm2 is not None:
            pc2 = rs.randn(dm2.nValidAct)
            consts.update(modes2_probe_in=pc2, modes2_probe_out=dm2.modes @ pc2, modes2_fro=np.linalg.norm(dm2.modes))
    if name == "c2_sh":
        # the ring operators of the headline geometry in full, in a file of their own (5.9 MB): with them injected the float64 shards
        # are compared at the same-operator tolerance (1e-12) at one FULL size (tests/test_gpu_parity.py, "f64-refAB")
        np.savez_compressed(os.path.join(GOLD, "c2_sh_AB.npz"), A=L1.A, B=L1.B)
    out = dict(consts)
    if c["R"] > 48:
        out.pop("F")                       # derived from the m2c input alone; keep the fixture small
    for seed in c["seeds"]:
        ep = run_episode(env, c, seed)
        if c["R"] > 48:
            ep.pop("mapShift")
        for k, v in ep.items():
            out[f"s{seed}_{k}"] = v
    path = os.path.join(GOLD, f"{name}.npz")
    np.savez_compressed(path, **out)
    s0 = c["seeds"][0]
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)  "
          f"nValidAct={env['D'].shape[1]} nSignal={wfs.nSignal} "
          f"strehl_last={out['s%d_strehl' % s0][-1]:.4f} res_last={out['s%d_residual' % s0][-1]:.1f} nm", flush=True)


def make_two_episodes(ref):
    """Two consecutive episodes of ONE env (tiny_sh geometry) with the trainers' prologue between them
    (MAIN/PO4AO/mbrl.py:49-55: generateNewPhaseScreen, dm.coefs = 0, tel*dm*wfs, reset_soft): the env's dm_prev is NOT
    cleared by that prologue (OOPAOEnv.py:314, 508-509), so step 0 of the second episode applies
    leak * (last command of the first episode) + action."""
    c = dict(CASES["tiny_sh"], steps=8)
    env = build(ref, c)
    ep1 = run_episode(env, c, 5)
    ep2 = run_episode(env, c, 0, dm_prev_in=ep1["dm_prev_end"])
    out = {"cfg_R": c["R"], "cfg_nsub": c["nsub"], "cfg_D": c["D"], "cfg_r0": c["r0"], "cfg_L0": c["L0"],
           "cfg_ws": np.array(c["ws"]), "cfg_wd": np.array(c["wd"]), "cfg_frac": np.array(c["frac"]), "cfg_alt": np.array(c["alt"]),
           "cfg_n_modes": c["n_modes"], "cfg_seeds": np.array([5, 0]), "m2c": env["m2c"]}
    for tag, ep in (("e1_", ep1), ("e2_", ep2)):
        for k in ("obs0", "actions", "obs", "reward", "strehl", "signal", "coefs", "residual", "total", "dm_prev_end"):
            out[tag + k] = ep[k]
    path = os.path.join(GOLD, "two_episodes.npz")
    np.savez_compressed(path, **out)
    print(f"two_episodes: wrote {path}; |dm_prev| carried into episode 2 = {np.abs(ep1['dm_prev_end']).max():.3e} m", flush=True)


def c4_test_opd(R, seed=31):
    """The wave-front of the ELT-size component fixture: smooth aberration + white roughness, metres (no pupil).  Also
    imported by the tests, so that the input does not have to be stored."""
    rs = np.random.RandomState(seed)
    y, x = np.mgrid[0:R, 0:R] / float(R)
    return (900e-9 * (x - 0.4) ** 2 - 700e-9 * x * y + 500e-9 * np.sin(7 * x + 3 * y) * (y - 0.5)
            + 60e-9 * rs.normal(size=(R, R)))


def make_c4_sh(ref):
    """BASELINE.json configs[3] geometry (39 m, 80x80 Shack-Hartmann, R = 480, 81x81 actuators): the reference's own
    Telescope / DeformableMirror / ShackHartmann on one fixed wave-front (c4_test_opd + a DM command): valid lenslets and
    actuators, reference slopes, slope units, the residual OPD, wfs.signal and the camera frame of ONE measurement."""
    R, ns, D = 480, 80, 39.0
    with RL.quiet():
        tel = ref.Telescope(resolution=R, diameter=D, samplingTime=1 / 500, centralObstruction=0, display_optical_path=False, fov=0)
        ngs = ref.Source(optBand="I", magnitude=8, coordinates=[0, 0])
        ngs * tel
        dm = ref.DeformableMirror(telescope=tel, nSubap=ns, mechCoupling=0.35, coordinates=None, pitch=tel.D / (ns + 1))
        wfs = ref.ShackHartmann(telescope=tel, nSubap=ns, lightRatio=0.5, is_geometric=False, shannon_sampling=True)
    rs = np.random.RandomState(77)
    coefs = rs.normal(size=dm.nValidAct) * 120e-9
    opd_in = c4_test_opd(R)
    with RL.quiet():
        dm.coefs = coefs
        tel.isPaired = True                                       # a paired telescope adds dm.OPD to OPD_no_pupil (DeformableMirror.py:469)
        tel.OPD_no_pupil = opd_in.copy()
        tel.OPD = opd_in * tel.pupil
        tel * dm * wfs
    frame = np.asarray(wfs.cam.frame, dtype=np.float64)
    signal = np.asarray(wfs.signal, dtype=np.float64).copy()
    opd_res = np.asarray(tel.OPD, dtype=np.float64).copy()
    # Three measurement groups of the zonal interaction matrix (OOPAOEnv.py:278-288: M2C = identity, stroke = lambda / 16,
    # nMeasurements = 6, single pass): the first, a middle and the last group of consecutive actuators, each measured by the
    # reference's InteractionMatrix exactly as a cycle of the full 5209-column call measures it (a cycle's pokes share the
    # centroid threshold, ShackHartmann.py:605-672; the last cycle holds the 5209 % 6 = 1 last actuator).
    nv = int(dm.nValidAct)
    imat_idx, imat_cols = [], []
    for gidx in (0, (nv // 6) // 2, nv // 6):
        lo, hi = 6 * gidx, min(6 * gidx + 6, nv)
        m2c = np.zeros((nv, hi - lo))
        m2c[np.arange(lo, hi), np.arange(hi - lo)] = 1.0
        with RL.quiet():
            cal = ref.InteractionMatrix(ngs=ngs, atm=None, tel=tel, dm=dm, wfs=wfs, M2C=m2c, stroke=ngs.wavelength / 16,
                                        nMeasurements=6, noise="off", invert=False, display=False, single_pass=True)
        imat_idx += list(range(lo, hi))
        imat_cols.append(np.asarray(cal.D, dtype=np.float64).reshape(wfs.nSignal, hi - lo))
        print(f"c4_sh: interaction-matrix group {gidx} (actuators {lo}..{hi - 1}) measured", flush=True)
    out = dict(cfg_R=R, cfg_nsub=ns, cfg_D=D, coefs=coefs, imat_idx=np.asarray(imat_idx, dtype=np.int64), imat_cols=np.hstack(imat_cols), validAct=np.asarray(dm.validAct, bool),
               valid_subap=np.asarray(wfs.valid_subapertures, bool), reference_slopes_maps=wfs.reference_slopes_maps,
               slopes_units=wfs.slopes_units, signal=signal,
               opd_res_rows=opd_res[::16], frame_rows=frame[::8],
               frame_sum=frame.sum(), frame_max=frame.max(), frame_sq=np.sqrt((frame ** 2).sum()),
               frame_colsum=frame.sum(axis=0), frame_rowsum=frame.sum(axis=1))
    path = os.path.join(GOLD, "c4_sh.npz")
    np.savez_compressed(path, **out)
    print(f"c4_sh: wrote {path} ({os.path.getsize(path) / 1e6:.2f} MB)  nValidAct={dm.nValidAct} nSignal={wfs.nSignal}", flush=True)


def make_detector(ref):
    """The reference's Detector (OOPAO/Detector.py) on a fixed synthetic frame, its four wall-clock seeded RandomStates
    replaced by known seeds: Razor camera settings (MAIN/OOPAOEnv/OOPAOEnvRazor.py:243-250, 333) and three partial ones."""
    from numpy.random import RandomState
    rs = RandomState(5)
    frame = rs.gamma(0.6, 400.0, size=(24, 24))                 # photons per pixel: mostly faint, a few bright spots
    frame[3, 4] = 30000.0                                        # one pixel beyond the full-well capacity
    out = {"frame": frame}
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
        with RL.quiet():
            det = ref.Detector(24, integrationTime=c["integrationTime"], bits=c["bits"], FWC=c["FWC"], sensor=c["sensor"],
                               QE=c["QE"], darkCurrent=c["darkCurrent"], readoutNoise=c["readoutNoise"],
                               photonNoise=c["photonNoise"])
        det.random_state_photon_noise = RandomState(11)
        det.random_state_readout_noise = RandomState(12)
        det.random_state_dark_shot_noise = RandomState(13)
        if c["integrationTime"] is not None:
            det._integrated_time = c["integrationTime"]          # one frame per read-out, as wfs*cam does each loop step
        with RL.quiet():
            det.integrate(frame.copy())
        out[name] = np.asarray(det.frame, dtype=np.float64)
    path = os.path.join(GOLD, "detector.npz")
    np.savez_compressed(path, **out)
    print(f"detector: wrote {path}")


def make_psf(ref):
    """The reference's Telescope.computePSF (OOPAO/Telescope.py:258-357) of a smooth + random residual phase, R = 24,
    zero-padding 2 and 4."""
    from numpy.random import RandomState
    with RL.quiet():
        tel = ref.Telescope(resolution=24, diameter=1.6, samplingTime=1 / 500, centralObstruction=0)
        ngs = ref.Source(optBand="I", magnitude=8)
        ngs * tel
    rs = RandomState(8)
    y, x = np.mgrid[0:24, 0:24] / 24.0
    opd = (300e-9 * (x - 0.3) ** 2 - 200e-9 * x * y + 40e-9 * rs.normal(size=(24, 24)))
    tel.OPD = opd * tel.pupil                                         # sets tel.src.phase = OPD * 2 pi / lambda
    out = {"pupil": np.asarray(tel.pupil, dtype=np.float64), "flux_map": np.asarray(tel.src.fluxMap, dtype=np.float64),
           "phase": np.asarray(tel.src.phase, dtype=np.float64)}
    for zp in (2, 4):
        with RL.quiet():
            tel.computePSF(zp)
        out[f"psf_zp{zp}"] = np.asarray(tel.PSF, dtype=np.float64)
    path = os.path.join(GOLD, "psf.npz")
    np.savez_compressed(path, **out)
    print(f"psf: wrote {path}")


def make_two_dm(ref):
    """Two chained deformable mirrors, tel*dm1*dm2*wfs (OOPAO/Telescope.py:533-544 adds each dm.OPD to tel.OPD_no_pupil while
    the telescope is paired to the atmosphere): an 8x8 and a 4x4 actuator-pitch DM on the small_sh geometry, three steps of
    turbulence, random commands on both mirrors."""
    from numpy.random import RandomState
    c = CASES["small_sh"]
    env = build(ref, c)
    tel, atm, dm1, wfs = env["tel"], env["atm"], env["dm"], env["wfs"]
    ns2 = 4
    with RL.quiet():
        dm2 = ref.DeformableMirror(telescope=tel, nSubap=ns2, mechCoupling=0.35, coordinates=None, pitch=tel.D / (ns2 + 1))
        atm.generateNewPhaseScreen(seed=21)
        tel + atm
    rs = RandomState(4)
    out = {"cfg_ns2": ns2, "valid2": np.asarray(dm2.validAct, dtype=bool), "modes2_probe": np.asarray(dm2.modes[::37], dtype=np.float64)}
    c1s, c2s, opds, sigs, atms = [], [], [], [], []
    for k in range(3):
        c1 = rs.normal(size=dm1.nValidAct) * 80e-9
        c2 = rs.normal(size=dm2.nValidAct) * 150e-9
        with RL.quiet():
            atm.update()
            atms.append(np.asarray(tel.OPD_no_pupil, dtype=np.float64).copy())
            dm1.coefs = c1
            dm2.coefs = c2
            tel * dm1 * dm2 * wfs
        c1s.append(c1); c2s.append(c2)
        opds.append(np.asarray(tel.OPD, dtype=np.float64).copy())
        sigs.append(np.asarray(wfs.signal, dtype=np.float64).copy())
    out.update(coefs1=np.stack(c1s), coefs2=np.stack(c2s), opd_atm=np.stack(atms), opd=np.stack(opds), signal=np.stack(sigs))
    path = os.path.join(GOLD, "two_dm.npz")
    np.savez_compressed(path, **out)
    print(f"two_dm: wrote {path}  (A1 = {dm1.nValidAct}, A2 = {dm2.nValidAct})")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    ref = RL.load()
    for name in CASES:
        if args.only and name != args.only:
            continue
        make_case(ref, name)
    if args.only in (None, "detector"):
        make_detector(ref)
    if args.only in (None, "psf"):
        make_psf(ref)
    if args.only in (None, "two_dm"):
        make_two_dm(ref)
    if args.only in (None, "two_episodes"):
        make_two_episodes(ref)
    if args.only in (None, "c4_sh"):
        make_c4_sh(ref)


if __name__ == "__main__":
    main()
