"""Diagnostic (not a test): replay a golden on the GPU and print error statistics per quantity."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlao_amd.env import BatchedAOEnv

name, dtype = sys.argv[1], sys.argv[2]
g = np.load(f"tests/golden/{name}.npz")
seeds = [int(s) for s in g["cfg_seeds"]]
prm = dict(diameter=float(g["cfg_D"]), nSubaperture=int(g["cfg_nsub"]), nPixelPerSubap=int(g["cfg_R"]) // int(g["cfg_nsub"]),
           r0=float(g["cfg_r0"]), L0=float(g["cfg_L0"]), windSpeed=list(g["cfg_ws"]), windDirection=list(g["cfg_wd"]),
           fractionnalR0=list(g["cfg_frac"]), altitude=list(g["cfg_alt"]), nModes=int(g["cfg_n_modes"]), nLoop=64)
env = BatchedAOEnv(n_envs=len(seeds), device=0, dtype=dtype)
pyr = "cfg_wfs" in g
if pyr:
    prm.update(modulation=float(g["cfg_modulation"]), psfCentering=bool(g["cfg_centering"]))
env.set_params(prm, camera="ideal", wfs_type="pyramid" if pyr else "shackhartmann", m2c=g["m2c"])
print("units", env.slopes_units, float(g["slopes_units"]) if "slopes_units" in g else 1.0, "imat relerr", np.abs(env.imat - g["imat"]).max() / np.abs(g["imat"]).max(),
      "recon relerr", np.abs(env.reconstructor - g["recon"]).max() / np.abs(g["recon"]).max())
env.env_seed_stride = (seeds[1] - seeds[0]) if len(seeds) > 1 else 1
env.generate_new_phase_screen(seeds[0])
env.dm.coefs = 0
env.measure()
obs0 = env.reset_soft().cpu().numpy()
S = env._atm_tables.S
scr = env._shard.download(0, (env.param.nLayer, env.n_envs, S, S))
for k, s in enumerate(seeds):
    d = np.abs(scr[:, k] - g[f"s{s}_mapShift0"])
    w = np.unravel_index(np.argmax(d), d.shape)
    print(f"seed {s}: obs0 err {np.abs(obs0[k]-g[f's{s}_obs0']).max():.3e}  mapShift0 err max {d.max():.3e} at {w} n>1e-11: {(d>1e-11).sum()}")
T = len(g[f"s{seeds[0]}_actions"])
E = {}
def upd(k, v): E[k] = max(E.get(k, 0.0), float(v))
for i in range(T):
    act = torch.as_tensor(np.stack([g[f"s{s}_actions"][i] for s in seeds]))
    obs, frame, rew, sr, done, info = env.step(i, act)
    obs, frame, rew, sr = obs.cpu().numpy(), frame.cpu().numpy(), rew.cpu().numpy(), sr.cpu().numpy()
    sig = env._shard.download(5, (env.n_envs, env.nSignal))
    coefs = env._shard.download(2, (env.n_envs, env.nValidAct))
    buff = env._shard.get_buff(env.param.nLayer)
    for k, s in enumerate(seeds):
        p = f"s{s}_"
        upd("signal", np.abs(sig[k] - g[p + "signal"][i]).max()); upd("obs", np.abs(obs[k] - g[p + "obs"][i]).max())
        upd("reward", abs(rew[k] - g[p + "reward"][i])); upd("strehl", abs(sr[k] - g[p + "strehl"][i]))
        upd("coefs_rel", np.abs(coefs[k] - g[p + "coefs"][i]).max() / np.abs(g[p + "coefs"][i]).max())
        upd("buff", np.abs(buff - g[p + "buff"][i]).max())
        full = {int(t): q for q, t in enumerate(g[p + "full_steps"])}
        if i in full:
            q = full[i]
            opd_atm = env._shard.download(1, (env.n_envs, env.R, env.R))[k]
            phase = env._shard.download(3, (env.n_envs, env.R, env.R))[k]
            upd("opd_atm_m", np.abs(opd_atm * env.pupil - g[p + "opd_atm"][q]).max())
            upd("opd_res_m", np.abs(phase * env.src_wavelength / (2 * np.pi) - g[p + "opd_res"][q]).max())
            upd("frame_rel", np.abs(frame[k] - g[p + "frame"][q]).max() / g[p + "frame"][q].max())
            if p + "mapShift" in g:
                sc = env._shard.download(0, (env.param.nLayer, env.n_envs, S, S))[:, k]
                upd("mapShift", np.abs(sc - g[p + "mapShift"][q]).max())
    if i < 3 or i == T - 1:
        print(i, {k: f"{v:.2e}" for k, v in E.items()})
tot, res = env.total, env.residual
if env.n_envs == 1: tot, res = tot[:, None], res[:, None]
for k, s in enumerate(seeds):
    upd("total_nm", np.abs(tot[:T, k] - g[f"s{s}_total"]).max()); upd("residual_nm", np.abs(res[:T, k] - g[f"s{s}_residual"]).max())
print("FINAL", name, dtype, {k: f"{v:.2e}" for k, v in E.items()})
