"""fov != 0 diagnosis: device screens / OPD of the tiny 3-layer fov = 1 arcsec case against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ao_oracle as O
from rlao_amd.env import BatchedAOEnv
from rlao_amd import _lib as L
g = np.load("tests/golden/tiny_3layer_fov1.npz")
P = dict(diameter=1.6, nSubaperture=4, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0, 12.0, 11.0], windDirection=[0.0, 72.0, 144.0],
         fractionalR0=list(g["cfg_frac"]), altitude=[0.0, 1000.0, 5000.0], nModes=8, nLoop=64, fov=1.0)
env = BatchedAOEnv(n_envs=1, device=0, dtype="f64")
env.set_params(P, camera="ideal", wfs_type="shackhartmann", m2c=g["m2c"])
print("layer_res", env._atm_tables.layer_res, "uniform", env._atm_tables.uniform)
env.generate_new_phase_screen(17)
scr = env._download_screens()
for l, s in enumerate(scr):
    S = s.shape[-1]
    want = g["s17_mapShift0"][l][:S, :S]
    d = np.abs(s[0] - want)
    print("layer", l, "S", S, "max|screen - golden| interior", d[1:-1, 1:-1].max(), "ring", max(d[0].max(), d[-1].max(), d[:, 0].max(), d[:, -1].max()),
          "scale", np.abs(want).max())
env.dm.coefs = 0
env.measure()
opd = env._shard.download(L.B_OPD_ATM, (1, env.R, env.R))[0]
o = O.OracleEnv(resolution=24, diameter=1.6, n_subap=4, r0=0.13, L0=30.0, windSpeed=P["windSpeed"], windDirection=P["windDirection"],
                fractionalR0=P["fractionalR0"], altitude=P["altitude"], m2c=g["m2c"], n_modes=8, fov_arcsec=1.0)
o.new_episode(17)
print("max|opd_atm - oracle|", np.abs(opd - o.atm.OPD_no_pupil).max(), "scale", np.abs(o.atm.OPD_no_pupil).max())
print("obs0 err", np.abs(env.reset_soft()[0].cpu().numpy() - g["s17_obs0"]).max())
