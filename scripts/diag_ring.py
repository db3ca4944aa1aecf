import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlao_amd.env import BatchedAOEnv
from rlao_amd import _lib as L
g = np.load("tests/golden/tiny_sh.npz")
prm = dict(diameter=1.6, nSubaperture=4, nPixelPerSubap=6, nModes=8, nLoop=64)
env = BatchedAOEnv(n_envs=1, device=0, dtype="f64")
env.set_params(prm, camera="ideal", wfs_type="shackhartmann", m2c=g["m2c"])
env.generate_new_phase_screen(17)
at = env._atm_tables
zx = env._shard.download(L.B_XI, (1, at.n_inner + at.n_outer))[0]
scr = env._shard.download(0, (1, 1, at.S, at.S))[0, 0]
gm = g["s17_mapShift0"][0]
Z = gm.reshape(-1)[at.inner_idx]
xi = np.random.RandomState(17).normal(size=at.n_outer)
print("Z err", np.abs(zx[:at.n_inner] - Z).max(), "xi err", np.abs(zx[at.n_inner:] - xi).max())
X = at.AB @ zx
ring_dev = scr.reshape(-1)[at.outer_idx]
ring_ref = gm.reshape(-1)[at.outer_idx]
print("dev ring vs AB@zx(host)", np.abs(ring_dev - X).max(), " ref ring vs AB@zx", np.abs(ring_ref - X).max())
print("A@Z + B@xi vs ref", np.abs(at.A @ Z + at.B @ xi - ring_ref).max())
