import os, sys, time, json
import numpy as np, torch
sys.path.insert(0, "/root/repo")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rlao_amd.env import BatchedAOEnv
from scripts.run_config import CONFIGS
cfg = CONFIGS["C3"]; n = 256
t0 = time.perf_counter()
env = BatchedAOEnv(n_envs=n, device=0, dtype="f32", return_frame=False, env_seed_stride=0)
env.set_params(dict(cfg["geo"], nLoop=200), wfs_type=cfg["wfs"])
print("init", time.perf_counter() - t0, flush=True)
env.generate_new_phase_screen(17); env.dm.coefs = 0; env.measure(); env.reset_soft()
env.run_integrator(0, 10); torch.cuda.synchronize()
env._shard.profile(True); env.run_integrator(10, 30); prof = env._shard.profile_read(env._stream()); env._shard.profile(False)
print({k: (round(1e3 * ms / c, 1), c) for k, (ms, c) in prof.items() if c})
