"""Pyramid pass timing at the C3 geometry (8 m, 40x40, nRes 528): python scripts/time_pyr.py [n_envs]; per-stage event times."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlao_amd.env import BatchedAOEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
env = BatchedAOEnv(n_envs=N, device=0, dtype="f32", return_frame=False)
env.set_params(dict(bench.CONFIGS["C3"]["geo"], nLoop=200), wfs_type="pyramid", camera="ideal")
env.generate_new_phase_screen(17); env.dm.coefs = 0; env.measure(); env.reset_soft()
env.run_integrator(0, 10); torch.cuda.synchronize()
env._shard.profile(True); env.run_integrator(10, 20); prof = env._shard.profile_read(env._stream()); env._shard.profile(False)
print("envs", N, {k: round(1e3 * ms / c, 1) for k, (ms, c) in prof.items() if c})
