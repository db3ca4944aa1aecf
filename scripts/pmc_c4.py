"""C4 (39 m, 80x80 Shack-Hartmann) at 64 envs, a few closed-loop steps: the workload for rocprofv3 counter passes."""
import os, sys
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from rlao_amd.env import BatchedAOEnv
from scripts.run_config import CONFIGS
cfg = CONFIGS["C4"]
env = BatchedAOEnv(n_envs=64, device=0, dtype="f32", return_frame=False, env_seed_stride=0)
env.set_params(dict(cfg["geo"], nLoop=50), wfs_type=cfg["wfs"])
env.generate_new_phase_screen(17); env.dm.coefs = 0; env.measure(); env.reset_soft()
env.run_integrator(0, 6); torch.cuda.synchronize()
print("done")
