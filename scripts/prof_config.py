"""One BASELINE config's per-GPU shard for a few closed-loop steps: the workload of the rocprofv3 passes (kernel trace, PMC).
    rocprofv3 --kernel-trace --stats ... -- python3 scripts/prof_config.py C3 [n_envs] [steps]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlao_amd.env import BatchedAOEnv
name = sys.argv[1]
cfg = bench.CONFIGS[name]
n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["envs"]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
env = BatchedAOEnv(n_envs=n, device=0, dtype="f32", return_frame=False)
env.set_params(dict(cfg["geo"], nLoop=steps + 16), wfs_type=cfg["wfs"], second_dm=cfg.get("second_dm"), camera="papyrus")
if os.environ.get("AOENV_DEBUG_OPTION"):                        # diagnostic builds / paths (aoenv_set_option 99)
    from rlao_amd import _lib as L
    L.check(env._shard.lib.aoenv_set_option(env._shard.h, 99, int(os.environ["AOENV_DEBUG_OPTION"])))
bench.start_episode(env)
env.run_integrator(0, steps)
torch.cuda.synchronize()
print("done", name, n, steps)
