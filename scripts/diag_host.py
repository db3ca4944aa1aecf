import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlao_amd.env import BatchedAOEnv
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
env = BatchedAOEnv(n_envs=N, device=0, dtype="f32")
env.set_params(dict(bench.GEOMETRY, nLoop=5000), wfs_type="shackhartmann")
env.generate_new_phase_screen(17); env.dm.coefs = 0; env.measure(); env.reset_soft()
env.run_integrator(0, 20); torch.cuda.synchronize()
for mode in ("single_call", "per_step"):
    t0 = time.perf_counter()
    if mode == "single_call":
        env.run_integrator(100, 200)
    else:
        for k in range(200):
            env.run_integrator(400 + k, 1)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={N} {mode}: host enqueue {1e6*(t1-t0)/200:.1f} us/step, total {1e6*(t2-t0)/200:.1f} us/step -> {N*200/(t2-t0):.0f} env-steps/s")
