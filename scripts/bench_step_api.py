"""Host overhead of the per-step Python API (what a trainer's rollout loop pays) against the on-device integrator loop."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlao_amd.env import BatchedAOEnv
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
env = BatchedAOEnv(n_envs=N, device=0, dtype="f32", return_frame=False)
env.set_params(dict(bench.GEOMETRY, nLoop=4000), wfs_type="shackhartmann")
env.generate_new_phase_screen(17); env.dm.coefs = 0; env.measure(); obs = env.reset_soft()
for i in range(20):
    obs, _, rew, sr, _, _ = env.step(i, 0.5 * obs)
torch.cuda.synchronize(); t0 = time.perf_counter(); K = 500
for i in range(20, 20 + K):
    obs, _, rew, sr, _, _ = env.step(i, 0.5 * obs)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"env.step loop (python, action = 0.5 * obs): {1e6 * dt / K:.1f} us/step -> {N * K / dt:.0f} env-steps/s")
t0 = time.perf_counter()
for i in range(600, 600 + K):
    obs, _, rew, sr, _, _ = env.step(i, obs)          # no torch op in the loop
dt2 = time.perf_counter() - t0; torch.cuda.synchronize()
print(f"env.step host time only: {1e6 * dt2 / K:.1f} us/call")
torch.cuda.synchronize(); t0 = time.perf_counter()
env.run_integrator(1200, K); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"run_integrator: {1e6 * dt / K:.1f} us/step -> {N * K / dt:.0f} env-steps/s")
