"""Throughput of the Pyramid configuration (reference Papyrus geometry: 1.52 m, 20x20, nRes 288), not the headline bench."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlao_amd.env import BatchedAOEnv
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mod = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
env = BatchedAOEnv(n_envs=N, device=0, dtype="f32", return_frame=False)
env.set_params(dict(diameter=1.52, nSubaperture=20, nPixelPerSubap=6, r0=0.25, L0=10.0, windSpeed=[20.0], windDirection=[72.0],
                    fractionalR0=[1.0], altitude=[0.0], nModes=50, nLoop=2000, modulation=mod), wfs_type="pyramid")
env.generate_new_phase_screen(17); env.dm.coefs = 0; env.measure(); env.reset_soft()
env.run_integrator(0, 10); torch.cuda.synchronize()
t0 = time.perf_counter(); K = 100
env.run_integrator(10, K); torch.cuda.synchronize()
dt = time.perf_counter() - t0
env._shard.profile(True); env.run_integrator(10 + K, 20); prof = env._shard.profile_read(env._stream()); env._shard.profile(False)
print(f"pyramid N={N} mod={mod}: {1e6*dt/K:.1f} us/step -> {N*K/dt:.0f} env-steps/s, strehl {float(env._strehl.mean()):.3f}",
      {k: round(1e3*ms/c, 1) for k, (ms, c) in prof.items() if c})
