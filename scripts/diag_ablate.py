import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlao_amd.env import BatchedAOEnv
from rlao_amd import _lib as L
import bench
N = int(sys.argv[1])
env = BatchedAOEnv(n_envs=N, device=0, dtype="f32")
env.set_params(dict(bench.GEOMETRY, nLoop=6000), wfs_type="shackhartmann")
env.generate_new_phase_screen(17); env.dm.coefs = 0; env.measure(); env.reset_soft()
env.run_integrator(0, 20); torch.cuda.synchronize()
i0 = 20
for ab in [0, 1, 2, 4, 8, 16, 32, 63]:
    L.check(env._shard.lib.aoenv_set_option(env._shard.h, 99, ab))
    env._shard.profile(True)
    env.run_integrator(i0, 60); i0 += 60
    prof = env._shard.profile_read(env._stream())
    env._shard.profile(False)
    print(f"N={N} ablate={ab:2d}: phase {1e3*prof['phase'][0]/prof['phase'][1]:.1f} us")
