"""In-process A/B of a config's measurement (or, with AB_MODE=step in the environment, of its closed-loop integrator step) under
diagnostic options (aoenv_set_option 99), interleaved rounds:
    python scripts/ab_pyr.py [C3|C3M|C4|C5] opt0 opt1 ..."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlao_amd import _lib as L
from rlao_amd.env import BatchedAOEnv
name = sys.argv[1]
opts = [int(x) for x in sys.argv[2:]] or [0]
cfg = bench.CONFIGS[name]
env = BatchedAOEnv(n_envs=cfg["envs"], device=0, dtype="f32", return_frame=False)
env.set_params(dict(cfg["geo"], nLoop=64), wfs_type=cfg["wfs"], camera="ideal")
bench.start_episode(env)
res = {o: [] for o in opts}
for rnd in range(4):
    for o in opts:
        L.check(env._shard.lib.aoenv_set_option(env._shard.h, 99, o))
        step = os.environ.get("AB_MODE") == "step"
        env.measure()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        if step:
            env.run_integrator(0, 10)
        else:
            for _ in range(10):
                env.measure()
        t1.record()
        torch.cuda.synchronize()
        res[o].append(t0.elapsed_time(t1) / 10)
for o in opts:
    print(f"option {o:5d}: {'step' if os.environ.get('AB_MODE') == 'step' else 'measure()'} {statistics.median(res[o]):7.3f} ms  [{min(res[o]):.3f} .. {max(res[o]):.3f}]", flush=True)
