import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlao_amd.env import BatchedAOEnv
geo = dict(bench.CONFIGS["C3"]["geo"], nSubaperture=20, nModes=50, nLoop=64)
env = BatchedAOEnv(n_envs=1024, device=0, dtype="f32", return_frame=False)
env.set_params(geo, wfs_type="pyramid", camera="ideal")
print("R", env.R, "nRes", env._pyr_tables.nRes, "cam", env.cam_res)
bench.start_episode(env)
for _ in range(3): env.measure()
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20): env.measure()
t1.record(); torch.cuda.synchronize()
print("measure() ms", t0.elapsed_time(t1) / 20)
