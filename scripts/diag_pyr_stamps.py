"""Per-wave timeline of the Pyramid column pass (diagnostic build with -DAO_PYR_STAMPS):
     scripts/build_variant.sh pyrstamps pyr528_kernels -DAO_PYR_STAMPS   (or the commands in its header)
     AOENV_LIB=build/pyrstamps/libaoenv.so python scripts/diag_pyr_stamps.py"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from rlao_amd import _lib as L
from rlao_amd.env import BatchedAOEnv
cfg = bench.CONFIGS["C3"]
env = BatchedAOEnv(n_envs=cfg["envs"], device=0, dtype="f32", return_frame=False)
env.set_params(dict(cfg["geo"], nLoop=64), wfs_type=cfg["wfs"], camera="ideal")
bench.start_episode(env)
for _ in range(3):
    env.measure()
torch.cuda.synchronize()
lib = L.load()
st = np.zeros((32, 40, 6, 8), dtype=np.uint64)
assert lib.aoenv_debug_pstamps(st.ctypes.data_as(C.c_void_p)) == 0
st = st.astype(np.int64)
ok = st[:, :, 0, 7] > 0                                         # workgroups that ran (slot 4: one env in eight)
w = st[ok]                                                      # [n_wg, 6 waves, 8]
t0 = w[:, :, 0].min(axis=1)[:, None]
lab = ["wave start", "T1 + mask loaded, 24-pt, twiddle, exchange written", "after barrier 1", "22-pt, mask, 22-pt^-1, twiddle", "after barrier 2",
       "exchange written + barrier 3", "exchange read + 24-pt^-1", "stores issued (end)"]
print(f"{ok.sum()} workgroups; ticks since the workgroup's first wave started, median over workgroups, waves 0..5")
for i, l in enumerate(lab):
    print(f"{l:52s}", " ".join(f"{int(v):7d}" for v in np.median(w[:, :, i] - t0, axis=0)))
occ = (C.c_int * 3)()
if hasattr(lib, "aoenv_debug_pyr_occupancy") and lib.aoenv_debug_pyr_occupancy(occ) == 0:
    print("resident workgroups per CU (hipOccupancyMaxActiveBlocksPerMultiprocessor): rows", occ[0], "cols", occ[1], "rows_inv", occ[2])
span = w[:, :, 7].max() - w[:, :, 0].min()
print(f"sampled workgroups span {span} ticks: {span / ok.sum():.1f} ticks per workgroup chip-wide = {256 * span / ok.sum():.0f} per CU")
life = w[:, :, 7].max(axis=1) - w[:, :, 0].min(axis=1)
print("workgroup life (ticks): median", np.median(life), "min", life.min(), "max", life.max())
# how many workgroups overlap in time on the sampled range (a rough view of the CU's occupancy)
t0e, t1e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0e.record()
for _ in range(10):
    env.measure()
t1e.record()
torch.cuda.synchronize()
print("measure() ms", t0e.elapsed_time(t1e) / 10)
