"""Per-wave timeline of the Pyramid column pass and the workgroups resident per CU (diagnostic build with -DAO_PYR_STAMPS):
     cd rlao_amd/csrc && make && mkdir -p ../../build/pyrstamps &&
       hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -DAO_PYR_STAMPS -c pyr528_kernels.hip -o ../../build/pyrstamps/pyr528_kernels.o &&
       hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build/pyrstamps/libaoenv.so $(ls *.o | grep -v pyr528) ../../build/pyrstamps/pyr528_kernels.o
     AOENV_LIB=build/pyrstamps/libaoenv.so python scripts/diag_pyr_stamps.py
   The stamps perturb the kernel (s_memtime waits for the scalar unit): the column pass runs ~25 % slower in this build; the residency
   per CU and the order of the phases are what it is for."""
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
rt = np.zeros((32, 40, 3), dtype=np.uint64)
if hasattr(lib, "aoenv_debug_prt") and lib.aoenv_debug_prt(rt.ctypes.data_as(C.c_void_p)) == 0:
    rt = rt.astype(np.int64)[ok]
    d_rt = (rt[:, 1] - rt[:, 0]) / 100.0                          # us (100 MHz)
    d_tk = w[:, 0, 7] - w[:, 0, 0]
    print(f"wave 0 life: {np.median(d_rt):.2f} us = {np.median(d_tk):.0f} ticks -> {np.median(d_tk) / np.median(d_rt) / 1e3:.3f} ticks / ns")
    hw = rt[:, 2]
    xcc, hwid = (hw >> 32) & 0xF, hw & 0xFFFFFFFF
    cu, sh, se = (hwid >> 8) & 0xF, (hwid >> 12) & 1, (hwid >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    print("distinct CUs seen:", len(np.unique(key)), " XCCs:", np.unique(xcc), " SEs:", np.unique(se), " CU ids:", np.unique(cu))
    over = []
    for k in np.unique(key):
        iv = rt[key == k]
        ev = sorted([(a, 1) for a in iv[:, 0]] + [(b, -1) for b in iv[:, 1]])
        cur = best = 0
        for _, d in ev:
            cur += d
            best = max(best, cur)
        over.append(best)
    print("max workgroups at a time on one CU: histogram", np.bincount(over))
    span = (rt[:, 1].max() - rt[:, 0].min()) / 100.0
    print(f"the {ok.sum()} sampled workgroups ran within {span:.1f} us: {ok.sum() * np.median(d_rt) / span:.1f} of them at a time, "
          f"{ok.sum() * np.median(d_rt) / span / 256:.2f} per CU if spread over the chip")
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
