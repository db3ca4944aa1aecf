"""Stage timing inside the fused step kernel (diagnostic build with -DAO_STEP_STAMPS):
   AOENV_LIB=build/stamps/libaoenv_stamps.so python scripts/diag_stamps.py [n_envs]"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlao_amd.env import BatchedAOEnv
from rlao_amd import _lib as L
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
CAM = sys.argv[2] if len(sys.argv) > 2 else "papyrus"
env = BatchedAOEnv(n_envs=N, device=0, dtype="f32")
env.set_params(dict(bench.GEOMETRY, nLoop=600), wfs_type="shackhartmann", camera=CAM)
print("camera:", CAM)
env.generate_new_phase_screen(17); env.dm.coefs = 0; env.measure(); env.reset_soft()
env.run_integrator(0, 50); torch.cuda.synchronize()
lib = L.load()
st = np.zeros((min(N, 1024), 32), dtype=np.uint64)
assert lib.aoenv_debug_stamps(st.ctypes.data_as(C.c_void_p), st.shape[0]) == 0
st = st.astype(np.int64)
names = {0: "start", 1: "p0 sync", 3: "p0 s1 (mfma) + amp loads issued", 4: "p0 tile in LDS", 5: "p0 interp done",
         7: "p1 start(after mfma/epilogue p0)", 9: "p1 s1", 10: "p1 tile in LDS", 11: "p1 interp done",
         13: "p1 epilogue done", 14: "E0 sync", 22: "spots (DFT) done", 23: "camera: faint pixels drawn, queue written",
         2: "camera: queue drawn (wave 0)", 15: "camera finished, frame stored", 16: "max sync", 17: "centroid sync",
         19: "tail: t = M s", 20: "tail: o = M2C t, integrator", 21: "tail: obs write + reduce", 18: "tail: scalar finish"}
idx = [0, 24, 25, 26, 27, 1, 3, 4, 5, 7, 9, 10, 11, 13, 14, 22] + ([23, 6, 8] if CAM != "ideal" else []) + [15, 16, 17, 19, 20, 21, 18]
names.update({23: "camera: lenslet indices loaded", 6: "camera: own share of the tables landed", 8: "camera: barrier (tables complete)",
              15: "camera drawn, frame stored"})
names.update({24: "prologue: loads issued, LDS zeroed, 1st barrier", 25: "command image written, 2nd barrier", 26: "Gy C on the matrix cores, s1 stored",
              27: "ring scatter (crossing steps) + next Z"})
names[1] = "screen ranges + barrier before stage A"; names[3] = "p0 amp loads issued, sync"; names[9] = "p1 amp loads issued, sync"
prev = None
for i in idx:
    d = "" if prev is None else f"{np.median(st[:, i] - st[:, prev]):9.0f} ticks"
    print(f"{i:2d} {names[i]:36s} {d}")
    prev = i
tot = np.median(st[:, 18] - st[:, 0])
print("total ticks", tot, " min/max", (st[:, 18] - st[:, 0]).min(), (st[:, 18] - st[:, 0]).max())
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record(); env.run_integrator(50, 200); t1.record(); torch.cuda.synchronize()
print("us/step", 1e3 * t0.elapsed_time(t1) / 200)

ws = np.zeros((256, 16, 8), dtype=np.uint64)
if hasattr(lib, "aoenv_debug_wstamps") and lib.aoenv_debug_wstamps(ws.ctypes.data_as(C.c_void_p)) == 0:
    ws = ws.astype(np.int64)
    t0 = ws[:, :, 0].min(axis=1, keepdims=True)
    lab = ["wave start", "at s1 barrier", "after s1 barrier", "stage A done", "after E0 barrier", "spots done (at max barrier)", "end"]
    print("per-wave timeline (ticks since the workgroup's first wave started), median over envs; waves 0..15")
    for i, l in enumerate(lab):
        print(f"{l:30s}", " ".join(f"{int(v):6d}" for v in np.median(ws[:N, :, i] - t0[:N], axis=0)))
