#!/usr/bin/env python3
"""gpurun_out/<tag>_trace, <tag>_pmc_* (scripts/pmc_collect.sh)  ->  profiles/<tag>_kernel_stats.csv + profiles/<tag>_pmc.json.

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts a 128-byte read request as 64 bytes
for 16-byte-per-lane reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte stores.  Counter values are
averaged over the launches of each kernel.
    python scripts/pmc_to_json.py TAG N_ENVS CAMERA ["note"]
"""
import collections, csv, glob, json, os, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, n_envs, camera = sys.argv[1], int(sys.argv[2]), sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else ""
OUT = os.path.join(REPO, "gpurun_out")
SHORT = {"k_env_step_sh6": "env_step", "k_ring_prepare": "ring_prepare", "k_gemm_nt_mfma": "gemm_ring", "k_phase_mfma": "phase",
         "k_sh_spots_p6": "sh_spots", "k_sh_centroid": "sh_centroid", "k_sh_tail": "sh_tail", "k_recon_finish": "recon_finish",
         "k_pyr_rows": "pyr_rows", "k_pyr_cols": "pyr_cols", "k_pyr_rows_inv": "pyr_rows_inv", "k_pyr_slopes": "pyr_slopes",
         "k_detector": "detector", "k_detector_sh6": "detector_sh6", "k_dm_rows": "dm_rows", "k_scatter_minmax": "ring_scatter",
         "k_ring_gemm_draw_ahead": "ring_gemm_draw_ahead", "k_ring_prepare_env": "ring_prepare_env"}


def short(name):
    n = name.replace("void ", "").replace("ao::", "").split("(")[0].split("<")[0]
    return SHORT.get(n, n)


stats = glob.glob(os.path.join(OUT, f"{tag}_trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(REPO, "profiles", f"{tag}_kernel_stats.csv"))
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(OUT, f"{tag}_pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
kernels = {}
for k, v in sorted(agg.items()):
    if not (k.startswith("k_") or k in SHORT.values()):
        continue
    per = {c: x / max(len(cnt[k][c]), 1) for c, x in v.items()}
    e = {"launches": max(len(s) for s in cnt[k].values()), "counters_per_launch": {c: round(x, 1) for c, x in sorted(per.items())}}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        e["fetch_size_kb"], e["write_size_kb"] = per["FETCH_SIZE"], per["WRITE_SIZE"]
        e["hbm_bytes_per_launch"] = int((2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024)
    kernels[k] = e
out = {"note": note or "rocprofv3 --pmc passes of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras` (scripts/pmc_collect.sh); "
       "counter values averaged per launch; hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction)",
       "n_envs": n_envs, "camera": camera, "kernels": kernels}
# MFMA utilisation of the kernels that use the matrix cores.  SQ_INSTS_VALU_MFMA_MOPS_F32 counts executed matrix work in units of
# 512 flops (checked on the step kernel: 480 v_mfma_f32_16x16x4_f32 of 2048 flops per env = 983 k flops x 256 envs / 491520 MOPS);
# achieved = that / the kernel's average duration in the kernel trace of the same command; peak 157.3 TFLOP/s (f32-input MFMA).
# SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs) is the same figure seen from the pipe.
dur = {}
if stats:
    for r in csv.DictReader(open(stats[0])):
        dur[short(r["Name"])] = float(r["AverageNs"])
mf = {}
for k, e in kernels.items():
    c = e["counters_per_launch"]
    if c.get("SQ_INSTS_VALU_MFMA_MOPS_F32") and k in dur:
        flops = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512
        mf[k] = {"mfma_flops_per_launch": flops, "avg_launch_us": dur[k] / 1e3, "achieved_tflops": flops / dur[k] / 1e3, "peak_tflops_f32": 157.3,
                 "frac": flops / dur[k] / 1e3 / 157.3,
                 "mfma_pipe_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (dur[k] * 2.4 * 1024)}
out["mfma"] = mf
out["avg_launch_us"] = {k: v / 1e3 for k, v in dur.items() if k in kernels}
json.dump(out, open(os.path.join(REPO, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps({k: {a: b for a, b in v.items() if a != "counters_per_launch"} for k, v in kernels.items()}, indent=1))
