#!/usr/bin/env python3
"""gpurun_out/<tag>_trace, <tag>_pmc_* (scripts/pmc_collect.sh, scripts/prof_configs.sh)  ->  profiles/<tag>_kernel_stats.csv +
profiles/<tag>_pmc.json.

A kernel is keyed by its FULL instantiation and its grid: `pyr_cols<float, 0>@270336` -- the float64 calibration launches of a
template (small grids, run once at set_params) and the float32 production launches (the shard's grid, every step) are different
entries, and a duration always belongs to the launches whose bytes it is divided into.  (Round 2 keyed by the bare name: 14
production + 95 calibration launches of k_pyr_cols were pooled, and a kernel came out at 12.8 TB/s.)

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: on gfx950 FETCH_SIZE counts a 128-byte read request as 64 bytes
for 16-byte-per-lane reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is exact for 16-byte stores.  Counter values are
averaged over the launches of an entry; durations come from the kernel trace of the same command (Start / End per dispatch).
`production` lists, per kernel, the entry with the largest total time: what bench.py's `roofline.traffic` reads.
    python scripts/pmc_to_json.py TAG N_ENVS CAMERA ["note"]
"""
import collections, csv, glob, json, os, shutil, sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, n_envs, camera = sys.argv[1], int(sys.argv[2]), sys.argv[3]
note = sys.argv[4] if len(sys.argv) > 4 else ""
OUT = os.path.join(REPO, "gpurun_out")
SHORT = {"k_env_step_sh6": "env_step", "k_ring_prepare": "ring_prepare", "k_gemm_nt_mfma": "gemm_mfma", "k_phase_mfma": "phase", "k_phase_mfma4": "phase",
         "k_sh_spots_p6": "sh_spots", "k_sh_centroid": "sh_centroid", "k_sh_tail": "sh_tail", "k_recon_finish": "recon_finish",
         "k_pyr_rows": "pyr_rows", "k_pyr_cols": "pyr_cols", "k_pyr_rows_inv": "pyr_rows_inv", "k_pyr_slopes": "pyr_slopes",
         "k_detector": "detector", "k_detector_sh6": "detector_sh6", "k_dm_rows": "dm_rows", "k_scatter_minmax": "ring_scatter",
         "k_ring_gemm_draw_ahead": "ring_gemm_draw_ahead", "k_ring_prepare_env": "ring_prepare_env",
         # the nRes = 528 float32 passes (pyr528_kernels.hip) are the same stages as the Stockham passes; their keys carry "/528"
         "k_pyr528_rows": "pyr_rows", "k_pyr528_cols": "pyr_cols", "k_pyr528_rows_inv": "pyr_rows_inv"}
HBM_PEAK = 8.0e12


def split_name(name):
    """'void ao::k_pyr_cols<float, 0>(ao::PyrArgs<float>)' -> ('pyr_cols', '<float, 0>')"""
    n = name.replace("void ", "").replace("ao::", "")
    depth, cut = 0, len(n)
    for i, ch in enumerate(n):                                    # the argument list starts at the first '(' outside <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    n = n[:cut]
    base, targs = (n.split("<", 1)[0], "<" + n.split("<", 1)[1]) if "<" in n else (n, "")
    return SHORT.get(base, base), targs + ("/528" if "528" in base else "")


def key_of(name, grid):
    base, targs = split_name(name)
    return f"{base}{targs}@{grid}", base


stats = glob.glob(os.path.join(OUT, f"{tag}_trace", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(REPO, "profiles", f"{tag}_kernel_stats.csv"))

# durations per (instantiation, grid) from the kernel trace
dur = collections.defaultdict(list)
base_of = {}
for f in glob.glob(os.path.join(OUT, f"{tag}_trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        k, b = key_of(r["Kernel_Name"], grid)
        base_of[k] = b
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))

agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(set))
for f in glob.glob(os.path.join(OUT, f"{tag}_pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k, b = key_of(r["Kernel_Name"], int(r["Grid_Size"]))
        base_of[k] = b
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])

kernels = {}
for k in sorted(set(agg) | set(dur)):
    b = base_of[k]
    if not (b.startswith("k_") or b in SHORT.values()):
        continue
    e = {"kernel": b}
    if dur.get(k):
        e["trace_launches"] = len(dur[k])
        e["avg_launch_us"] = sum(dur[k]) / len(dur[k]) / 1e3
        e["total_us"] = sum(dur[k]) / 1e3
    if k in agg:
        per = {c: x / max(len(cnt[k][c]), 1) for c, x in agg[k].items()}
        e["pmc_launches"] = max(len(s) for s in cnt[k].values())
        e["counters_per_launch"] = {c: round(x, 1) for c, x in sorted(per.items())}
        if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
            e["fetch_size_kb"], e["write_size_kb"] = per["FETCH_SIZE"], per["WRITE_SIZE"]
            e["hbm_bytes_per_launch"] = int((2 * per["FETCH_SIZE"] + per["WRITE_SIZE"]) * 1024)
            if e.get("avg_launch_us"):
                e["hbm_GBs"] = e["hbm_bytes_per_launch"] / e["avg_launch_us"] / 1e3
                e["hbm_frac_of_peak"] = e["hbm_bytes_per_launch"] / (e["avg_launch_us"] * 1e-6) / HBM_PEAK
        c = e["counters_per_launch"]
        # MFMA utilisation.  SQ_INSTS_VALU_MFMA_MOPS_F32 counts executed matrix work in units of 512 flops (checked on the step
        # kernel: 480 v_mfma_f32_16x16x4_f32 of 2048 flops per env); peak 157.3 TFLOP/s (f32-input MFMA); the pipe-busy counter
        # SQ_VALU_MFMA_BUSY_CYCLES / (duration x 2.4 GHz x 1024 SIMDs) is the same figure seen from the pipe.
        if c.get("SQ_INSTS_VALU_MFMA_MOPS_F32") and e.get("avg_launch_us"):
            flops, ns = c["SQ_INSTS_VALU_MFMA_MOPS_F32"] * 512, e["avg_launch_us"] * 1e3
            e["mfma"] = {"mfma_flops_per_launch": flops, "achieved_tflops": flops / ns / 1e3, "peak_tflops_f32": 157.3,
                         "frac": flops / ns / 1e3 / 157.3, "mfma_pipe_busy_frac": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (ns * 2.4 * 1024)}
    kernels[k] = e

# per kernel: the entry of the production launches -- the float32 instantiation where there is one (the calibration shards run in
# float64 and, at the ELT size or with a modulated Pyramid, for longer than the dozen profiled steps), the largest total time among those
production = {}
for b in sorted({e["kernel"] for e in kernels.values()}):
    cand = [k for k, e in kernels.items() if e["kernel"] == b and "total_us" in e]
    f32 = [k for k in cand if "double" not in k]
    cand = f32 or cand
    if cand:
        production[b] = max(cand, key=lambda k: kernels[k]["total_us"])
over = [k for k, e in kernels.items() if e.get("hbm_frac_of_peak", 0) > 1.0]
out = {"note": note or "rocprofv3 --pmc passes of `python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras` (scripts/pmc_collect.sh); "
       "entries keyed `kernel<template args>@grid`; counter values averaged per launch of the entry; durations from the kernel trace of "
       "the same command; hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE correction)",
       "n_envs": n_envs, "camera": camera, "kernels": kernels, "production": production, "entries_above_hbm_peak": over}
# (kept for bench.py: MFMA figure and duration of the production entry of every kernel, by bare kernel name)
out["mfma"] = {b: kernels[k]["mfma"] | {"avg_launch_us": kernels[k]["avg_launch_us"]} for b, k in production.items() if "mfma" in kernels[k]}
out["avg_launch_us"] = {b: kernels[k]["avg_launch_us"] for b, k in production.items()}
json.dump(out, open(os.path.join(REPO, "profiles", f"{tag}_pmc.json"), "w"), indent=1)
for b, k in sorted(production.items()):
    e = kernels[k]
    print(f"{k:60s} n={e.get('trace_launches', 0):5d} avg {e.get('avg_launch_us', 0):9.1f} us  hbm {e.get('hbm_bytes_per_launch', 0) / 1e6:9.2f} MB "
          f"= {e.get('hbm_GBs', 0):8.1f} GB/s")
if over:
    print("WARNING: entries above the HBM peak (bytes and duration of different launches?):", over)
