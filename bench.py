#!/usr/bin/env python3
"""Headline benchmark: AO env-steps/s of N batched closed loops (BASELINE.json configs[1]:
8 m telescope, 20x20 Shack-Hartmann, 256 envs per GPU, integrator closed loop, synthetic von Karman
turbulence), with the HBM roofline of the dominant kernel and the CPU oracle timed on the same host.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One process per GPU; envs are independent, so each rank steps its own shard (weak scaling: 256 envs per
GPU) and the only exchange is one all-gather of the per-env episode returns after the timed steps.
A "step" = one env.step() of every env of the job: turbulence update, DM, WFS, reconstruction, reward,
integrator command -- all on the device, inputs resident in HBM.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GEOMETRY = dict(diameter=8.0, nSubaperture=20, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0],
                windDirection=[72.0], fractionalR0=[1.0], altitude=[0.0], magnitude=8.0, opticalBand="I",
                mechanicalCoupling=0.35, nModes=50, gainCL=0.5, leak=0.99)
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_TRAFFIC = os.path.join(REPO, "profiles", "r01_g_pmc_traffic.json")   # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes


def measured_traffic(kernel, n_envs):
    """HBM bytes per launch of `kernel` from the committed PMC passes (same command, same n_envs), else None."""
    try:
        with open(PMC_TRAFFIC) as f:
            d = json.load(f)
        if d.get("n_envs") == n_envs:
            return d["kernels"][kernel]["hbm_bytes_per_launch"]
    except Exception:
        pass
    return None


def algorithmic_bytes(env):
    """float32 bytes that must cross HBM per env-step and per launch of each kernel."""
    R, S, L_, A, nsig, cam = env.R, env._atm_tables.S, env.param.nLayer, env.nValidAct, env.nSignal, env.cam_res
    step = L_ * S * S * 4 + R * R * 4 + R * R * 4 + cam * cam * 4 + (A + nsig + A) * 4
    per_kernel = {
        "env_step": step,                                    # the fused step kernel does all of it in one launch
        "phase": L_ * S * S * 4 + R * R * 4,                 # read the screens, write the residual phase
        "sh_spots": R * R * 4 + cam * cam * 4,               # read the phase, write the camera frame
        "sh_centroid": cam * cam * 4 + nsig * 4,             # read the frame, write the slopes
    }
    return step, per_kernel


def cpu_baseline(budget_s=20.0):
    """The NumPy oracle (port of the reference's CPU path) on one host core, same geometry, 1 env."""
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                                          # pragma: no cover
        threadpool_limits = None
    from oracle import ao_oracle as O                          # checker / baseline only, never the product path
    ctx = threadpool_limits(limits=1) if threadpool_limits else None
    try:
        env = O.OracleEnv(resolution=120, diameter=8.0, n_subap=20, r0=0.13, L0=30.0, windSpeed=[10.0],
                          windDirection=[72.0], fractionalR0=[1.0], altitude=[0.0], n_modes=50)
        env.new_episode(17)
        obs = env.reset_soft()
        for i in range(3):
            obs = env.step(i, 0.5 * obs)[0]
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < budget_s and n < 5000:
            obs = env.step(3 + n, np.float32(0.5 * obs))[0]
            n += 1
        dt = time.perf_counter() - t0
    finally:
        if ctx is not None:
            ctx.unregister() if hasattr(ctx, "unregister") else ctx.__exit__(None, None, None)
    return {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n} closed-loop steps of 1 env (8 m, 20x20 SH, R=120), NumPy float64 oracle, 1 BLAS thread"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=256)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--noise", nargs="?", const="razor", default="off", choices=["off", "photon", "razor"],
                    help="WFS camera: off = the ideal detector of the parity configuration; photon = photon (Poisson) noise only, "
                         "the reference envs' default (MAIN/OOPAOEnv/OOPAOEnv.py:379); razor = the Razor env's camera (photon + dark "
                         "+ read-out noise, QE, FWC, 10-bit ADC: MAIN/OOPAOEnv/OOPAOEnvRazor.py:243-250, 333)")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from rlao_amd import dist as aodist
    from rlao_amd.env import BatchedAOEnv

    rank, world = aodist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    n_local = args.envs_per_gpu
    n_total = n_local * world
    K, W = args.steps, args.warmup

    env = BatchedAOEnv(n_envs=n_local, device=local, dtype=args.dtype, return_frame=True,
                       env_index_offset=rank * n_local)
    env.set_params(dict(GEOMETRY, nLoop=3 * (K + W) + 48), wfs_type="shackhartmann")
    if args.noise == "razor":
        cam = env.wfs.cam
        cam.sensor, cam.FWC, cam.bits, cam.QE, cam.darkCurrent, cam.integrationTime = "CMOS", 10000, 10, 0.56, 5, 1 / 500
        cam.photonNoise, cam.readoutNoise = True, 14
    elif args.noise == "photon":
        env.wfs.cam.photonNoise = True
    env.generate_new_phase_screen(17)              # env e of the job uses seed 17 + e
    env.dm.coefs = 0
    env.measure()
    env.reset_soft()
    returns = torch.zeros(n_local, device=env.device, dtype=env.tdtype)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if world > 1:                                   # warm the communicator outside the timed region
        aodist.all_gather_returns(returns, n_total)
    env.run_integrator(0, W)
    env.accumulate_returns(returns)                 # every step adds its reward on the device (episode return)
    barrier()
    t0 = time.perf_counter()
    env.run_integrator(W, K)                        # K closed-loop steps of every env: one library call, K x (1..5) launches
    all_returns = aodist.all_gather_returns(returns, n_total)
    barrier()
    dt = time.perf_counter() - t0
    env.accumulate_returns(None)
    if world > 1:
        tmax = torch.tensor([dt], device=env.device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax[0])
    strehl = float(env._strehl.mean())

    # roofline leg: the same K steps again with a hipEvent pair around every kernel launch
    env._shard.profile(True)
    env.run_integrator(W + K, K)
    prof = env._shard.profile_read(env._stream())
    env._shard.profile(False)
    # the same loop with the reference envs' default camera setting (photon noise on, OOPAOEnv.py:379): reported beside the
    # headline, which runs the ideal detector like the parity tests and the CPU baseline
    photon = None
    if args.noise == "off" and world == 1:
        env.wfs.cam.photonNoise = True
        env.run_integrator(W + 2 * K, 10)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        env.run_integrator(W + 2 * K + 10, K)
        torch.cuda.synchronize()
        dtp = time.perf_counter() - t1
        env.wfs.cam.photonNoise = False
        photon = {"value": n_total * K / dtp, "unit": "env-steps/s", "ms_per_step": 1e3 * dtp / K,
                  "note": "same workload with photon (Poisson) noise on every camera pixel, Philox4x32-7 streams"}
    if rank != 0:
        return
    step_bytes, kbytes = algorithmic_bytes(env)
    per_kernel = {k: {"avg_us": 1e3 * ms / max(c, 1), "launches": c} for k, (ms, c) in prof.items() if c}
    dom = max((k for k in per_kernel if k in kbytes), key=lambda k: per_kernel[k]["avg_us"] * per_kernel[k]["launches"])
    dur_s = per_kernel[dom]["avg_us"] * 1e-6
    achieved = kbytes[dom] * n_local / dur_s / 1e9
    out = {
        "metric": "AO env-steps/sec (batched loops)", "value": n_total * K / dt, "unit": "env-steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "8m / 20x20 Shack-Hartmann, 256 batched envs per GPU, integrator closed loop "
                               "(BASELINE.json configs[1])",
                   "envs_per_gpu": n_local, "envs_total": n_total, "resolution": env.R, "n_valid_act": env.nValidAct,
                   "n_signal": env.nSignal, "layers": env.param.nLayer, "controller": "leaky integrator, gain 0.5",
                   "noise": {"razor": "Razor camera: photon + dark + read-out noise, QE 0.56, FWC 1e4, 10-bit ADC",
                             "photon": "photon (Poisson) noise, Philox4x32-7 streams", "off": "off"}[args.noise], "parallelism": f"env-shards x{world}, all-gather of episode returns"},
        "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(dom, n_local),
                     "algorithmic_bytes_per_launch": kbytes[dom] * n_local, "avg_launch_us": per_kernel[dom]["avg_us"]},
        "step_roofline": {"algorithmic_bytes_per_env_step": step_bytes,
                          "achieved_GBs": step_bytes * n_local / (dt / K) / 1e9,
                          "frac_of_hbm_peak": step_bytes * n_local / (dt / K) / 1e9 / HBM_PEAK_GBS},
        "kernels": per_kernel,
        "mean_strehl_last_step": strehl,
        "mean_episode_return": float(all_returns.mean()),
    }
    if photon is not None:
        out["with_photon_noise"] = photon
    if not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
