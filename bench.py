#!/usr/bin/env python3
"""Benchmark of the batched AO environment: AO env-steps/s of N closed loops per GPU.

    python bench.py --gpus N --steps K --warmup W          (N > 1: the ranks are spawned here as child processes)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Headline (`value`): BASELINE.json configs[1] -- 8 m telescope, 20x20 Shack-Hartmann, 256 envs per GPU, leaky-integrator closed
loop, synthetic von Karman turbulence, the reference env's default camera (photon noise on every WFS pixel,
MAIN/OOPAOEnv/OOPAOEnv.py:379).  One process per GPU; envs are independent, so every rank steps its own shard (weak scaling) and
the only exchange is one all-gather of the per-env episode returns inside each timed region.  A "step" = one env.step() of
every env of the job: turbulence update, DM, WFS camera, slopes, reconstruction, reward, integrator command -- all on the
device, inputs resident in HBM.

Timing: W warm-up steps, then R timed regions of EXACTLY K steps each, every region bracketed by barrier +
torch.cuda.synchronize() on both sides and reduced with MAX over the ranks; `ms_per_step` / `value` come from the MEDIAN region
(R is chosen so that the regions add up to >= --min-seconds; their spread is reported).  Beside it, in the same JSON line:
`roofline` (dominant kernel, HIP-event timed on the launch stream), `cpu_baseline` (the NumPy oracle on this host: one core,
and one process per core), `ideal_detector` / `razor_camera` (the same workload with other cameras), `step_api` (the per-call
Python API a trainer uses) and `configs` (BASELINE configs[2..4] per-GPU shards, driver-timed; N = 1 only).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

GEOMETRY = dict(diameter=8.0, nSubaperture=20, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0],
                windDirection=[72.0], fractionalR0=[1.0], altitude=[0.0], magnitude=8.0, opticalBand="I",
                mechanicalCoupling=0.35, nModes=50, gainCL=0.5, leak=0.99)
# BASELINE.json configs[2..4], one GPU's shard each
CONFIGS = {
    "C3": dict(label="8m / 40x40 Pyramid WFS, 1024 batched envs, PO4AO policy rollout (BASELINE.json configs[2])",
               wfs="pyramid", envs=1024, controller="policy",
               geo=dict(diameter=8.0, nSubaperture=40, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
                        fractionalR0=[1.0], altitude=[0.0], nModes=50, modulation=0.0)),
    "C3M": dict(label="8m / 40x40 Pyramid WFS with modulation 3 lambda/D (nTheta = 20), 256 batched envs, integrator closed loop "
                      "(SURVEY.md 8a A6 / 8d: the x20 FFT term)",
                wfs="pyramid", envs=256, controller="integrator",
                geo=dict(diameter=8.0, nSubaperture=40, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
                         fractionalR0=[1.0], altitude=[0.0], nModes=50, modulation=3.0)),
    "C4": dict(label="ELT-scale 39m, 80x80 Shack-Hartmann, 4096 envs over 8 GPUs = 512 per GPU (BASELINE.json configs[3])",
               wfs="shackhartmann", envs=512, controller="integrator",
               geo=dict(diameter=39.0, nSubaperture=80, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0], windDirection=[72.0],
                        fractionalR0=[1.0], altitude=[0.0], nModes=300)),
    "C5": dict(label="3-layer atmosphere + dual-DM MCAO, 2048 envs over 8 GPUs = 256 per GPU (BASELINE.json configs[4])",
               wfs="shackhartmann", envs=256, controller="integrator", second_dm=dict(nSubaperture=10),
               geo=dict(diameter=8.0, nSubaperture=20, nPixelPerSubap=6, r0=0.13, L0=30.0, windSpeed=[10.0, 12.0, 11.0],
                        windDirection=[0.0, 72.0, 144.0], fractionalR0=[0.45 / 0.65, 0.1 / 0.65, 0.1 / 0.65],
                        altitude=[0.0, 1000.0, 5000.0], nModes=50)),
}
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = the f32 vector rate
# rocprofv3 --pmc passes of this command (scripts/pmc_collect.sh -> scripts/pmc_to_json.py), one file per camera setting, and of the
# BASELINE configs' shards (scripts/prof_configs.sh): `roofline.traffic` is read from these COMMITTED passes (a benchmark run under
# the driver cannot collect counters itself), never measured in the run that prints the line
PMC_FILES = {"papyrus": os.path.join(REPO, "profiles", "r03_c2_pmc.json"), "ideal": os.path.join(REPO, "profiles", "r03_c2_ideal_pmc.json"),
             **{c: os.path.join(REPO, "profiles", f"r03_{c}_pmc.json") for c in ("C3", "C3M", "C4", "C5")}}
# the kernels behind every profiled stage of aoenv_profile (names as scripts/pmc_to_json.py shortens them)
STAGE_KERNELS = {"env_step": ["env_step"], "phase": ["phase", "dm_rows"], "sh_spots": ["sh_spots"], "detector": ["detector_sh6", "detector"],
                 "pyramid": ["pyr_rows", "pyr_cols", "pyr_rows_inv", "pyr_slopes"], "sh_centroid": ["sh_centroid"], "sh_tail": ["sh_tail"],
                 "gemm_ring": ["ring_gemm_draw_ahead", "gemm_mfma"], "recon_finish": ["recon_finish"]}
CAMERA_NOTE = {"papyrus": "photon (Poisson) noise on every WFS pixel: the reference env's default camera (OOPAOEnv.py:379)",
               "razor": "Razor camera: photon + dark + read-out noise, QE 0.56, FWC 1e4, 10-bit ADC (OOPAOEnvRazor.py:243-250, 332-333)",
               "ideal": "ideal detector (the parity configuration)"}


# ------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without a torchrun environment spawns the N ranks as child processes
# ------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launcher_command(argv, n_gpus: int, port=None):
    """The torch.distributed.run command line that starts one rank per GPU of this node."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + list(argv)


def maybe_spawn_ranks(args, argv):
    """N > 1 and no rank environment: nothing has touched the GPU yet, so the ranks are started as CHILD processes (never an exec
    from a process that initialised the GPU) and this process exits with their return code."""
    if args.gpus <= 1 or "RANK" in os.environ:
        return
    cmd = launcher_command([a for a in argv if a != "--dry-run-launch"], args.gpus)
    if args.dry_run_launch:
        print(json.dumps({"launch": cmd}))
        raise SystemExit(0)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    raise SystemExit(subprocess.run(cmd, env=env).returncode)


# ------------------------------------------------------------------------------------------------------------------
# CPU baseline: the NumPy oracle (port of the reference's CPU path), timed on this host before anything touches the GPU
# ------------------------------------------------------------------------------------------------------------------
def _oracle_env(camera: str):
    from oracle import ao_oracle as O                          # checker / baseline only, never the product path
    env = O.OracleEnv(resolution=120, diameter=8.0, n_subap=20, r0=0.13, L0=30.0, windSpeed=[10.0],
                      windDirection=[72.0], fractionalR0=[1.0], altitude=[0.0], n_modes=50)
    if camera == "papyrus":
        env.wfs.cam = O.Detector(photonNoise=True, seed=os.getpid())
    elif camera == "razor":
        env.wfs.cam = O.Detector(photonNoise=True, readoutNoise=14, QE=0.56, darkCurrent=5, integrationTime=1 / 500, FWC=10000,
                                 bits=10, sensor="CMOS", seed=os.getpid())
    env.new_episode(17)
    return env


def _oracle_steps(env, budget_s, max_steps=100000):
    obs = env.reset_soft()
    for i in range(3):
        obs = env.step(i, 0.5 * obs)[0]
    t0, n = time.perf_counter(), 0
    while time.perf_counter() - t0 < budget_s and n < max_steps:
        obs = env.step((3 + n) % 9000, np.float32(0.5 * obs))[0]
        n += 1
    return n, time.perf_counter() - t0


def _limit_threads():
    try:
        from threadpoolctl import threadpool_limits
        return threadpool_limits(limits=1)
    except Exception:                                          # pragma: no cover
        return None


def available_cores() -> int:
    """Cores this process may really use: the affinity mask, cut down to the cgroup CPU quota when there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def _cpu_worker(conn, camera, budget_s):
    _limit_threads()
    env = _oracle_env(camera)
    conn.send("ready")
    conn.recv()                                                # all workers start together
    conn.send(_oracle_steps(env, budget_s))


def cpu_baseline(camera: str, budget_s: float, all_cores: bool):
    """One env on one core (the reference is a single-process simulator), then P = one process per available core, each with its
    own env (envs are independent): aggregate env-steps/s of this host.  Same geometry and camera as the GPU headline."""
    ctx = _limit_threads()
    env = _oracle_env(camera)
    n, dt = _oracle_steps(env, budget_s)
    if ctx is not None and hasattr(ctx, "unregister"):
        ctx.unregister()
    out = {"value": n / dt, "unit": "env-steps/s", "cores": 1, "kind": "port",
           "sample": f"{n} closed-loop steps of 1 env (8 m, 20x20 SH, R=120, camera: {camera}), NumPy float64 oracle, 1 BLAS thread, "
                     f"{dt:.1f} s"}
    if all_cores:
        import multiprocessing as mp
        P = available_cores()
        mpc = mp.get_context("fork")                           # forked before torch / HIP are imported in this process
        procs = []
        for _ in range(P):
            a, b = mpc.Pipe()
            p = mpc.Process(target=_cpu_worker, args=(b, camera, budget_s), daemon=True)
            p.start()
            procs.append((p, a))
        for _, a in procs:
            a.recv()
        t0 = time.perf_counter()
        for _, a in procs:
            a.send("go")
        res = [a.recv() for _, a in procs]
        wall = time.perf_counter() - t0
        for p, _ in procs:
            p.join(timeout=10)
        steps = sum(r[0] for r in res)
        out["all_cores"] = {"value": steps / max(r[1] for r in res), "unit": "env-steps/s", "cores": P, "kind": "port",
                            "sample": f"{P} processes x 1 env x 1 BLAS thread, {steps} steps in {wall:.1f} s wall "
                                      f"(envs are independent; the reference itself is single-process)"}
    return out


# ------------------------------------------------------------------------------------------------------------------
# byte / flop models (float32 state) -- stated in DESIGN.md section 4
# ------------------------------------------------------------------------------------------------------------------
def algorithmic_bytes(env):
    """bytes that must cross HBM per env-step, total and per profiled kernel stage."""
    R, S, L_, A, nsig, cam = env.R, env._atm_tables.S, env.param.nLayer, env.nValidAct, env.nSignal, env.cam_res
    step = L_ * S * S * 4 + R * R * 4 + R * R * 4 + cam * cam * 4 + (A + nsig + A) * 4
    per = {"env_step": step, "phase": L_ * S * S * 4 + R * R * 4, "sh_spots": R * R * 4 + cam * cam * 4,
           "sh_centroid": cam * cam * 4 + nsig * 4, "sh_tail": cam * cam * 4 + (nsig + 2 * A) * 4,
           "detector": 2 * cam * cam * 4}                          # the stand-alone camera: frame in, frame out
    if env.wfs_type == "pyr":
        N, nt = env._pyr_tables.nRes, env._wfs_n_theta
        # as implemented: rows pass R x N complex out, column pass reads it and writes N x N, inverse row pass reads that
        per["pyramid"] = R * R * 4 + nt * (2 * R * N * 8 + 2 * N * N * 8) + cam * cam * 4
        step = L_ * S * S * 4 + R * R * 4 + per["pyramid"] + (A + nsig + A) * 4
        per["survey_formula"] = L_ * S * S * 4 + 2 * R * R * 4 + cam * cam * 4 + nt * 2 * 2 * 2 * N * N * 8
    return step, per


def mfma_flops_per_env_step(env):
    """float32 matrix-core work per env-step: DM surface (separable: Gy C then (Gy C) Gx^T), reconstruction t = M s, o = M2C t,
    and the ring extrusion [A | B] [Z; xi] weighted by how often a layer crosses a pixel."""
    R, nA, A, nsig, K = env.R, env.nActuator, env.nValidAct, env.nSignal, env.param.nModes
    dm = 2 * R * nA * nA + 2 * R * R * nA
    rec = 2 * K * (nsig + A)
    at = env._atm_tables
    ratio = np.abs(at.wind_ratio(env.param.windSpeed, env.param.windDirection, env.param.samplingTime))
    crossings = float(np.minimum(ratio.sum(axis=1), 2.0).sum())
    ring = crossings * 2 * at.n_outer * (at.n_inner + at.n_outer)
    return {"dm": dm, "recon": rec, "ring": ring, "total": dm + rec + ring}


def load_pmc(n_envs, which, camera="papyrus"):
    """The committed counter summary of this workload (`which`: a camera name for the headline, else a config name), or None."""
    try:
        with open(PMC_FILES[which]) as f:
            d = json.load(f)
        if d.get("n_envs") == n_envs and d.get("camera") == camera and "production" in d:
            d["file"] = os.path.relpath(PMC_FILES[which], REPO)
            return d
    except Exception:
        pass
    return None


def pmc_traffic(pmc, stage, launches_per_invocation=1):
    """HBM bytes one invocation of `stage` moves, by the counters: sum over the stage's kernels (their production entries)."""
    if pmc is None:
        return None, {}
    parts = {}
    for b in STAGE_KERNELS.get(stage, [stage]):
        k = pmc["production"].get(b)
        if k and "hbm_bytes_per_launch" in pmc["kernels"][k]:
            parts[k] = pmc["kernels"][k]["hbm_bytes_per_launch"] * launches_per_invocation
    return (sum(parts.values()) if parts else None), parts


def attach_traffic(roof, pmc, launches_per_invocation=1):
    if roof is None:
        return
    t, parts = pmc_traffic(pmc, roof["kernel"], launches_per_invocation)
    roof["traffic"] = t
    if t is not None:
        roof["traffic_source"] = f"committed rocprofv3 --pmc passes of this workload ({pmc['file']}: FETCH_SIZE x 2 + WRITE_SIZE), not measured in this run"
        roof["traffic_vs_algorithmic"] = t / roof["algorithmic_bytes_per_launch"]
        roof["traffic_by_kernel"] = parts


# ------------------------------------------------------------------------------------------------------------------
def stats(ts):
    a = np.sort(np.asarray(ts))
    return {"median": float(np.median(a)), "min": float(a[0]), "max": float(a[-1]),
            "p10": float(a[int(0.1 * (len(a) - 1))]), "p90": float(a[int(round(0.9 * (len(a) - 1)))])}


class Timer:
    """R regions of exactly K steps, each bracketed by barrier + synchronize and reduced with MAX over the ranks."""

    def __init__(self, torch, dist, world, device):
        self.torch, self.dist, self.world, self.device = torch, dist, world, device

    def barrier(self):
        if self.world > 1:
            self.dist.barrier()
        if getattr(self.device, "type", "cuda") == "cuda":
            self.torch.cuda.synchronize()

    def regions(self, body, min_seconds, min_repeats=5, max_repeats=2000):
        """Region times (MAX over the ranks).  self.rank_spread: per region, (fastest, slowest) rank's own time."""
        times = []
        self.rank_spread = []
        while True:
            self.barrier()
            t0 = time.perf_counter()
            body(len(times))
            self.barrier()
            dt = time.perf_counter() - t0
            if self.world > 1:
                t = self.torch.tensor([dt], device=self.device, dtype=self.torch.float64)
                every = [self.torch.empty_like(t) for _ in range(self.world)]
                self.dist.all_gather(every, t)
                own = [float(x[0]) for x in every]
                self.rank_spread.append((min(own), max(own)))
                dt = max(own)
            times.append(dt)
            if len(times) >= max_repeats or (len(times) >= min_repeats and sum(times) >= min_seconds):
                return times


def set_camera(env, camera):
    cam = env.wfs.cam
    if camera == "razor":
        cam.configure(sensor="CMOS", FWC=10000, bits=10, QE=0.56, darkCurrent=5, integrationTime=1 / 500, photonNoise=True, readoutNoise=14)
    else:
        cam.configure(sensor="CCD", FWC=None, bits=None, QE=1, darkCurrent=0, readoutNoise=0, photonNoise=camera == "papyrus")


def start_episode(env, seed=17):
    env.generate_new_phase_screen(seed)            # env e of the job uses seed + e
    env.dm.coefs = 0
    env.measure()
    return env.reset_soft()


def kernel_profile(env, run, n_local):
    """HIP-event pairs around every kernel launch of `run()` (aoenv_profile): {stage: {avg_us, launches}} and the dominant stage's
    roofline.  The events sit on the launch stream, so a stage's time is its kernels' device time plus their launch gaps."""
    env._shard.profile(True)
    run()
    prof = env._shard.profile_read(env._stream())
    env._shard.profile(False)
    step_bytes, kbytes = algorithmic_bytes(env)
    per_kernel = {k: {"avg_us": 1e3 * ms / max(c, 1), "launches": c} for k, (ms, c) in prof.items() if c}
    known = [k for k in per_kernel if k in kbytes]
    if not known:
        return per_kernel, None, step_bytes
    dom = max(known, key=lambda k: per_kernel[k]["avg_us"] * per_kernel[k]["launches"])
    achieved = kbytes[dom] * n_local / (per_kernel[dom]["avg_us"] * 1e-6) / 1e9
    roof = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": None, "algorithmic_bytes_per_launch": kbytes[dom] * n_local, "avg_launch_us": per_kernel[dom]["avg_us"]}
    return per_kernel, roof, step_bytes


class ConvPolicy:
    """PO4AO policy shape (MAIN/PO4AO/conv_models_simple.py:56-111): 3 x Conv2d(3x3, 64 filters), LeakyReLU, clamp to [-1, 1],
    projection F on the controlled modes; input = observation + n_history-1 past observations + n_history-1 past actions; fixed
    random weights (seed 5, MAIN/PO4AO/mbrl.py:18-20), eval mode.  The caller's network: stock PyTorch-ROCm, not part of the env."""

    def __init__(self, env, n_history=20, n_filt=64):
        import torch
        import torch.nn as nn
        torch.manual_seed(5)
        self.torch = torch
        self.xv = torch.as_tensor(env.xvalid, device=env.device)
        self.yv = torch.as_tensor(env.yvalid, device=env.device)
        self.F = torch.as_tensor(env.F, device=env.device, dtype=torch.float32)
        self.net = nn.Sequential(nn.Conv2d(2 * n_history - 1, n_filt, 3, padding=1), nn.LeakyReLU(),
                                 nn.Conv2d(n_filt, n_filt, 3, padding=1), nn.LeakyReLU(), nn.Conv2d(n_filt, 1, 3, padding=1))
        for m in self.net.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.normal_(m.weight, mean=0, std=0.1)
                nn.init.constant_(m.bias, 0)
        self.net = self.net.to(env.device).eval()

    def __call__(self, obs, past_obs, past_act):
        """obs [n, A, A]; past_obs / past_act [n, n_history - 1, A, A] windows (views of the device ring buffers), newest first"""
        torch = self.torch
        with torch.no_grad():
            out = self.net(torch.cat([obs.unsqueeze(1), past_obs, past_act], dim=1)).clamp(-1, 1).squeeze(1)
            vec = out[:, self.xv, self.yv] @ self.F.T
            ret = torch.zeros_like(out)
            ret[:, self.xv, self.yv] = vec
            return ret


def bench_config(name, args, torch, timer, rank=0, world=1):
    """One BASELINE config's per-GPU shard: build, warm up, R regions of K steps, per-kernel profile.  world > 1 (`--config NAME
    --gpus N`): every rank steps its own shard of cfg["envs"] envs (global env index = rank * envs + e: atmosphere seeds and camera
    noise streams), and every timed region ends with the all-gather of the per-env episode returns."""
    from rlao_amd import dist as aodist
    from rlao_amd.env import BatchedAOEnv
    from rlao_amd.wrappers import DeviceHistory
    cfg = CONFIGS[name]
    K = min(args.steps, 20)
    n = cfg["envs"] if args.envs_per_gpu is None else args.envs_per_gpu
    n_total = n * world
    t0 = time.perf_counter()
    env = BatchedAOEnv(n_envs=n, device=timer.device.index, dtype="f32", return_frame=False, env_index_offset=rank * n)
    env.set_params(dict(cfg["geo"], nLoop=12 * K + 64), wfs_type=cfg["wfs"], second_dm=cfg.get("second_dm"), camera="papyrus")
    init_s = time.perf_counter() - t0
    obs = start_episode(env)
    returns = torch.zeros(n, device=env.device, dtype=env.tdtype)
    if world > 1:
        aodist.all_gather_returns(returns, n_total)             # warm the communicator outside the timed regions
    gathered = {}
    out = {"workload": cfg["label"], "envs_per_gpu": n, "envs_total": n_total, "steps": K, "resolution": env.R, "n_valid_act": env.nValidAct,
           "n_signal": env.nSignal, "layers": env.param.nLayer, "camera": CAMERA_NOTE["papyrus"], "init_s": round(init_s, 1)}
    if cfg["controller"] == "policy":
        n_history, A = 20, env.nActuator
        policy = ConvPolicy(env, n_history)
        # observation / action histories as mirrored device ring buffers (rlao_amd.wrappers.DeviceHistory): a push is two slot
        # writes, the policy's input window a view -- no history is rolled or re-concatenated per step
        hist_o = DeviceHistory(n, n_history - 1, A, env.device)
        hist_a = DeviceHistory(n, n_history - 1, A, env.device)
        st = {"obs": obs}

        def rollout(i0, k):                                     # MAIN/PO4AO/mbrl.py:64-89 with a leading env dimension
            for t in range(i0, i0 + k):
                action = policy(st["obs"], hist_o.window(), hist_a.window())
                nxt, _, reward, strehl, _, _ = env.step(t, action)
                hist_o.push(st["obs"])
                hist_a.push(action)
                returns.add_(reward)
                st["obs"] = nxt
            return strehl

        def region(r):
            rollout(8 + (r % 8) * K, K)
            gathered["returns"] = aodist.all_gather_returns(returns, n_total)

        rollout(0, 3)
        times = timer.regions(region, args.min_seconds, min_repeats=3, max_repeats=200)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            policy(st["obs"], hist_o.window(), hist_a.window())
        torch.cuda.synchronize()
        out["controller"] = "ConvPolicy 3 x Conv2d(64), n_history 20, random weights seed 5, evaluated for all envs on the GPU"
        out["policy_ms_per_step"] = 1e3 * (time.perf_counter() - t1) / 5
        prof_run = lambda: rollout(8 + 9 * K, K)               # noqa: E731
        out["mean_strehl"] = float(env._strehl.mean())
        out["mean_strehl_note"] = "random-weight policy: carries no information about the loop; see integrator_mean_strehl"
        # the same shard under the integrator (gain 0.5), 2 K steps from a fresh episode: a Strehl that says the loop closes
        env.dm_prev = 0                                         # (the prologue alone keeps the integrator state: here the random policy's)
        obs = start_episode(env)
        env.run_integrator(0, 2 * K)
        out["integrator_mean_strehl"] = float(env._strehl.mean())
        out["integrator_residual_over_total_rms"] = float(env.residual[2 * K - 1].mean() / env.total[2 * K - 1].mean())
    else:
        env.run_integrator(0, 4)
        env.accumulate_returns(returns)

        def region(r):
            env.run_integrator(8 + (r % 8) * K, K)
            gathered["returns"] = aodist.all_gather_returns(returns, n_total)

        times = timer.regions(region, args.min_seconds, min_repeats=3, max_repeats=500)
        env.accumulate_returns(None)
        out["controller"] = "leaky integrator, gain 0.5, on the device"
        last = 8 + ((len(times) - 1) % 8) * K + K - 1
        out["residual_over_total_rms"] = float(env.residual[last].mean() / env.total[last].mean())
        prof_run = lambda: env.run_integrator(8 + 9 * K, K)     # noqa: E731
        out["mean_strehl"] = float(env._strehl.mean())
    s = stats(times)
    out.update(value=n_total * K / s["median"], unit="env-steps/s", ms_per_step=1e3 * s["median"] / K, repeats=len(times),
               region_ms={k: 1e3 * v for k, v in s.items()}, mean_episode_return=float(gathered["returns"].mean()))
    if world > 1:
        out["region_ms_per_rank"] = {"fastest_rank_min": 1e3 * min(a for a, _ in timer.rank_spread),
                                     "slowest_rank_max": 1e3 * max(b for _, b in timer.rank_spread)}
    per_kernel, roof, step_bytes = kernel_profile(env, prof_run, n)
    nt = getattr(env, "_wfs_n_theta", 1)
    attach_traffic(roof, load_pmc(n, name), launches_per_invocation=-(-nt // 4) if roof and roof["kernel"] == "pyramid" else 1)
    out["kernels"] = per_kernel
    out["roofline"] = roof
    out["step_roofline"] = {"algorithmic_bytes_per_env_step": step_bytes, "achieved_GBs": step_bytes * n / (s["median"] / K) / 1e9,
                            "frac_of_hbm_peak": step_bytes * n / (s["median"] / K) / 1e9 / HBM_PEAK_GBS}
    fl = mfma_flops_per_env_step(env)
    out["mfma"] = {"flops_per_env_step": fl["total"], "achieved_tflops": fl["total"] * n / (s["median"] / K) / 1e12,
                   "peak_tflops_f32": MFMA_F32_PEAK_TFLOPS}
    env.close()
    del env
    torch.cuda.empty_cache()
    return out


def config_headline(args, torch, dist, rank, world, local, cpu):
    """`--config C3|C3M|C4|C5 [--gpus N]`: that BASELINE config as the one JSON line of the contract (rank 0 prints it)."""
    timer = Timer(torch, dist, world, torch.device("cuda", local))
    name = args.config
    r = bench_config(name, args, torch, timer, rank=rank, world=world)
    roof = r.get("roofline")
    line = {"metric": "AO env-steps/sec (batched loops)", "value": r["value"], "unit": "env-steps/s", "n_gpus": world,
            "steps": r["steps"], "warmup": 4, "ms_per_step": r["ms_per_step"], "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "repeats": r["repeats"], "region_ms": r["region_ms"],
            "config": {"workload": r["workload"], "name": name, "envs_per_gpu": r["envs_per_gpu"], "envs_total": r["envs_total"],
                       "resolution": r["resolution"], "n_valid_act": r["n_valid_act"], "n_signal": r["n_signal"], "layers": r["layers"],
                       "controller": r["controller"], "camera": r["camera"],
                       "parallelism": f"env-shards x{world}, all-gather of episode returns"},
            "roofline": roof, "step_roofline": r["step_roofline"], "mfma": r["mfma"], "kernels": r["kernels"],
            "mean_strehl_last_step": r["mean_strehl"], "mean_episode_return": r["mean_episode_return"]}
    for k in ("region_ms_per_rank", "integrator_mean_strehl", "integrator_residual_over_total_rms", "residual_over_total_rms", "policy_ms_per_step"):
        if k in r:
            line[k] = r[k]
    if cpu is not None:
        line["cpu_baseline"] = cpu
    return line


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=None, help="default: 256 (C2) / the config's own per-GPU shard")
    ap.add_argument("--config", default="C2", type=str.upper, choices=["C2"] + sorted(CONFIGS),
                    help="the workload of the headline line: C2 (default) = BASELINE configs[1]; C3 / C4 / C5 = BASELINE configs[2..4] "
                         "as stated (with --gpus 8: 512 resp. 256 envs per GPU), C3M = the modulated Pyramid.  Same timer: K-step "
                         "regions, all-gather of the episode returns inside each")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--min-seconds", type=float, default=0.6, help="the timed regions of K steps are repeated until they add up to this")
    ap.add_argument("--noise", nargs="?", const="razor", default="photon", choices=["off", "photon", "razor"],
                    help="WFS camera of the headline: photon (default) = photon (Poisson) noise only, the reference envs' default "
                         "(MAIN/OOPAOEnv/OOPAOEnv.py:379); razor = the Razor env's camera (photon + dark + read-out noise, QE, FWC, "
                         "10-bit ADC: OOPAOEnvRazor.py:243-250, 333); off = the ideal detector of the parity configuration")
    ap.add_argument("--configs", default="C3,C3M,C4,C5", help="comma list of BASELINE configs timed beside the headline (N = 1 only); 'none' skips them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="headline + roofline only (profiling runs)")
    ap.add_argument("--cpu-seconds", type=float, default=8.0)
    ap.add_argument("--dry-run-launch", action="store_true", help="print the rank launcher command of --gpus N and exit")
    args = ap.parse_args()
    maybe_spawn_ranks(args, sys.argv[1:])
    camera = {"photon": "papyrus", "razor": "razor", "off": "ideal"}[args.noise]

    rank_env, world_env = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    cpu = None
    if not args.no_cpu_baseline and rank_env == 0 and world_env == 1 and args.config == "C2":
        cpu = cpu_baseline(camera, args.cpu_seconds, all_cores=not args.no_extras)   # before torch / HIP are loaded: the workers are forks

    import torch
    import torch.distributed as dist
    from rlao_amd import dist as aodist
    from rlao_amd.env import BatchedAOEnv

    rank, world = aodist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if args.config != "C2":
        line = config_headline(args, torch, dist, rank, world, local, cpu)
        if rank == 0:
            print(json.dumps(line))
        return
    n_local = args.envs_per_gpu or 256
    n_total = n_local * world
    K, W = args.steps, args.warmup

    env = BatchedAOEnv(n_envs=n_local, device=local, dtype=args.dtype, return_frame=True, env_index_offset=rank * n_local)
    env.set_params(dict(GEOMETRY, nLoop=W + 10 * K + 64), wfs_type="shackhartmann", camera=camera)
    timer = Timer(torch, dist, world, env.device)
    start_episode(env)
    returns = torch.zeros(n_local, device=env.device, dtype=env.tdtype)
    if world > 1:                                   # warm the communicator outside the timed regions
        aodist.all_gather_returns(returns, n_total)
    env.run_integrator(0, W)                        # W untimed warm-up steps
    env.accumulate_returns(returns)                 # every step adds its reward on the device (episode return)
    gathered = {}

    def region(r):
        env.run_integrator(W + (r % 8) * K, K)      # K closed-loop steps of every env: one library call, K x (1..3) launches
        gathered["returns"] = aodist.all_gather_returns(returns, n_total)

    times = timer.regions(region, args.min_seconds)
    env.accumulate_returns(None)
    s = stats(times)
    dt = s["median"]
    strehl = float(env._strehl.mean())

    # roofline leg: the same loop again with a hipEvent pair around every kernel launch
    n_prof = min(max(K, 100), 2 * K + 64)
    per_kernel, roof, step_bytes = kernel_profile(env, lambda: env.run_integrator(W + 8 * K, n_prof), n_local)
    extras = {}
    if not args.no_extras and world == 1:
        for cam2, key in (("ideal", "ideal_detector"), ("razor", "razor_camera"), ("papyrus", "photon_noise")):
            if cam2 == camera:
                continue
            set_camera(env, cam2)
            env.run_integrator(W, 10)
            t2 = timer.regions(lambda r: env.run_integrator(W + (r % 8) * K, K), args.min_seconds / 3)
            s2 = stats(t2)
            extras[key] = {"value": n_total * K / s2["median"], "unit": "env-steps/s", "ms_per_step": 1e3 * s2["median"] / K,
                           "repeats": len(t2), "camera": CAMERA_NOTE[cam2]}
        set_camera(env, camera)
        # the per-call Python API a trainer's rollout loop uses (MAIN/PO4AO/mbrl.py:64-89): action = gain * obs in torch
        obs = env.reset_soft()
        for i in range(10):
            obs = env.step(i, 0.5 * obs)[0]
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        n_api = 300
        for i in range(n_api):
            obs = env.step(10 + i % 32, 0.5 * obs)[0]
        t_host = time.perf_counter() - t1
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t1
        extras["step_api"] = {"us_per_call_host": 1e6 * t_host / n_api, "us_per_step": 1e6 * t_all / n_api,
                              "value": n_total * n_api / t_all, "unit": "env-steps/s",
                              "note": "env.step(i, 0.5 * obs) from Python, 256 envs, frame returned; host time = until the call returns"}
        env.return_frame = "view"                                 # the frame as an alias of the library's buffer: no 14.7 MB copy per step
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_api):
            obs = env.step(10 + i % 32, 0.5 * obs)[0]
        torch.cuda.synchronize()
        t_view = time.perf_counter() - t1
        env.return_frame = True
        extras["step_api"]["frame_view"] = {"us_per_step": 1e6 * t_view / n_api, "value": n_total * n_api / t_view,
                                            "note": 'BatchedAOEnv(return_frame="view")'}
        # every env its own wind (a trainer that draws the wind per run): per-env clocks on the device, the ring kernels on every
        # step.  LAST: the shard keeps its per-env clocks from here on.
        rs = np.random.RandomState(11)
        nl = env.param.nLayer
        env.set_wind_per_env(rs.uniform(5.0, 15.0, size=(n_local, nl)), rs.uniform(0.0, 360.0, size=(n_local, nl)), reset=True)
        env.run_integrator(W, 10)
        t3 = timer.regions(lambda r: env.run_integrator(W + (r % 8) * K, K), args.min_seconds / 3)
        s3 = stats(t3)
        extras["per_env_wind"] = {"value": n_total * K / s3["median"], "unit": "env-steps/s", "ms_per_step": 1e3 * s3["median"] / K,
                                  "repeats": len(t3), "note": "every env its own wind: 5-15 m/s, any direction (aoenv_set_wind_env); same camera as the headline"}
    if rank != 0:
        return
    pmc = load_pmc(n_local, camera, camera)
    attach_traffic(roof, pmc)
    fl = mfma_flops_per_env_step(env)
    out = {
        "metric": "AO env-steps/sec (batched loops)", "value": n_total * K / dt, "unit": "env-steps/s",
        "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": 1e3 * dt / K, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "repeats": len(times), "region_ms": {k: 1e3 * v for k, v in s.items()},
        "config": {"workload": "8m / 20x20 Shack-Hartmann, 256 batched envs per GPU, integrator closed loop "
                               "(BASELINE.json configs[1])",
                   "envs_per_gpu": n_local, "envs_total": n_total, "resolution": env.R, "n_valid_act": env.nValidAct,
                   "n_signal": env.nSignal, "layers": env.param.nLayer, "controller": "leaky integrator, gain 0.5",
                   "camera": CAMERA_NOTE[camera], "noise": args.noise,
                   "parallelism": f"env-shards x{world}, all-gather of episode returns"},
        "roofline": roof,
        "step_roofline": {"algorithmic_bytes_per_env_step": step_bytes,
                          "achieved_GBs": step_bytes * n_local / (dt / K) / 1e9,
                          "frac_of_hbm_peak": step_bytes * n_local / (dt / K) / 1e9 / HBM_PEAK_GBS},
        "mfma": {"flops_per_env_step": fl["total"], "split": {k: v for k, v in fl.items() if k != "total"},
                 "achieved_tflops": fl["total"] * n_local / (dt / K) / 1e12, "peak_tflops_f32": MFMA_F32_PEAK_TFLOPS,
                 "frac": fl["total"] * n_local / (dt / K) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                 "pmc": None if pmc is None else pmc.get("mfma", {}).get("env_step")},
        "kernels": per_kernel,
        "mean_strehl_last_step": strehl,
        "mean_episode_return": float(gathered["returns"].mean()),
    }
    out.update(extras)
    if cpu is not None:
        out["cpu_baseline"] = cpu
    env.close()
    del env
    torch.cuda.empty_cache()
    if world == 1 and not args.no_extras and args.configs.lower() != "none":
        out["configs"] = {}
        for name in [c.strip().upper() for c in args.configs.split(",") if c.strip()]:
            if name not in CONFIGS:
                raise SystemExit(f"unknown config {name}: choose from {sorted(CONFIGS)}")
            try:
                args.envs_per_gpu = None
                out["configs"][name] = bench_config(name, args, torch, timer)
            except Exception as exc:                           # a config leg must not take the headline line down with it
                out["configs"][name] = {"error": f"{type(exc).__name__}: {exc}"}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
