#!/usr/bin/env python3
"""Full-size runs of the BASELINE.json configurations other than the headline one (which is bench.py):
    python scripts/run_config.py C3 | C3mod | C4 | C5  [n_envs] [steps]
Builds the env (calibration on the GPU included), steps the closed loop, prints one JSON line with the throughput
and size-independent checks: float32 vs float64 agreement of one measurement, batch invariance (env 0 == env N-1 when
seeded alike), finite outputs.  Single GPU: C4 / C5 run their per-GPU shard (n_envs / 8)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rlao_amd.env import BatchedAOEnv
from rlao_amd import _lib as L

CONFIGS = {
    # 8 m / 40x40 Pyramid, 1024 envs (test_po4ao.sh geometry scaled to 8 m), unmodulated as papyrus_config.yaml:17
    "C3": dict(wfs="pyramid", n_envs=1024, geo=dict(diameter=8.0, nSubaperture=40, nPixelPerSubap=6, r0=0.13, L0=30.0,
               windSpeed=[10.0], windDirection=[72.0], fractionalR0=[1.0], altitude=[0.0], nModes=200, modulation=0.0)),
    "C3mod": dict(wfs="pyramid", n_envs=128, geo=dict(diameter=8.0, nSubaperture=40, nPixelPerSubap=6, r0=0.13, L0=30.0,
                  windSpeed=[10.0], windDirection=[72.0], fractionalR0=[1.0], altitude=[0.0], nModes=200, modulation=3.0)),
    # ELT scale 39 m / 80x80 SH, 4096 envs over 8 GPUs = 512 per GPU
    "C4": dict(wfs="shackhartmann", n_envs=512, geo=dict(diameter=39.0, nSubaperture=80, nPixelPerSubap=6, r0=0.13, L0=30.0,
               windSpeed=[10.0], windDirection=[72.0], fractionalR0=[1.0], altitude=[0.0], nModes=300)),
    # 3-layer atmosphere + two chained DMs (20x20 and 10x10 actuator pitch), C2 geometry, 2048 envs over 8 GPUs = 256 per GPU
    "C5": dict(wfs="shackhartmann", n_envs=256, second_dm=dict(nSubaperture=10), geo=dict(diameter=8.0, nSubaperture=20, nPixelPerSubap=6, r0=0.13, L0=30.0,
               windSpeed=[10.0, 12.0, 11.0], windDirection=[0.0, 72.0, 144.0], fractionalR0=[0.45 / 0.65, 0.1 / 0.65, 0.1 / 0.65],
               altitude=[0.0, 1000.0, 5000.0], nModes=50)),
}
CONFIGS["C5_1dm"] = dict(CONFIGS["C5"], second_dm=None)         # the same atmosphere with the single 20x20 DM (fused step kernel)


def main():
    name = sys.argv[1]
    cfg = CONFIGS[name]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["n_envs"]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    t0 = time.perf_counter()
    env = BatchedAOEnv(n_envs=n, device=0, dtype="f32", return_frame=False, env_seed_stride=0)
    env.set_params(dict(cfg["geo"], nLoop=steps + 40), wfs_type=cfg["wfs"], second_dm=cfg.get("second_dm"))
    t_init = time.perf_counter() - t0
    print(f"[{name}] init {t_init:.1f} s: R={env.R} A={env.nValidAct} nSignal={env.nSignal} cam={env.cam_res}", file=sys.stderr, flush=True)
    env.generate_new_phase_screen(17)
    env.dm.coefs = 0
    env.measure()
    sig32 = env._shard.download(L.B_SIGNAL, (n, env.nSignal))
    obs = env.reset_soft()
    # float64 reference shard of 1 env, same seed: one measurement
    e64 = BatchedAOEnv(n_envs=1, device=0, dtype="f64", return_frame=False)
    e64.set_params(dict(cfg["geo"], nLoop=8), wfs_type=cfg["wfs"], second_dm=cfg.get("second_dm"))
    e64.generate_new_phase_screen(17)
    e64.dm.coefs = 0
    e64.measure()
    sig64 = e64._shard.download(L.B_SIGNAL, (1, e64.nSignal))[0]
    e64.close()
    env.run_integrator(0, 20)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    o, r, s = env.run_integrator(20, steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t1
    out = {"config": name, "n_envs": n, "steps": steps, "resolution": env.R, "n_valid_act": env.nValidAct, "n_signal": env.nSignal,
           "init_s": round(t_init, 1), "us_per_step": 1e6 * dt / steps, "env_steps_per_s": n * steps / dt,
           "slopes_f32_vs_f64_max_abs": float(np.abs(sig32[0] - sig64).max()), "slopes_rms": float(np.sqrt((sig64 ** 2).mean())),
           "batch_invariant": bool((sig32[0] == sig32[-1]).all() and bool((o[0] == o[-1]).all())),
           "finite": bool(torch.isfinite(o).all() and torch.isfinite(r).all() and torch.isfinite(s).all()),
           "mean_strehl": float(s.mean()), "residual_nm_last": float(np.mean(env.residual[20 + steps - 1]))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
