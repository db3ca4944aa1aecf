"""A/B timing of kernel builds on one GPU box: the headline bench (no extras) with each library in turn, ROUNDS times interleaved
(A B A B ...: clocks and neighbours drift between runs), median per library.
    python scripts/ab_bench.py [--noise off|photon|razor] [--rounds 3] [--config C2] name=path/to/libaoenv.so ...   ("main" = the tree's library)
"""
import json, os, statistics, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
noise, rounds, config = "photon", 3, "C2"
while args and args[0].startswith("--"):
    k, v = args[0], args[1]
    args = args[2:]
    if k == "--noise": noise = v
    elif k == "--rounds": rounds = int(v)
    elif k == "--config": config = v
libs = [a.split("=", 1) for a in args]
res = {n: {"step_us": [], "kernel_us": []} for n, _ in libs}
for r in range(rounds):
    for name, path in libs:
        env = dict(os.environ)
        if path != "main":
            env["AOENV_LIB"] = os.path.join(REPO, path)
        cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "20", "--warmup", "5", "--no-extras", "--no-cpu-baseline", "--min-seconds", "0.4",
               "--noise", noise, "--config", config]
        out = subprocess.run(cmd, env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print(name, "FAILED", out.stderr[-400:])
            continue
        d = json.loads(out.stdout.strip().splitlines()[-1])
        res[name]["step_us"].append(1e3 * d["ms_per_step"])
        dom = d["roofline"]["kernel"]
        res[name]["kernel_us"].append(d["kernels"][dom]["avg_us"])
for name, v in res.items():
    if v["step_us"]:
        print(f"{name:24s} step {statistics.median(v['step_us']):8.2f} us  [{min(v['step_us']):.2f} .. {max(v['step_us']):.2f}]   dominant kernel "
              f"{statistics.median(v['kernel_us']):8.2f} us  [{min(v['kernel_us']):.2f} .. {max(v['kernel_us']):.2f}]", flush=True)
