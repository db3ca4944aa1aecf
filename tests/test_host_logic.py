"""CPU: host-side logic that needs no GPU -- parameter parsing, sharding, wrappers, the warp spec of the oracle."""
import os
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import ao_oracle as O
from rlao_amd import calib
from rlao_amd import dist as aodist
from rlao_amd.wrappers import TimeDelayEnv, TorchWrapper

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_params_accept_both_reference_spellings():
    a = calib.params_from_args(SimpleNamespace(r0=0.1, L0=25, fractionnalR0=[0.7, 0.3], windSpeed=[5, 6],
                                               windDirection=[0, 90], altitude=[0, 0], nLoop=100, gainCL=0.3, modulation=0))
    b = calib.params_from_args(dict(r0=0.1, L0=25, fractionalR0=[0.7, 0.3], windSpeed=[5, 6], windDirection=[0, 90],
                                    altitude=[0, 0], nLoop=100, gainCL=0.3))
    assert a.fractionalR0 == b.fractionalR0 == [0.7, 0.3] and a.nLayer == 2
    assert a.resolution == 120 and a.nActuator == 21 and a.modulation == 0 and a.extra == {}
    assert calib.params_from_args(dict(savedir="x")).extra == {"savedir": "x"}
    with pytest.raises(ValueError):
        calib.params_from_args(dict(fractionalR0=[1.0], windSpeed=[1.0, 2.0]))


def test_shard_bounds_cover_everything_once():
    for n, w in [(256, 1), (256, 8), (4096, 8), (10, 4), (3, 8)]:
        seen = []
        for r in range(w):
            lo, hi = aodist.shard_bounds(n, r, w)
            assert 0 <= lo <= hi <= n
            seen += list(range(lo, hi))
        assert seen == list(range(n))
    with pytest.raises(ValueError):
        aodist.shard_bounds(8, 8, 8)


class _FakeEnv:
    """NumPy env with the reference's return convention (for the wrappers)."""
    output = "numpy"
    nActuator = 3
    n_envs = 1

    def __init__(self):
        self.seen = []

    def reset_soft(self):
        return np.ones((3, 3))

    def step(self, i, action):
        self.seen.append(np.array(action, dtype=float))
        return np.full((3, 3), float(i)), np.zeros((2, 2)), -1.5, 0.25, False, {"strehl": 0.25}


def test_torch_wrapper_reference_convention():
    env = TorchWrapper(_FakeEnv())
    obs = env.reset_soft()
    assert obs.dtype == torch.float32 and obs.shape == (3, 3)
    nxt, wfsf, rew, sr, done, info = env.step(4, torch.full((3, 3), 2.0))
    assert nxt.dtype == torch.float32 and float(nxt[0, 0]) == 4.0 and rew == -1.5 and sr == 0.25 and done is False
    assert info[0][0] == "strehl" and info[0][1].dtype == torch.float32
    assert env.nActuator == 3                                   # attribute forwarding


def test_time_delay_env_is_a_fifo():
    inner = _FakeEnv()
    env = TimeDelayEnv(inner, 2)
    env.reset_soft()
    for i in range(4):
        env.step(i, np.full((3, 3), float(i + 1)))
    got = [float(a[0, 0]) for a in inner.seen]
    assert got == [0.0, 0.0, 1.0, 2.0]


def test_warp_spec_properties():
    """Frozen spec of skimage.warp(order=3) for translations (the one stage the reference cannot pin)."""
    rs = np.random.RandomState(0)
    img = rs.randn(20, 23)
    np.testing.assert_array_equal(O.warp_translate(img, 0.0, 0.0), img)
    for tx, ty in [(1, 0), (0, -1), (-1, 1)]:                    # integer shifts move pixels exactly, zero fill
        out = O.warp_translate(img, tx, ty)
        ref = np.zeros_like(img)
        ys, xs = np.mgrid[0:20, 0:23]
        sy, sx = ys - ty, xs - tx
        ok = (sy >= 0) & (sy < 20) & (sx >= 0) & (sx < 23)
        ref[ok] = img[sy[ok], sx[ok]]
        np.testing.assert_array_equal(out, np.clip(ref, img.min(), img.max()))
    # Catmull-Rom reproduces quadratics exactly away from the border
    yy, xx = np.mgrid[0:30, 0:30].astype(float)
    quad = 0.3 * xx ** 2 - 0.2 * xx * yy + 0.1 * yy ** 2 + xx - 2 * yy + 5
    out = O.warp_translate(quad, 0.37, -0.61)
    want = 0.3 * (xx - 0.37) ** 2 - 0.2 * (xx - 0.37) * (yy + 0.61) + 0.1 * (yy + 0.61) ** 2 + (xx - 0.37) - 2 * (yy + 0.61) + 5
    want = np.clip(want, quad.min(), quad.max())                 # clip=True: the minimum of the bowl is cut
    np.testing.assert_allclose(out[3:-3, 3:-3], want[3:-3, 3:-3], atol=1e-10)
    # output is clipped to the input range
    spike = np.zeros((12, 12))
    spike[6, 6] = 1.0
    out = O.warp_translate(spike, 0.5, 0.5)
    assert out.min() >= 0.0 and out.max() <= 1.0


def test_catmull_rom_weights_match_the_nested_form():
    """The HIP kernels use tap weights; the oracle uses skimage's nested cubic.  Same polynomial."""
    for x in [0.0, 0.1, 0.5, 0.999]:
        w = np.array([0.5 * (-x ** 3 + 2 * x ** 2 - x), 0.5 * (3 * x ** 3 - 5 * x ** 2 + 2),
                      0.5 * (-3 * x ** 3 + 4 * x ** 2 + x), 0.5 * (x ** 3 - x ** 2)])
        f = np.array([0.3, -1.2, 2.5, 0.7])
        assert abs(w @ f - O._cubic(x, *f)) < 1e-14
        assert abs(w.sum() - 1) < 1e-14


class _FakeBatchedEnv:
    """Torch env on the CPU with the batched return convention (for HistoryEnv)."""
    output = "torch"
    nActuator = 3
    n_envs = 2
    device = "cpu"

    def __init__(self):
        self.seen = []
        self.param = type("P", (), {"nLoop": 5})()

    def vec_to_img(self, v, use_torch=False):
        img = torch.zeros(v.shape[:-1] + (3, 3))
        img[..., [0, 1, 2], [0, 1, 2]] = v
        return img

    def step(self, i, action):
        self.seen.append((i, action.clone()))
        obs = torch.full((2, 3, 3), float(len(self.seen)))
        return obs, None, -obs.sum(dim=(1, 2)), torch.tensor([0.1, 0.2]) * len(self.seen), torch.zeros(2, dtype=torch.bool), {}


def test_history_env_rolls_and_delays():
    """gymnasium-style facade (MAIN/OOPAOEnv/OOPAOEnv_VPG.py:553-608, 660-681): newest observation at index 0, Strehl as the
    reward, action FIFO, frame counter wrapping at nLoop, command vectors scattered to images."""
    from rlao_amd.wrappers import HistoryEnv
    inner = _FakeBatchedEnv()
    env = HistoryEnv(inner, n_history=4, delay=2)
    assert env.observation_space.shape == (2, 4, 3, 3) and env.action_space.shape == (2, 3, 3)
    for k in range(7):
        hist, rew, term, trunc, info = env.step(torch.full((2, 3, 3), float(k + 1)))
    assert hist.shape == (2, 4, 3, 3)
    assert [float(hist[0, j, 0, 0]) for j in range(4)] == [7.0, 6.0, 5.0, 4.0]      # newest first
    assert torch.allclose(rew, torch.tensor([0.7, 1.4])) and info["strehl"] is rew
    assert term.dtype == torch.bool and not bool(term.any()) and not bool(trunc.any())
    assert [float(a[0, 0, 0]) for _, a in inner.seen] == [0.0, 1.0, 2.0, 3.0, 4.0, 5.0, 6.0]   # one extra frame of delay
    assert [i for i, _ in inner.seen] == [0, 1, 2, 3, 4, 0, 1]                          # wraps at nLoop = 5
    env.step(torch.tensor([[1.0, 2.0, 3.0], [4.0, 5.0, 6.0]]))                           # vectors of 3 "actuators"
    assert inner.seen[-1][1].shape == (2, 3, 3)
    single = HistoryEnv(_FakeEnv(), n_history=3, delay=1)
    h, r, t, tr, inf = single.step(np.full((3, 3), 2.0))
    assert isinstance(h, np.ndarray) and h.shape == (3, 3, 3) and r == 0.25 and t is False
    assert float(single._env.seen[-1][0, 0]) == 2.0


def test_bench_spawns_its_ranks_for_gpus_n():
    """`python bench.py --gpus N` outside a torchrun environment starts the N ranks as child processes of
    torch.distributed.run on 127.0.0.1 (dry run: the command only) and forwards its own flags."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "20", "--warmup", "5",
                          "--dry-run-launch"], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr
    cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(REPO, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "20", "--warmup", "5"]
    # BASELINE configs[3] / configs[4] as they are stated: `--gpus 8 --config C4|C5` goes to the ranks unchanged
    for cfg in ("C4", "C5"):
        out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "8", "--config", cfg, "--dry-run-launch"],
                             capture_output=True, text=True, env=env, timeout=120)
        assert out.returncode == 0, out.stderr
        cmd = json.loads(out.stdout.strip().splitlines()[-1])["launch"]
        assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--config") + 1] == cfg
    import bench
    assert bench.CONFIGS["C4"]["envs"] * 8 == 4096 and bench.CONFIGS["C5"]["envs"] * 8 == 2048      # BASELINE.json configs[3], [4]
    # inside a rank environment nothing is spawned: the same flags fall through to the benchmark itself
    import bench
    args = type("A", (), {"gpus": 4, "dry_run_launch": False})()
    os.environ["RANK"] = "0"
    try:
        assert bench.maybe_spawn_ranks(args, []) is None
    finally:
        del os.environ["RANK"]


def test_bench_cpu_baseline_runs_the_oracle_with_the_camera():
    import bench
    out = bench.cpu_baseline("papyrus", 0.3, all_cores=False)
    assert out["cores"] == 1 and out["kind"] == "port" and out["value"] > 0 and "camera: papyrus" in out["sample"]


def test_atmosphere_clock_source_matches_the_oracle(tmp_path):
    """rlao_amd/csrc/common.hpp::clock_subpixel / taps_from_buff -- ONE source for the host clock (one wind per shard) and for the
    device clocks of per-env winds (k_ring_prepare_env) -- compiled for the host and stepped next to the oracle's updateLayer
    arithmetic (OOPAO/Atmosphere.py:392-407): accumulators and pixel crossings bit for bit, the warp's tap offsets exactly, its four
    Catmull-Rom weights against the oracle's cubic evaluated on unit vectors."""
    import shutil
    import subprocess
    from oracle import ao_oracle as O
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = tmp_path / "clock_driver"
    subprocess.run([hipcc, "-O1", "-std=c++17", "-x", "hip", "--cuda-host-only", f"-I{repo}/include", f"-I{repo}/rlao_amd/csrc",
                    os.path.join(repo, "tests", "native", "clock_driver.cpp"), "-o", str(exe)], check=True, capture_output=True)
    cases = [(0.1537, 0.047), (-0.9, 0.0), (0.0, -0.31), (0.73, -0.73), (1e-3, 0.999), (-0.25, -0.5)]
    n = 300
    out = subprocess.run([str(exe)], input="".join(f"{rx!r} {ry!r} {n}\n" for rx, ry in cases), text=True, capture_output=True, check=True)
    rows = np.array([[float(v) for v in line.split()] for line in out.stdout.strip().splitlines()]).reshape(len(cases), n, 14)
    unit = np.eye(4)
    for (rx, ry), got in zip(cases, rows):
        ratio = np.array([rx, ry])
        buff = np.zeros(2)
        crossings = 0
        for i in range(n):
            buff = buff + (np.abs(ratio) % 1) * np.sign(ratio)                 # ao_oracle.OracleLayer.update
            step = np.zeros(2)
            if np.abs(buff[0]) >= 1 or np.abs(buff[1]) >= 1:
                step = 1 * np.sign(buff)
                step[np.where(np.abs(buff) < 1)] = 0
            buff = (np.abs(buff) % 1) * np.sign(buff)
            crossings += int(step.any())
            assert got[i, 0] == step[0] and got[i, 1] == step[1], (rx, ry, i)
            assert got[i, 2] == buff[0] and got[i, 3] == buff[1], (rx, ry, i)   # bit for bit
            # warp_translate(map, tx = buff_x, ty = buff_y): sampling point r - ty -> offset floor(-ty), fraction, cubic weights
            for d, (b, col) in enumerate(((buff[1], 6), (buff[0], 10))):
                f = -b
                k = np.floor(f)
                assert got[i, 4 + d] == k
                w = np.array([O._cubic(f - k, *unit[j]) for j in range(4)])
                np.testing.assert_allclose(got[i, col:col + 4], w, rtol=0, atol=2e-15)
        assert crossings >= int(n * np.abs(ratio).max()) - 1                      # at least the faster axis' pixel count


def test_committed_counter_summaries_match_the_bench_workloads():
    """bench.py reads `roofline.traffic` from the committed rocprofv3 --pmc summaries (profiles/r03_*_pmc.json): each must describe the
    workload of its bench line (env count, camera) and name a production entry for every kernel of the line's dominant stage -- a
    summary written with another camera key silently turns the traffic fields of a line into null."""
    import bench
    stage = {"C3": "pyramid", "C3M": "pyramid", "C4": "phase", "C5": "env_step"}
    for name, cfg in bench.CONFIGS.items():
        pmc = bench.load_pmc(cfg["envs"], name)
        assert pmc is not None, name
        total, parts = bench.pmc_traffic(pmc, stage[name])
        assert total and total > 0 and len(parts) >= 1, (name, parts)
        assert not pmc["entries_above_hbm_peak"], (name, pmc["entries_above_hbm_peak"])
    for cam in ("papyrus", "ideal"):
        pmc = bench.load_pmc(256, cam, cam)
        assert pmc is not None and bench.pmc_traffic(pmc, "env_step")[0] > 0, cam
