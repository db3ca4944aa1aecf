"""CPU, world_size = 2, gloo: the only collective of the multi-GPU path (all-gather of per-env episode
returns over uneven shards) and the shard seeding convention."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from rlao_amd import dist as aodist


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = aodist.init_from_env(backend="gloo")
    lo, hi = aodist.shard_bounds(n_total, r, w)
    local = torch.arange(lo, hi, dtype=torch.float32) * 10 + 1            # "episode return" of global env index
    full = aodist.all_gather_returns(local, n_total)
    ok = torch.equal(full, torch.arange(n_total, dtype=torch.float32) * 10 + 1)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    q.put((rank, bool(ok), float(t[0]), (lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [8, 7])
def test_all_gather_returns_world2(n_total):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [2.0, 2.0]
    assert res[0][3][1] == res[1][3][0] and res[1][3][1] == n_total


def _bench_worker(rank, world, port, q):
    """bench.py's timer over two ranks with a configs[4]-shaped shard (256 envs per rank): K-step regions ending in the
    all-gather of the per-env returns, MAX over ranks, per-rank spread."""
    import sys
    import time
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    r, w = aodist.init_from_env(backend="gloo")
    n = bench.CONFIGS["C5"]["envs"]
    n_total = n * w
    timer = bench.Timer(torch, dist, w, torch.device("cpu"))
    returns = torch.zeros(n)
    got = {}

    def region(k):
        time.sleep(0.02 * (rank + 1))                             # rank 1 is the slow one
        returns.add_(torch.arange(rank * n, rank * n + n, dtype=torch.float32))    # "reward" = global env index (env_index_offset + e)
        got["all"] = aodist.all_gather_returns(returns, n_total)

    times = timer.regions(region, min_seconds=0.0, min_repeats=3, max_repeats=3)
    ok = torch.equal(got["all"], 3 * torch.arange(n_total, dtype=torch.float32))
    q.put((rank, bool(ok), times, timer.rank_spread, n_total))
    dist.barrier()
    dist.destroy_process_group()


def test_bench_timer_and_config_shards_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bench_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res) and res[0][4] == 512
    assert res[0][2] == res[1][2]                                  # both ranks report the same (MAX) region times
    for (fast, slow), dt in zip(res[0][3], res[0][2]):
        assert fast < slow and dt == slow and slow >= 0.04
