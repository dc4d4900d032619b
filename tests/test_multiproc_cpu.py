"""N > 1 control path of bench.py on CPU: two processes over gloo exercise exactly the helpers the GPU bench uses
(read sharding without overlap, barrier-bracketed clock, MAX of time and SUM of mapped bases).  The data path itself has
no collective, so there is nothing else to rehearse across ranks."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_pkg


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    pkg = load_pkg()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        clock = pkg.JobClock(dist, torch.device("cpu"))
        lo, hi = pkg.read_range(1001, rank, world)
        clock.start()
        units = sum(range(lo, hi))  # stands for "bases of the reads this rank mapped"
        elapsed = clock.stop()
        fake = 1.0 + rank  # rank 1 is the slow one
        tmax, total = clock.aggregate(fake, units)
        q.put((rank, lo, hi, elapsed, tmax, total, pkg.rank_seed(5, rank)))
    finally:
        dist.destroy_process_group()


def test_two_ranks_shard_and_aggregate():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in ps:
        p.start()
    out = sorted(q.get(timeout=120) for _ in range(world))
    for p in ps:
        p.join(60)
        assert p.exitcode == 0
    (r0, lo0, hi0, _, tmax0, tot0, s0), (r1, lo1, hi1, _, tmax1, tot1, s1) = out
    assert (lo0, hi1) == (0, 1001) and hi0 == lo1 and abs((hi0 - lo0) - (hi1 - lo1)) <= 1
    assert tmax0 == tmax1 == 2.0  # MAX over ranks
    assert tot0 == tot1 == float(sum(range(1001)))  # SUM over ranks, nothing lost or double-counted
    assert s0 != s1


@pytest.mark.parametrize("n,world", [(0, 1), (1, 4), (7, 8), (4096, 8), (4097, 3)])
def test_read_range_partitions_every_read_once(n, world):
    pkg = load_pkg()
    cover = []
    for r in range(world):
        lo, hi = pkg.read_range(n, r, world)
        assert 0 <= lo <= hi <= n
        cover += list(range(lo, hi))
    assert cover == list(range(n))


def test_cost_balanced_ranges():
    """mixed read lengths: every read in exactly one range, ranges contiguous, DP cost per rank within one read of the ideal"""
    import numpy as np
    pkg = load_pkg()
    rng = np.random.default_rng(3)
    lens = np.concatenate([rng.integers(100, 300, size=5000), rng.integers(5000, 25000, size=300), rng.integers(100, 300, size=2000)])
    for world in (1, 2, 8):
        b = pkg.read_ranges_by_cost(lens, world)
        assert b[0] == 0 and b[-1] == len(lens) and all(x <= y for x, y in zip(b, b[1:])) and len(b) == world + 1
        cost = (2 * lens - 1) * np.minimum(1001, lens)
        per = [int(cost[b[r]:b[r + 1]].sum()) for r in range(world)]
        assert max(per) - min(per) <= 2 * int(cost.max())


def test_effective_cpus_respects_quota_and_affinity():
    """the worker count is derived from what the container may use, not from the host's thread count; the library's C entry point
    (gdiet_hip_effective_cpus) and the Python helper agree"""
    import os
    from conftest import load_pkg
    pkg = load_pkg()
    n = pkg.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert n <= len(os.sched_getaffinity(0))
    lib = pkg.load_library()
    assert lib.gdiet_hip_effective_cpus() == n


def test_c_read_ranges_equal_the_python_mirror():
    """gdiet_hip_read_ranges_by_cost (what gdiet_hip_map_batch_multi cuts a mini-batch with; host arithmetic, callable without a GPU)
    against shard.read_ranges_by_cost: the same boundaries for HiFi-, ONT- and short-read-like length mixes, 1..8 parts, empty input"""
    import numpy as np
    from conftest import load_pkg
    pkg = load_pkg()
    rng = np.random.default_rng(3)
    cases = [np.clip(rng.normal(15000, 2000, 5120), 5000, 25000).astype(np.int32), np.clip(rng.lognormal(np.log(50000), 0.35, 700), 5000, 150000).astype(np.int32),
             np.full(4096, 150, np.int32), rng.integers(0, 400, 333).astype(np.int32), np.zeros(0, np.int32), np.array([7], np.int32)]
    for lens in cases:
        for parts in (1, 2, 3, 4, 8):
            for band in (150, 1000, 1300):
                want = pkg.read_ranges_by_cost(lens, parts, band=band)
                got = pkg.read_ranges_by_cost_c(lens, parts, band=band)
                assert got == [int(x) for x in want], (len(lens), parts, band)
                assert got[0] == 0 and got[-1] == len(lens) and all(a <= b for a, b in zip(got, got[1:]))
