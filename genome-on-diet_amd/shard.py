"""Multi-GPU form of the path: reads are independent, so ranks own disjoint read ranges, the index is
replicated, and there is no data-path collective.  The only cross-rank traffic is the bookkeeping below (a barrier
around the timed region, MAX of the elapsed time, SUM of the mapped bases); on GPUs it goes over RCCL ("nccl"),
in the CPU tests over gloo."""
import time


def read_range(n_reads, rank, world):
    """contiguous share of a mini-batch for `rank`: sizes differ by at most one read, order of reads preserved
    (what a multi-process host would hand each GPU from one kt_for batch, reference map.c:2132)"""
    base, rem = divmod(n_reads, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def read_ranges_by_cost(lens, world, band=1000):
    """contiguous ranges, one per rank, balanced by the DP cost of the reads rather than by their number: the
    cost of a read is its number of DP cells, (2 * len - 1) * min(band + 1, len).  Returns world + 1 boundaries; order of reads
    preserved (results are emitted in input order)."""
    import numpy as np
    lens = np.asarray(lens, np.int64)
    cost = (2 * lens - 1).clip(min=0) * np.minimum(band + 1, lens)
    cum = np.concatenate([[0], np.cumsum(cost)])
    total = int(cum[-1])
    bounds = [0]
    for r in range(1, world):
        bounds.append(max(bounds[-1], int(np.searchsorted(cum, total * r / world, side="left"))))
    bounds.append(len(lens))
    return bounds


def effective_cpus():
    """CPUs this process may really use: min(os.cpu_count(), affinity mask, cgroup CPU quota).  A container with a 16-CPU quota on a
    256-thread host reports 256 from os.cpu_count(); workers sized by that are throttled together every scheduling period."""
    import math
    import os
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:  # cgroup v2
            q, p = f.read().split()[:2]
        if q != "max" and int(p) > 0:
            n = min(n, max(1, math.ceil(int(q) / int(p))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = int(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = int(f.read())
            if q > 0 and p > 0:
                n = min(n, max(1, math.ceil(q / p)))
        except (OSError, ValueError):
            pass
    return max(1, n)


def rank_seed(base_seed, rank):
    """weak scaling: every rank draws its own reads (same distribution, different stream)"""
    return base_seed + rank


class JobClock:
    """barrier-bracketed wall clock + whole-job aggregation; `dist` is torch.distributed or None (single process)"""

    def __init__(self, dist=None, device=None, sync=None):
        self.dist, self.device, self.sync = dist, device, sync or (lambda: None)
        self.t0 = None

    def _barrier(self):
        if self.dist is not None:
            self.dist.barrier()

    def start(self):
        self.sync()
        self._barrier()
        self.sync()
        self.t0 = time.perf_counter()

    def stop(self):
        self.sync()
        self._barrier()
        return time.perf_counter() - self.t0

    def aggregate(self, elapsed, units):
        """(max over ranks of elapsed, sum over ranks of units)"""
        if self.dist is None:
            return float(elapsed), float(units)
        import torch
        t = torch.tensor([float(elapsed)], dtype=torch.float64, device=self.device)
        u = torch.tensor([float(units)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        self.dist.all_reduce(u, op=self.dist.ReduceOp.SUM)
        return float(t.item()), float(u.item())
