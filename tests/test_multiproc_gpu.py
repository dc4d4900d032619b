"""N > 1 data path on ONE GPU box: two ranks (gloo) share device 0, each maps its contiguous, cost-balanced share of one read set
(read_ranges_by_cost -- SURVEY 8e: a mini-batch split into contiguous ranges balanced by DP cost, results gathered in input
order), rank 0 concatenates the per-rank SAM and compares it with the single-process golden SAM.  No data-path collective: the
only traffic is the gather of the finished text."""
import os
import socket
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, kind, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch  # noqa: F401  (first: one HIP runtime)
    import torch.distributed as dist
    from conftest import load_pkg
    from fixture_io import OVERRIDES, SETS, read_fasta, reads_of
    pkg = load_pkg()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        base, stem, preset = SETS[kind]
        names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
        reads = reads_of(kind)
        bounds = pkg.read_ranges_by_cost([len(r[1]) for r in reads], world, band=1000)
        mine = reads[bounds[rank]:bounds[rank + 1]]
        ctx = pkg.Context(0)  # every rank on the one device of the box (the driver's 8-GPU run gives each rank its own)
        m = pkg.Mapper(ctx, names, seqs, preset=preset, **OVERRIDES.get(kind, {}))
        sam = m.sam_batch(m.map([r[1] for r in mine]), mine) if mine else ""
        m.close()
        ctx.close()
        parts = [None] * world if rank == 0 else None
        dist.gather_object((bounds[rank], bounds[rank + 1], sam), parts, dst=0)
        if rank == 0:
            q.put(parts)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["hifi_sv", "sr"])
def test_two_ranks_concatenate_to_the_single_process_sam(kind):
    import torch.multiprocessing as mp
    from fixture_io import golden_sam, reads_of
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, kind, q)) for r in range(world)]
    for p in ps:
        p.start()
    try:
        # poll instead of one blocking get: a rank that dies before rank 0 posts the result must fail the test at once, and no worker
        # may be left behind holding device 0 (a survivor would sit in gather_object for the rest of the session)
        import queue
        import time
        parts, t_end = None, time.time() + 600
        while parts is None:
            try:
                parts = q.get(timeout=2)
            except queue.Empty:
                dead = [p.exitcode for p in ps if p.exitcode not in (None, 0)]
                assert not dead, "a rank exited with %s before the result was posted" % dead
                assert time.time() < t_end, "no result within 600 s"
        for p in ps:
            p.join(120)
            assert p.exitcode == 0
    finally:
        for p in ps:
            if p.is_alive():
                p.terminate()
            p.join(30)
    n = len(reads_of(kind))
    assert parts[0][0] == 0 and parts[-1][1] == n and all(a[1] == b[0] for a, b in zip(parts, parts[1:]))  # contiguous, every read once
    assert all(hi > lo for lo, hi, _ in parts)  # both ranks had work
    assert "".join(p[2] for p in parts) == "".join(l + "\n" for l in golden_sam(kind))
