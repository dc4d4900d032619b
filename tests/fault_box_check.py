"""run by tests/test_map_gpu.py::test_degenerate_box_fails_its_read_not_the_batch in a process of its own, with GDIET_FAULT_BOX=5 in
the environment (the library reads it when it first maps): read 5 of every batch is treated as having a degenerate DP box"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (first: one HIP runtime)
from conftest import load_pkg  # noqa: E402
from fixture_io import OVERRIDES, SETS, golden_sam, read_fasta, reads_of  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "hifi"  # "sr": the box stage runs on the device (map_sr_box_kernel marks the read there)
pkg = load_pkg()
ctx = pkg.Context(0)
base, stem, preset = SETS[kind]
names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
reads = reads_of(kind)
m = pkg.Mapper(ctx, names, seqs, preset=preset, **OVERRIDES.get(kind, {}))
victim = reads[int(os.environ["GDIET_FAULT_BOX"])][0]
assert any(l.split("\t")[0] == victim and l.split("\t")[1] != "4" for l in golden_sam(kind)), "the victim must be a read that maps"
want = [l for l in golden_sam(kind) if l.split("\t")[0] != victim]


def check(res):
    lines = [l for l in m.sam_batch(res, reads).split("\n") if l]
    mine = [l for l in lines if l.split("\t")[0] == victim]
    assert len(mine) == 1 and mine[0].split("\t")[1] == "4", mine[:1]  # the victim: one unmapped record
    assert [l for l in lines if l.split("\t")[0] != victim] == want      # everybody else: the golden records
    last, total, what = m.failed_reads()
    assert last == 1 and "degenerate DP box" in what, (last, what)


check(m.map([r[1] for r in reads]))
b = m.upload([r[1] for r in reads])
check(m.wait(m.submit(b)))  # ... and through a lane of the batches-in-flight path
assert m.failed_reads()[1] == 2
m.free_batch(b)
m.close()
ctx.close()
print("ok")
