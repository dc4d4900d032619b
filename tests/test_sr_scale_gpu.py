"""GPU parity at a size the fixtures do not have: 120 000 synthetic 150-base reads (1 % substitutions, a sixth of them with indels, both
strands) against a 60 Mbp synthetic reference with repeats and N runs, through the ShortReads whole path of the library -- pre-filter
with its proofs, skewed pipelines, grouped kernels for the rare geometries, device-side box stage, P1 in integer arithmetic -- and through
the reference's own GDiet_avx (oracle/_ref/gdiet_sr_avx, compiled from /root/reference by oracle/build_ref.py; the test is skipped where
it was not built): every SAM line must be the same."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

SR_CMD = "-ax sr -Z 10 -W 2 -i 2 -k 21 -w 11 -N 1 -r 0.05,150,200 -n 0.95,0.3 -s 100 --AF_max_loc 2 --secondary=yes -a".split()  # tests/golden/sr/sr.cmd
EXE = os.path.join(ROOT, "oracle", "_ref", "gdiet_sr_avx")


def test_synthetic_short_read_batch_matches_gdiet_avx(gpu_ctx, pkg, tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("oracle/_ref/gdiet_sr_avx not built (needs /root/reference at build time)")
    sys.path.insert(0, ROOT)
    import bench
    from tools.bench_variant import synth_reads
    names, contigs = bench.synth_reference(60, seed=11)
    seqs = synth_reads(np.random.default_rng(23), contigs, 120000, "sr")
    reads = [("r%d" % i, s, b"I" * len(s)) for i, s in enumerate(seqs)]
    fa, fq = str(tmp_path / "c.fa"), str(tmp_path / "r.fq")
    with open(fa, "wb") as f:
        for nm, c in zip(names, contigs):
            f.write(b">" + nm.encode() + b"\n" + np.asarray(c).tobytes() + b"\n")
    with open(fq, "wb") as f:
        for nm, s, q in reads:
            f.write(b"@" + nm.encode() + b"\n" + s + b"\n+\n" + q + b"\n")
    r = subprocess.run([EXE, "-t", "8"] + SR_CMD + [fa, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode(errors="ignore")[-2000:]
    want = [l for l in r.stdout.decode().split("\n") if l and not l.startswith("@")]
    m = pkg.Mapper(gpu_ctx, names, contigs, preset="sr")
    try:
        res = m.map([s for _, s, _ in reads])
        got = [l for l in m.sam_batch(res, reads).split("\n") if l]
        assert gpu_ctx.last_kernel_mask() & 16  # the batch's alignments ran as skewed pipelines

        def norm(l):  # (the ms:i tag of a reverse-strand record over reference Ns is undefined in the reference: bench._ms_undefined)
            return "\t".join(x for x in l.split("\t") if not x.startswith("ms:i:")) if bench._ms_undefined(l) else l
        assert len(got) == len(want)
        bad = [i for i in range(len(want)) if norm(got[i]) != norm(want[i])]
        assert not bad, (len(bad), got[bad[0]][:400], want[bad[0]][:400])
        assert sum(1 for l in want if l.split("\t")[1] != "4") > 100000  # (most of the reads map)
    finally:
        m.close()
