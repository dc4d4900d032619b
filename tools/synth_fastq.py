"""write a synthetic reference (FASTA) and short reads drawn from it (four-line FASTQ) for tools/map_file.py:
    python tools/synth_fastq.py <dir> [ref_mbp=100] [n_reads=4000000] [sr|hifi] [noref]
(noref: only reads.fq is written; tools/map_file.py then takes "synth:<ref_mbp>" as its reference and makes the same sequences in-process)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tools.bench_variant import synth_reads  # noqa: E402


def main():
    d = sys.argv[1]
    ref_mbp = float(sys.argv[2]) if len(sys.argv) > 2 else 100
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 4_000_000
    os.makedirs(d, exist_ok=True)
    names, contigs = bench.synth_reference(ref_mbp, seed=2)
    if "noref" not in sys.argv[5:]:
        with open(os.path.join(d, "ref.fa"), "wb") as f:
            for nm, c in zip(names, contigs):
                f.write(b">" + nm.encode() + b"\n" + c.tobytes() + b"\n")
    kind = sys.argv[4] if len(sys.argv) > 4 else "sr"
    if kind == "hifi":
        base = min(n, 20480)
        reads = [s for _, s in bench.synth_hifi_reads(np.random.default_rng(5), contigs, base)]
        with open(os.path.join(d, "reads.fq"), "wb") as f:
            for rep in range(max(1, n // base)):
                for i, s in enumerate(reads):
                    f.write(b"@h%d_%d\n" % (rep, i) + s + b"\n+\n" + b"I" * len(s) + b"\n")
        return
    base = min(n, 500_000)
    reads = synth_reads(np.random.default_rng(7), contigs, base, "sr")
    q = b"I" * 200
    with open(os.path.join(d, "reads.fq"), "wb") as f:
        for rep in range(max(1, n // base)):
            f.write(b"".join(b"@r%d_%d\n" % (rep, i) + (s if isinstance(s, bytes) else s.encode()) + b"\n+\n" + q[:len(s)] + b"\n" for i, s in enumerate(reads)))


if __name__ == "__main__":
    main()
