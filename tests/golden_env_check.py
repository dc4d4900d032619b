"""run by GPU tests in a process of its own, with switches of the library in the environment (they are read once per process):
maps the kinds given on the command line and compares every SAM record with the kind's golden SAM.  python golden_env_check.py kind..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (first: one HIP runtime)
from conftest import load_pkg  # noqa: E402
from fixture_io import OVERRIDES, SETS, golden_sam, read_fasta, reads_of  # noqa: E402

pkg = load_pkg()
ctx = pkg.Context(0)
for kind in sys.argv[1:]:
    base, stem, preset = SETS[kind]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
    reads = reads_of(kind)
    m = pkg.Mapper(ctx, names, seqs, preset=preset, **OVERRIDES.get(kind, {}))
    got = m.sam_batch(m.map([r[1] for r in reads]), reads)
    want = "".join(l + "\n" for l in golden_sam(kind))
    if got != want:
        g, w = got.split("\n"), want.split("\n")
        bad = [i for i in range(min(len(g), len(w))) if g[i] != w[i]]
        print("DIFF", kind, len(g), len(w), bad[:3], (g[bad[0]][:300], w[bad[0]][:300]) if bad else "")
        sys.exit(1)
    m.close()
ctx.close()
print("ok")
