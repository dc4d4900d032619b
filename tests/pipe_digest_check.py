"""run by a GPU test in a process of its own (GDIET_SR_PIPE is read once per process): a large seeded batch of short-read-shaped pairs
(150 x 150 at w = 150, 151 x 151 at w = 150, a tail of other lengths; a quarter of them exact matches) through gdiet_hip_ksw_extd2_batch;
prints the kernel mask and a digest over every score and CIGAR.  python pipe_digest_check.py [n_pairs]"""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402  (first: one HIP runtime)
from conftest import load_pkg  # noqa: E402

pkg = load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120000
rng = np.random.default_rng(20251005)
qs, ts, ws = [], [], []
lens = rng.choice([150, 150, 150, 150, 151, 151, 149, 148, 120, 100, 76], size=n)
for i in range(n):
    ln = int(lens[i])
    t = rng.integers(0, 4, size=ln, dtype=np.uint8)
    q = t.copy()
    if i % 4:  # (every fourth pair stays an exact match: the pre-filter's share)
        rate = 0.01 if i % 3 else 0.06
        m = rng.random(ln) < rate
        q[m] = (q[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        if i % 5 == 0:  # a deletion and an insertion: the length stays
            p = int(rng.integers(5, ln - 10))
            q = np.concatenate([q[:p], q[p + 3:], rng.integers(0, 4, size=3, dtype=np.uint8)])
    qs.append(np.ascontiguousarray(q)), ts.append(t), ws.append(150)
ex = np.array([len(q) * 2 for q in qs], np.int32)
ctx = pkg.Context(0)
sc, cg = ctx.ksw_extd2_batch(qs, ts, ws, pkg.KswScore.from_preset("sr"), exact_score=ex)
h = hashlib.sha1()
h.update(np.asarray(sc, np.int32).tobytes())
for c in cg:
    h.update(np.asarray(c, np.uint32).tobytes())
    h.update(b"|")
print("mask", ctx.last_kernel_mask(), "pairs", n, "digest", h.hexdigest())
ctx.close()
