#!/usr/bin/env python3
"""BASELINE configs[1]: ksw2_extz2 kernel only, 100 000 synthetic 150 bp x 150 bp pairs on one MI355X (kernel time from HIP events).
    python tools/bench_extz2.py [pairs=100000] [d]       (d: the same pairs through ksw_extd2, the dual-affine form the ShortReads path calls)"""
import json
import os
import sys

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg  # noqa: E402

pkg = _load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(150)
T = rng.integers(0, 4, size=(n, 150), dtype=np.uint8)
Q = T.copy()
m = rng.random((n, 150)) < 0.02
Q[m] = (Q[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
qs, ts = list(Q), list(T)
ctx = pkg.Context(0)
score = pkg.KswScore.from_preset("sr")
dual = len(sys.argv) > 2 and sys.argv[2] == "d"
for rep in range(3):
    sc, cg = ctx.ksw_extd2_batch(qs, ts, 150, score) if dual else ctx.ksw_extz2_batch(qs, ts, 150, score)
dp, bt = ctx.last_kernel_ms()
cells, alg = ctx.last_dp_work()
print(json.dumps({"config": "BASELINE configs[1]" + (" pairs through ksw_extd2" if dual else ""), "pairs": n, "dp_ms": dp, "backtrack_ms": bt, "pairs_per_s": n / ((dp + bt) * 1e-3), "gcups": cells / dp / 1e6,
                  "algorithmic_GBps": alg / dp / 1e6, "kernel_mask": ctx.last_kernel_mask(), "mean_score": float(np.mean(sc))}))
