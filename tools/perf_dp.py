#!/usr/bin/env python3
"""quick A/B timing of the DP kernels on synthetic pairs (GPU box only): python tools/perf_dp.py [n] [len] [w] [preset]"""
import os
import sys
import time

import numpy as np
import torch  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from __graft_entry__ import _load_pkg  # noqa: E402
import gdo  # noqa: E402

pkg = _load_pkg()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ln = int(sys.argv[2]) if len(sys.argv) > 2 else 15000
w = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
preset = sys.argv[4] if len(sys.argv) > 4 else "hifi"
modes = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0, 1]
rng = np.random.default_rng(1)
qs, ts = [], []
for i in range(n):
    t = rng.integers(0, 4, size=ln, dtype=np.uint8)
    pos = np.sort(rng.choice(ln, size=max(1, ln // 300), replace=False))
    q = t.copy()
    q[pos] = (q[pos] + 1) & 3
    q = np.delete(q, pos[::4])
    qs.append(q), ts.append(t)
cells = sum((len(q) + len(t) - 1) * min(w + 1, len(q), len(t)) for q, t in zip(qs, ts))
ctx = pkg.Context(0)
print(ctx.device_name)
for mode in modes:
    ctx.set_kernel_mode(mode)
    for rep in range(3):
        t0 = time.time()
        sc, cg = ctx.ksw_extd2_batch(qs, ts, w, pkg.KswScore.from_preset(preset))
        dt = time.time() - t0
        dp, bt = ctx.last_kernel_ms()
        print("mode %d mask %d rep %d: wall %.3fs  dp %.2f ms  backtrack %.2f ms  GCUPS(dp) %.1f  Mbases/s(dp+bt) %.1f  score0 %d"
              % (mode, ctx.last_kernel_mask(), rep, dt, dp, bt, cells / dp / 1e6, sum(map(len, qs)) / (dp + bt) / 1e3, sc[0]))
