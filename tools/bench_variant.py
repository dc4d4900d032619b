#!/usr/bin/env python3
"""Whole-path throughput of the other two canonical configurations (BASELINE configs[2] ShortReads, configs[4] ONT) on one GPU.
Not the headline bench (bench.py = configs[3]); same structure, smaller synthetic reference by default.

    python tools/bench_variant.py --kind sr  [--batch 262144] [--inflight 8] [--ref-mbp 3088] [--steps 256]
      (a ShortReads batch is a chain of short kernels and host stages: eight batches in flight keep the GPU busy, 32.8 M reads/s; four: 22.6 M)
    python tools/bench_variant.py --kind ont [--batch 12288] [--inflight 3] [--ref-mbp 3088] [--steps 4]
      (12288 reads per batch = three rounds of the 4096 resident wavefronts of the checkpointed wide-band kernel (96-block ring, 4 per SIMD): the long tail of the
      read-length distribution -- a 150 kbp read runs three times as long as the median one -- is then hidden behind the refill)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (first: one HIP runtime for torch and libgdiet_hip.so)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from __graft_entry__ import _load_pkg  # noqa: E402


def synth_reads(rng, contigs, n, kind):
    lens = np.array([len(c) for c in contigs], np.float64)
    cs = rng.choice(len(contigs), size=n, p=lens / lens.sum())
    out = []
    for i in range(n):
        c = int(cs[i])
        if kind == "sr":
            ln, sub, ind = 150, 0.01, 0.0005
        else:
            ln, sub, ind = int(np.clip(rng.lognormal(np.log(50000), 0.35), 5000, 150000)), 0.03, 0.02
        ln = min(ln, len(contigs[c]) - 2)
        st = int(rng.integers(0, len(contigs[c]) - ln))
        s = contigs[c][st:st + ln].copy()
        m = np.flatnonzero((rng.random(ln) < sub) & (s != 78))
        if len(m):
            s[m] = bench.BASES[(np.searchsorted(bench.BASES, s[m]) + rng.integers(1, 4, size=len(m))) & 3]
        if kind != "sr" or rng.random() < 0.15:
            s = s[~(rng.random(len(s)) < ind)]
            ip = np.flatnonzero(rng.random(len(s)) < ind)
            s = np.insert(s, ip, bench.BASES[rng.integers(0, 4, size=len(ip))])
        if rng.random() < 0.5:
            s = bench.COMP[s[::-1]]
        out.append(s.tobytes())
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", choices=["sr", "ont"], required=True)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--ref-mbp", type=float, default=400)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--inflight", type=int, default=1)
    ap.add_argument("--host-threads", type=int, default=0)
    a = ap.parse_args()
    n = a.batch or (262144 if a.kind == "sr" else 12288)
    pkg = _load_pkg()
    ctx = pkg.Context(0)
    names, contigs = bench.synth_reference(a.ref_mbp, seed=2)
    t0 = time.time()
    m = pkg.Mapper(ctx, names, contigs, preset=a.kind, n_threads=pkg.effective_cpus())
    t_idx = time.time() - t0
    m.set_host_threads(a.host_threads or pkg.effective_cpus())
    reads = synth_reads(np.random.default_rng(7), contigs, n, a.kind)
    batch = m.upload(reads)
    bases = sum(len(r) for r in reads)
    res = m.map_uploaded(batch)  # warm-up
    if a.inflight > 1:
        m.set_inflight(a.inflight)
        for t in [m.submit(batch) for _ in range(a.inflight)]:  # every lane allocates its scratch once
            res = m.wait(t)
    st, open_t, main_t, t0 = [], [], [], time.perf_counter()
    for _ in range(a.steps):
        if a.inflight == 1:
            res = m.map_uploaded(batch)
            st.append(m.stage_seconds())
        else:
            ta = time.perf_counter()
            open_t.append(m.submit(batch))
            tb = time.perf_counter()
            if len(open_t) == a.inflight:
                res = None  # (frees the previous step's records)
                tc = time.perf_counter()
                res = m.wait(open_t.pop(0))
                td = time.perf_counter()
                st.append(m.stage_seconds())
                main_t.append((tb - ta, tc - tb, td - tc))
    while open_t:
        res = m.wait(open_t.pop(0))
        st.append(m.stage_seconds())
    dt = (time.perf_counter() - t0) / a.steps
    mapped = sum(len(reads[i]) for i in range(n) if res.n_regs[i] > 0)
    dp, bt = ctx.last_kernel_ms()
    cells, alg = ctx.last_dp_work()
    s = np.mean(np.array(st), axis=0)
    print(json.dumps({"kind": a.kind, "reads_per_step": n, "bases_per_step": bases, "mapped_fraction": mapped / bases, "ms_per_step": 1e3 * dt,
                      "mapped_Mbases_per_s": mapped / dt / 1e6, "reads_per_s": n / dt, "index_build_s": round(t_idx, 1), "ref_mbp": a.ref_mbp,
                      "stage_ms": {"seed": 1e3 * s[0], "vote": 1e3 * s[1], "host_geometry": 1e3 * s[2], "gather_dp_backtrack": 1e3 * s[3],
                                   "host_post": 1e3 * s[4], "other": 1e3 * s[5]},
                      "caller_ms": dict(zip(("submit", "free_previous", "wait"), (1e3 * np.mean(np.array(main_t), axis=0)).round(2).tolist())) if main_t else None,
                      "dp_kernel_ms": dp, "backtrack_ms": bt, "dp_gcups": cells / dp / 1e6 if dp else None, "kernel_mask": ctx.last_kernel_mask()}))


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        # (a failed run -- e.g. a batch too large for the device -- leaves the context's worker threads alive: print the error and leave
        # without running the interpreter's and the library's exit handlers beside them)
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
