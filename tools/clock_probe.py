#!/usr/bin/env python3
"""The shader clock the 64-lane DP kernel sustains (GPU box only):  python tools/clock_probe.py [n_pairs] [len]

Every wavefront of ksw_extd2_wave_kernel<64> stamps s_memtime and s_memrealtime around its DP rows (gdiet_hip_last_dp_clock reduces them;
this tool reads the raw table through a measurement build, build_ab/libgdiet_clk.so = the library with -DGD_CLOCK_STAMP, which exports it).
s_memrealtime ticks at the constant 100 MHz reference (tools/clock_probe.hip checks that against HIP events), so per wavefront
    sclk = (s_memtime ticks / s_memrealtime ticks) x 100 MHz
and, with the instruction counts of the kernel (PMC: profiles/r02_pmc_valu.json), cycles per VALU instruction follow."""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "build_ab", "libgdiet_clk.so")

if "--build" in sys.argv:
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
                           "-DGD_CLOCK_STAMP", "-o", LIB, os.path.join(ROOT, "genome-on-diet_amd", "csrc", "gdiet_hip.hip"), "-lz"])
    print(LIB)
    sys.exit(0)

os.environ["GDIET_HIP_LIB"] = LIB
import numpy as np  # noqa: E402
import torch  # noqa: F401,E402

sys.path.insert(0, ROOT)
from __graft_entry__ import _load_pkg  # noqa: E402

pkg = _load_pkg()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
n = int(args[0]) if args else 9400
ln = int(args[1]) if len(args) > 1 else 15000
w = 1000
rng = np.random.default_rng(1)
qs, ts = [], []
for i in range(n):
    L = int(np.clip(rng.normal(ln, ln * 0.13), ln // 3, ln * 5 // 3))
    t = rng.integers(0, 4, size=L, dtype=np.uint8)
    pos = np.sort(rng.choice(L, size=max(1, L // 300), replace=False))
    q = t.copy()
    q[pos] = (q[pos] + 1) & 3
    q = np.delete(q, pos[::4])
    qs.append(q), ts.append(t)
cells = sum((len(q) + len(t) - 1) * min(w + 1, len(q), len(t)) for q, t in zip(qs, ts))
ctx = pkg.Context(0)
ctx.set_kernel_mode(0)
lib = pkg.load_library()
lib.gdiet_hip_debug_clock_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = {}
for rep in range(3):
    sc, cg = ctx.ksw_extd2_batch(qs, ts, w, pkg.KswScore.from_preset("hifi"))
    dp_ms, _ = ctx.last_kernel_ms()
    buf = (C.c_ulonglong * (4 * n))()
    got = lib.gdiet_hip_debug_clock_stamps(buf, n)
    st = np.frombuffer(buf, np.uint64).reshape(-1, 4)[:got].astype(np.float64)
    mt, rt = st[:, 2] - st[:, 0], st[:, 3] - st[:, 1]
    ok = rt > 0
    f = mt[ok] / rt[ok] * 100.0
    span_ms = (st[ok, 3].max() - st[ok, 1].min()) * 1e-5
    # instructions per cell of this kernel, from the committed PMC summary
    valu_per_cell = None
    for name in ("r03_pmc_valu.json", "r02_pmc_valu.json"):
        p = os.path.join(ROOT, "profiles", name)
        if os.path.exists(p):
            valu_per_cell = json.load(open(p)).get("valu_insts_per_cell")
            break
    out = {"pairs": n, "mean_len": ln, "dp_cells": int(cells), "kernel_ms_events": dp_ms, "kernel_ms_by_s_memrealtime": span_ms, "wavefronts_stamped": int(ok.sum()),
           "sclk_mhz_min": float(f.min()), "sclk_mhz_p10": float(np.percentile(f, 10)), "sclk_mhz_median": float(np.median(f)), "sclk_mhz_p90": float(np.percentile(f, 90)),
           "sclk_mhz_max": float(f.max()), "wavefront_ms_median": float(np.median(rt[ok]) * 1e-5)}
    if valu_per_cell:
        insts_per_simd = valu_per_cell * cells / 1024.0
        out["valu_insts_per_cell_pmc"] = valu_per_cell
        out["ns_per_valu_inst_per_simd"] = dp_ms * 1e6 / insts_per_simd
        out["sclk_cycles_per_valu_inst"] = dp_ms * 1e-3 * out["sclk_mhz_median"] * 1e6 / insts_per_simd
    print(json.dumps(out))
ctx.close()
