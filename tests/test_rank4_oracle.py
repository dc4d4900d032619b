"""CPU: the oracles of SURVEY 8f rank 4 (ksw_exts2, mg_lchain_dp -- kernels GDiet keeps in its tree and never calls) reproduce the
reference's own outputs stored in tests/golden/ (written by oracle/pin_rank4.py from ksw_exts2_sse / mg_lchain_dp)."""
import numpy as np

from golden_io import SCALARS, load_exts2


def test_exts2_oracle_matches_reference_golden(oracle):
    gdo, lib = oracle
    cases = load_exts2()
    assert len(cases) >= 300
    assert sum(int(((c["cigar"] & 0xf) == 3).sum()) for c in cases) >= 100  # introns (N_SKIP) are really found
    assert sum(c["zdropped"] for c in cases) >= 10 and len({c["flag"] for c in cases}) >= 40
    for c in cases:
        o = gdo.oracle_exts2(lib, c["q"], c["t"], c["mat"], c["go"], c["ge"], c["go2"], c["noncan"], c["zdrop"], c["junc_bonus"], c["flag"], c["junc"])
        for k in SCALARS:
            assert o[k] == c[k], (k, o[k], c[k], hex(c["flag"]))
        assert np.array_equal(o["cigar"], c["cigar"]), hex(c["flag"])


def test_lchain_oracle_matches_reference_golden(oracle):
    from golden_io import load_lchain
    gdo, lib = oracle
    cases = load_lchain()
    assert len(cases) >= 60 and sum(len(c["u"]) for c in cases) >= 300
    for c in cases:
        o = gdo.oracle_lchain(lib, c["a"], c["par"])
        assert np.array_equal(o["u"], c["u"]) and np.array_equal(o["a"], c["b"])


def test_product_chaining_code_on_the_host_matches_reference_golden(tmp_path):
    """the product's pair score (lchain_core.h: the float arithmetic the device kernel compiles) and host stage (lchain_host.h: backtrack,
    compaction, the restated unstable radix sort), driven by a sequential fill (tests/emul/lchain_emul.cpp), against mg_lchain_dp's outputs"""
    import ctypes as C
    import subprocess
    from conftest import ROOT
    from golden_io import load_lchain
    import os
    so = str(tmp_path / "lchain_emul.so")
    subprocess.check_call(["g++", "-O2", "-w", "-shared", "-fPIC", "-ffp-contract=off", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"),
                           os.path.join(ROOT, "tests", "emul", "lchain_emul.cpp"), "-o", so])
    lib = C.CDLL(so)
    u64p = C.POINTER(C.c_uint64)
    lib.lchain_emul.argtypes = [C.c_int64, u64p] + [C.c_int] * 7 + [C.c_float, C.c_float, C.c_int, C.c_int, u64p, u64p, C.POINTER(C.c_int64)]
    for c in load_lchain():
        a, par = np.ascontiguousarray(c["a"]), c["par"]
        n = len(a)
        u, b, n_v = np.zeros(max(n, 1), np.uint64), np.zeros((max(n, 1), 2), np.uint64), C.c_int64(0)
        n_u = lib.lchain_emul(n, a.reshape(-1).ctypes.data_as(u64p) if n else None, par["max_dist_x"], par["max_dist_y"], par["bw"], par["max_skip"], par["max_iter"],
                              par["min_cnt"], par["min_sc"], par["chn_pen_gap"], par["chn_pen_skip"], par["is_cdna"], par["n_seg"],
                              u.ctypes.data_as(u64p), b.reshape(-1).ctypes.data_as(u64p), C.byref(n_v)) if n else 0
        assert n_u == len(c["u"]) and np.array_equal(u[:n_u], c["u"]) and np.array_equal(b[:n_v.value], c["b"])
