"""GPU parity of the SURVEY 8f rank-4 kernels (gdiet_hip_ksw_exts2_batch, gdiet_hip_lchain_dp_batch) against the reference's own
outputs (golden vectors written by oracle/pin_rank4.py) and against the pinned oracle on fresh seeded inputs."""
import os
import sys

import numpy as np
import pytest

from golden_io import SCALARS, load_exts2

pytestmark = pytest.mark.gpu


def _groups(cases):
    """one launch per parameter set: the scalar arguments of ksw_exts2 are per batch"""
    g = {}
    for c in cases:
        key = (bytes(c["mat"]), c["go"], c["ge"], c["go2"], c["noncan"], c["zdrop"], c["junc_bonus"], c["flag"], c["junc"] is not None)
        g.setdefault(key, []).append(c)
    return g.values()


def test_exts2_matches_reference_golden(gpu_ctx):
    n = 0
    for cs in _groups(load_exts2()):
        c0 = cs[0]
        ez, cg = gpu_ctx.ksw_exts2_batch([c["q"] for c in cs], [c["t"] for c in cs], c0["mat"], c0["go"], c0["ge"], c0["go2"], c0["noncan"], c0["zdrop"],
                                         c0["junc_bonus"], c0["flag"], [c["junc"] for c in cs] if c0["junc"] is not None else None)
        for i, c in enumerate(cs):
            for k in SCALARS:
                assert ez[i][k] == c[k], (k, ez[i][k], c[k], hex(c["flag"]), len(c["q"]), len(c["t"]))
            assert np.array_equal(cg[i], c["cigar"]), (hex(c["flag"]), len(c["q"]), len(c["t"]))
            n += 1
    assert n >= 300


def test_exts2_fresh_pairs_against_oracle(gpu_ctx, oracle):
    gdo, lib = oracle
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import pin_rank4
    rng = np.random.default_rng(2024)
    cases = [dict(q=q, t=t, mat=mat, go=go, ge=ge, go2=go2, noncan=nc, zdrop=zd, junc_bonus=jb, flag=flag, junc=junc)
             for q, t, mat, go, ge, go2, nc, zd, jb, flag, junc in pin_rank4.exts2_cases(rng, 600)]
    for cs in _groups(cases):
        c0 = cs[0]
        ez, cg = gpu_ctx.ksw_exts2_batch([c["q"] for c in cs], [c["t"] for c in cs], c0["mat"], c0["go"], c0["ge"], c0["go2"], c0["noncan"], c0["zdrop"],
                                         c0["junc_bonus"], c0["flag"], [c["junc"] for c in cs] if c0["junc"] is not None else None)
        for i, c in enumerate(cs):
            o = gdo.oracle_exts2(lib, c["q"], c["t"], c["mat"], c["go"], c["ge"], c["go2"], c["noncan"], c["zdrop"], c["junc_bonus"], c["flag"], c["junc"])
            for k in SCALARS:
                assert ez[i][k] == o[k], (k, ez[i][k], o[k], hex(c["flag"]))
            assert np.array_equal(cg[i], o["cigar"]), hex(c["flag"])


def test_exts2_refuses_what_the_reference_returns_on(gpu_ctx, pkg):
    q = np.zeros(10, np.uint8)
    mat = np.array([1, -2, -2, -2, 0] * 4 + [0] * 5, np.int8)
    with pytest.raises(pkg.GdietError):
        gpu_ctx.ksw_exts2_batch([q], [q], mat, 4, 2, 6, 5)  # q2 <= q + e (SR/ksw2_exts2_sse.c:72)
    with pytest.raises(pkg.GdietError):
        gpu_ctx.ksw_exts2_batch([q], [q], np.array([1, -40, -40, -40, 0] * 4 + [0] * 5, np.int8), 4, 2, 24, 5)  # -min_sc > 2 (q + e), :90


def test_lchain_matches_reference_golden(gpu_ctx):
    from golden_io import load_lchain
    cases = load_lchain()
    groups = {}
    for c in cases:  # the scalar arguments of mg_lchain_dp are per batch
        groups.setdefault(tuple(sorted(c["par"].items())), []).append(c)
    n_chains = 0
    for cs in groups.values():
        out = gpu_ctx.lchain_dp_batch([c["a"] for c in cs], cs[0]["par"])
        for c, (u, b) in zip(cs, out):
            assert np.array_equal(u, c["u"]) and np.array_equal(b, c["b"]), (len(c["a"]), c["par"])
            n_chains += len(u)
    assert n_chains >= 300


def test_lchain_fresh_reads_against_oracle(gpu_ctx, oracle):
    """one batch of 300 reads with common parameters (max_iter small enough to cut windows, max_skip small enough to leave scans early)"""
    gdo, lib = oracle
    rng = np.random.default_rng(77)
    for par_pick in range(3):
        reads, par = [], None
        for a, p in gdo.lchain_cases(rng, 300):
            par = par or dict(p, max_iter=(50, 5000, 300)[par_pick], max_skip=(5, 25, 2)[par_pick], n_seg=2, is_cdna=par_pick == 2)
            reads.append(a)
        reads.append(np.zeros((0, 2), np.uint64))  # an empty read
        out = gpu_ctx.lchain_dp_batch(reads, par)
        for a, (u, b) in zip(reads, out):
            o = gdo.oracle_lchain(lib, a, par)
            assert np.array_equal(u, o["u"]) and np.array_equal(b, o["a"]), (len(a), par)
