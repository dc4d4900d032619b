"""CPU: the oracle (oracle/gdo_ksw2.c) reproduces the reference's outputs stored in tests/golden/."""
import numpy as np

from golden_io import SCALARS, load_exact, load_ksw


def test_extd2_oracle_matches_reference_golden(oracle):
    gdo, lib = oracle
    cases = load_ksw("ksw2_extd2")
    assert len(cases) >= 300
    for c in cases:
        a, b, q, e, q2, e2 = gdo.PRESETS[c["preset"]]
        o = gdo.oracle_extd2(lib, c["q"], c["t"], gdo.score_matrix(a, b), q, e, q2, e2, c["w"], c["zdrop"], c["end_bonus"], c["flag"])
        for k in SCALARS:
            assert o[k] == c[k], (k, o[k], c[k], c["preset"], c["w"], c["flag"])
        assert np.array_equal(o["cigar"], c["cigar"])


def test_extz2_oracle_matches_reference_golden(oracle):
    gdo, lib = oracle
    for c in load_ksw("ksw2_extz2"):
        a, b, q, e, q2, e2 = gdo.PRESETS[c["preset"]]
        o = gdo.oracle_extz2(lib, c["q"], c["t"], gdo.score_matrix(a, b), q, e, c["w"], c["zdrop"], c["end_bonus"], c["flag"])
        for k in SCALARS:
            assert o[k] == c[k], (k, o[k], c[k])
        assert np.array_equal(o["cigar"], c["cigar"])


def test_extz2_exact_mode_oracle_matches_reference_golden(oracle):
    """flag 0 / KSW_EZ_EXTZ_ONLY with and without z-drop (tests/golden/ksw2_extz2_exact.npz): the oracle the exact-maximum GPU
    kernel is fuzzed against reproduces ksw_extz2_sse's scalars and CIGARs"""
    gdo, lib = oracle
    cases = load_ksw("ksw2_extz2_exact")
    assert len(cases) >= 300 and sum(c["zdropped"] for c in cases) >= 20
    for c in cases:
        a, b, q, e, q2, e2 = gdo.PRESETS[c["preset"]]
        o = gdo.oracle_extz2(lib, c["q"], c["t"], gdo.score_matrix(a, b), q, e, c["w"], c["zdrop"], c["end_bonus"], c["flag"])
        for k in SCALARS:
            assert o[k] == c[k], (k, o[k], c[k])
        assert np.array_equal(o["cigar"], c["cigar"])


def test_exact_match_oracle_matches_reference_golden(oracle):
    gdo, lib = oracle
    cases = load_exact()
    assert any(e for _, _, e in cases) and any(not e for _, _, e in cases)
    for q, t, expect in cases:
        assert gdo.oracle_exact_match(lib, q, t) == expect


def test_cigar_consumes_both_sequences(oracle):
    """size-independent property: a reported global alignment consumes exactly qlen and tlen"""
    gdo, lib = oracle
    rng = np.random.default_rng(99)
    for _ in range(60):
        q, t = gdo.make_pair(rng, int(rng.integers(50, 900)), 0.02, 0.01, 0.01)
        a, b, go, ge, go2, ge2 = gdo.PRESETS["hifi"]
        o = gdo.oracle_extd2(lib, q, t, gdo.score_matrix(a, b), go, ge, go2, ge2, 200)
        if o["score"] == gdo.NEG_INF:
            assert len(o["cigar"]) == 0
            continue
        ops, lens = o["cigar"] & 0xf, o["cigar"] >> 4
        assert lens[(ops == 0) | (ops == 1)].sum() == len(q)
        assert lens[(ops == 0) | (ops == 2)].sum() == len(t)


def test_self_alignment_is_all_match(oracle):
    gdo, lib = oracle
    rng = np.random.default_rng(5)
    for n in (1, 15, 16, 17, 150, 1000):
        t = rng.integers(0, 4, size=n, dtype=np.uint8)
        a, b, go, ge, go2, ge2 = gdo.PRESETS["sr"]
        o = gdo.oracle_extd2(lib, t, t, gdo.score_matrix(a, b), go, ge, go2, ge2, 150)
        assert o["score"] == n * a and list(o["cigar"]) == [n << 4]


def test_extz2_equals_extd2_with_equal_gap_models(oracle):
    """the identity K3 rests on (include/gdiet_hip.h): in APPROX_MAX mode ksw_extz2(q,e) and ksw_extd2(q,e,q,e) give the same
    score, CIGAR and band-exhaustion status -- checked on the oracle (pinned to the reference) and, for the golden extz2
    vectors, on ksw_extz2_sse's own outputs"""
    gdo, lib = oracle
    import pin_ksw2
    rng = np.random.default_rng(31)
    n = 0
    for tag, q, t, preset, w, zdrop, eb, flag in pin_ksw2.cases(rng, 900, heavy=False):
        if flag != gdo.EZ_APPROX_MAX or zdrop != -1:
            continue
        a, b, go, ge, _, _ = gdo.PRESETS[preset]
        mat = gdo.score_matrix(a, b)
        z = gdo.oracle_extz2(lib, q, t, mat, go, ge, w)
        d = gdo.oracle_extd2(lib, q, t, mat, go, ge, go, ge, w)
        assert z["score"] == d["score"] and z["zdropped"] == d["zdropped"] and np.array_equal(z["cigar"], d["cigar"]), (tag, len(q), len(t), w)
        n += 1
    assert n > 700
    for c in load_ksw("ksw2_extz2"):
        if c["flag"] != 8 or c["zdrop"] != -1:
            continue
        a, b, go, ge, _, _ = gdo.PRESETS[c["preset"]]
        d = gdo.oracle_extd2(lib, c["q"], c["t"], gdo.score_matrix(a, b), go, ge, go, ge, c["w"])
        assert d["score"] == c["score"] and np.array_equal(d["cigar"], c["cigar"])


def _diag_cases(rng, n_cases, a, b, thr):
    """N-free pairs of equal length whose only differences are m substitutions with m (a + b) <= thr: random, tandem-repeat and two-letter
    targets (where a gapped path re-aligns the most), scattered and adjacent mismatches"""
    mmax = thr // (a + b)
    for it in range(n_cases):
        n = int(rng.integers(17, 260))
        kind = it % 4
        if kind == 0:
            t = rng.integers(0, 4, size=n, dtype=np.uint8)
        elif kind == 1:
            t = np.resize(rng.integers(0, 4, size=int(rng.integers(1, 7)), dtype=np.uint8), n).astype(np.uint8)
        elif kind == 2:
            t = rng.integers(0, 2, size=n, dtype=np.uint8)
        else:
            t = np.resize(rng.integers(0, 4, size=int(rng.integers(2, 9)), dtype=np.uint8), n).astype(np.uint8)
            t[rng.integers(0, n, size=3)] = rng.integers(0, 4, size=3)
        m = int(rng.integers(0, mmax + 1))
        q = t.copy()
        pos = [int(x) for x in rng.choice(n, size=m, replace=False)] if m else []
        if it % 3 == 0 and m >= 2:
            pos[1] = min(n - 1, pos[0] + 1)
        for p in set(pos):
            q[p] = (q[p] + 1 + rng.integers(0, 3)) & 3
        mm = int((q != t).sum())
        if mm * (a + b) <= thr:
            yield q, t, mm


def test_few_mismatches_mean_the_main_diagonal(oracle):
    """what the library's widened pre-filter rests on (ksw_exact_match_kernel, csrc/ksw_backtrack.hip.h): an N-free pair of equal length
    with m substitutions and m (a + b) <= a + 2 (q + e) aligns along its main diagonal -- score (n - m) a - m b, CIGAR "<n>M" -- because any
    other corner-to-corner path pays two gap opens and scores one pair less (at equality a gapped path may tie; the backtrack's priority
    order still walks the diagonal).  Checked on the oracle for the three presets, wide and
    narrow bands, and on the reference's own ksw_extd2_sse where oracle/_ref is built"""
    gdo, lib = oracle
    ref = gdo.load_ref() if gdo.have_ref() else None
    rng = np.random.default_rng(2025)
    n = 0
    for name, (a, b, go, ge, go2, ge2) in gdo.PRESETS.items():
        if go2 + ge2 < go + ge:
            go, ge, go2, ge2 = go2, ge2, go, ge
        mat = gdo.score_matrix(a, b)
        for q, t, mm in _diag_cases(rng, 700, a, b, a + 2 * (go + ge)):
            ln = len(q)
            for w in (ln, ln // 3 + 20):
                o = gdo.oracle_extd2(lib, q, t, mat, go, ge, go2, ge2, w)
                assert o["score"] == (ln - mm) * a - mm * b and list(o["cigar"]) == [ln << 4], (name, ln, mm, w)
                if ref is not None and n % 5 == 0:
                    r = gdo.ref_extd2(ref, q, t, mat, go, ge, go2, ge2, w)
                    assert r["score"] == o["score"] and np.array_equal(r["cigar"], o["cigar"]), (name, ln, mm, w)
                n += 1
    assert n > 3000
