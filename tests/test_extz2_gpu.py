"""GPU parity of K3 (gdiet_hip_ksw_extz2_batch) against ksw_extz2_sse's own outputs (golden vectors written by
oracle/pin_ksw2.py from the reference) and against the oracle at BASELINE config-2 shape (150 x 150 pairs)."""
import numpy as np
import pytest

from golden_io import load_ksw

pytestmark = pytest.mark.gpu


def _score(pkg, preset):
    s = pkg.KswScore.from_preset(preset)
    return s


@pytest.mark.parametrize("mode", [1, 0])  # generic LDS kernel only / automatic dispatch (16- and 64-lane kernels)
def test_extz2_matches_reference_golden(gpu_ctx, pkg, mode):
    cases = [c for c in load_ksw("ksw2_extz2") if c["flag"] == 8]
    gpu_ctx.set_kernel_mode(mode)
    try:
        n = 0
        for preset in ("sr", "hifi", "ont"):
            cs = [c for c in cases if c["preset"] == preset]
            sc, cg = gpu_ctx.ksw_extz2_batch([c["q"] for c in cs], [c["t"] for c in cs], [c["w"] for c in cs], _score(pkg, preset))
            for i, c in enumerate(cs):
                assert sc[i] == c["score"], (preset, i, len(c["q"]), len(c["t"]), c["w"], sc[i], c["score"])
                assert np.array_equal(cg[i], c["cigar"]), (preset, i, len(c["q"]), len(c["t"]), c["w"])
                n += 1
        assert n >= 250
    finally:
        gpu_ctx.set_kernel_mode(0)


def test_extz2_config2_shape_against_oracle(gpu_ctx, pkg, oracle):
    """BASELINE configs[1]: 150 bp vs 150 bp pairs (here 20 000 of the 100 000; every 40th pair is checked against the
    oracle, all of them through size-independent properties)"""
    gdo, lib = oracle
    rng = np.random.default_rng(150)
    n = 20000
    T = rng.integers(0, 4, size=(n, 150), dtype=np.uint8)
    Q = T.copy()
    m = rng.random((n, 150)) < 0.02
    Q[m] = (Q[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
    qs, ts = [], []
    for i in range(n):
        q = Q[i]
        if i % 3 == 1:  # a deletion and an insertion: lengths stay 150
            p = int(rng.integers(10, 130))
            q = np.concatenate([q[:p], q[p + 2:], rng.integers(0, 4, size=2, dtype=np.uint8)])
        qs.append(np.ascontiguousarray(q)), ts.append(T[i])
    Q[0] = T[0]
    qs[0] = T[0].copy()
    score = pkg.KswScore.from_preset("sr")
    sc, cg = gpu_ctx.ksw_extz2_batch(qs, ts, 150, score)
    assert gpu_ctx.last_kernel_mask() in (4, 20)  # everything on the short-alignment kernels (bit 4: as skewed pipelines)
    a, b, q_, e, _, _ = gdo.PRESETS["sr"]
    mat = gdo.score_matrix(a, b)
    assert sc[0] == 150 * a and list(cg[0]) == [150 << 4]
    for i in range(n):
        ops, lens = cg[i] & 0xf, cg[i] >> 4
        assert lens[(ops == 0) | (ops == 1)].sum() == 150 and lens[(ops == 0) | (ops == 2)].sum() == 150
    for i in range(0, n, 40):
        o = gdo.oracle_extz2(lib, qs[i], ts[i], mat, q_, e, 150)
        assert sc[i] == o["score"] and np.array_equal(cg[i], o["cigar"]), i


SCALARS = ("score", "zdropped", "max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "reach_end")


def test_extz2_exact_mode_matches_reference_golden(gpu_ctx, pkg):
    """gdiet_hip_ksw_extz2_batch_ex (flag 0 / KSW_EZ_EXTZ_ONLY, z-drop -1 / 20 / 100 / 400, end bonus 0 / 5) against ksw_extz2_sse's own
    outputs: every scalar of ksw_extz_t and the CIGAR (tests/golden/ksw2_extz2_exact.npz, plus the flag-0 / 0x40 rows of ksw2_extz2.npz)"""
    cases = load_ksw("ksw2_extz2_exact") + [c for c in load_ksw("ksw2_extz2") if c["flag"] in (0, 0x40)]
    n, n_drop, n_end = 0, 0, 0
    groups = {}
    for c in cases:
        groups.setdefault((c["preset"], c["flag"], c["zdrop"], c["end_bonus"]), []).append(c)
    for (preset, flag, zdrop, end_bonus), cs in groups.items():
        sc = pkg.KswScore.from_preset(preset)
        sc.flag = flag
        ez, cg = gpu_ctx.ksw_extz2_batch_ex([c["q"] for c in cs], [c["t"] for c in cs], [c["w"] for c in cs], sc, zdrop, end_bonus)
        for i, c in enumerate(cs):
            got = {k: ez[i][k] for k in SCALARS}
            want = {k: c[k] for k in SCALARS}
            assert got == want, (preset, flag, zdrop, end_bonus, len(c["q"]), len(c["t"]), c["w"], {k: (got[k], want[k]) for k in SCALARS if got[k] != want[k]})
            assert np.array_equal(cg[i], c["cigar"]), (preset, flag, zdrop, len(c["q"]), len(c["t"]), c["w"])
            n += 1
            n_drop += c["zdropped"]
            n_end += c["reach_end"]
    assert n >= 300 and n_drop >= 20 and n_end >= 20


def test_extz2_exact_mode_config2_shape_against_oracle(gpu_ctx, pkg, oracle):
    """BASELINE configs[1] with flag 0: 150 x 150 pairs, w = 150; every 25th of 5 000 pairs against the oracle, all of them through the
    identities max >= score, mqe >= score, mte >= score (the corner is one of the cells each maximum ranges over)"""
    gdo, lib = oracle
    rng = np.random.default_rng(151)
    n = 5000
    T = rng.integers(0, 4, size=(n, 150), dtype=np.uint8)
    Q = T.copy()
    m = rng.random((n, 150)) < 0.02
    Q[m] = (Q[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
    qs, ts = [np.ascontiguousarray(Q[i]) for i in range(n)], [T[i] for i in range(n)]
    score = pkg.KswScore.from_preset("sr")
    score.flag = 0
    ez, cg = gpu_ctx.ksw_extz2_batch_ex(qs, ts, 150, score)
    a, b, q_, e, _, _ = gdo.PRESETS["sr"]
    mat = gdo.score_matrix(a, b)
    for i in range(n):
        assert ez[i]["max"] >= ez[i]["score"] and ez[i]["mqe"] >= ez[i]["score"] and ez[i]["mte"] >= ez[i]["score"] and not ez[i]["zdropped"]
    for i in range(0, n, 25):
        o = gdo.oracle_extz2(lib, qs[i], ts[i], mat, q_, e, 150, flag=0)
        assert all(ez[i][k] == o[k] for k in SCALARS) and np.array_equal(cg[i], o["cigar"]), i


def test_extz2_rejects_other_modes(gpu_ctx, pkg):
    s = pkg.KswScore.from_preset("sr")
    s.flag = 0  # exact-max mode has an entry point of its own: the APPROX_MAX one must fail loudly rather than answer approximately
    q = np.zeros(10, np.uint8)
    with pytest.raises(pkg.GdietError):
        gpu_ctx.ksw_extz2_batch([q], [q], 10, s)
    s.flag = 2  # KSW_EZ_RIGHT: not implemented in either
    with pytest.raises(pkg.GdietError):
        gpu_ctx.ksw_extz2_batch_ex([q], [q], 10, s)
