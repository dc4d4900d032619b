"""GPU parity: the HIP ksw_extd2 batch (through the C ABI) against the reference's golden vectors and against the
oracle on seeded fuzz; plus size-independent properties at HiFi-sized inputs."""
import numpy as np
import pytest

from golden_io import load_exact, load_ksw

pytestmark = pytest.mark.gpu


def _run(ctx, pkg, cases_by_preset):
    out = {}
    for preset, cases in cases_by_preset.items():
        sc, cg = ctx.ksw_extd2_batch([c["q"] for c in cases], [c["t"] for c in cases], [c["w"] for c in cases],
                                     pkg.KswScore.from_preset(preset))
        out[preset] = (sc, cg)
    return out


@pytest.mark.parametrize("mode", [1, 0])  # 1 = generic LDS kernel only, 0 = automatic dispatch (wave kernels)
def test_extd2_matches_reference_golden(gpu_ctx, pkg, mode):
    gpu_ctx.set_kernel_mode(mode)
    cases = [c for c in load_ksw("ksw2_extd2") if c["flag"] == 8]
    by = {}
    for c in cases:
        by.setdefault(c["preset"], []).append(c)
    res = _run(gpu_ctx, pkg, by)
    gpu_ctx.set_kernel_mode(0)
    n = 0
    for preset, cs in by.items():
        sc, cg = res[preset]
        for i, c in enumerate(cs):
            assert sc[i] == c["score"], (preset, i, len(c["q"]), len(c["t"]), c["w"], sc[i], c["score"])
            assert np.array_equal(cg[i], c["cigar"]), (preset, i, len(c["q"]), len(c["t"]), c["w"])
            n += 1
    assert n >= 250


@pytest.mark.parametrize("mode", [1, 0])
def test_extd2_matches_oracle_fuzz(gpu_ctx, pkg, oracle, mode):
    gdo, lib = oracle
    import pin_ksw2
    rng = np.random.default_rng(2024 + mode)
    cases = [c for c in pin_ksw2.cases(rng, 480, heavy=False) if c[7] == 8]
    gpu_ctx.set_kernel_mode(mode)
    try:
        for preset in ("sr", "hifi", "ont"):
            cs = [c for c in cases if c[3] == preset]
            sc, cg = gpu_ctx.ksw_extd2_batch([c[1] for c in cs], [c[2] for c in cs], [c[4] for c in cs], pkg.KswScore.from_preset(preset))
            a, b, q, e, q2, e2 = gdo.PRESETS[preset]
            mat = gdo.score_matrix(a, b)
            for i, c in enumerate(cs):
                o = gdo.oracle_extd2(lib, c[1], c[2], mat, q, e, q2, e2, c[4])
                assert sc[i] == o["score"], (preset, c[0], len(c[1]), len(c[2]), c[4], sc[i], o["score"])
                assert np.array_equal(cg[i], o["cigar"]), (preset, c[0], len(c[1]), len(c[2]), c[4])
    finally:
        gpu_ctx.set_kernel_mode(0)


def test_exact_match_prefilter(gpu_ctx, pkg, oracle):
    gdo, lib = oracle
    cases = load_exact()
    qs, ts = [c[0] for c in cases], [c[1] for c in cases]
    ex = np.full(len(cases), 777, np.int32)
    sc, cg = gpu_ctx.ksw_extd2_batch(qs, ts, 150, pkg.KswScore.from_preset("sr"), exact_score=ex)
    a, b, q, e, q2, e2 = gdo.PRESETS["sr"]
    for i, (qq, tt, expect) in enumerate(cases):
        if expect:
            assert sc[i] == 777 and list(cg[i]) == [len(qq) << 4]
        else:
            o = gdo.oracle_extd2(lib, qq, tt, gdo.score_matrix(a, b), q, e, q2, e2, 150)
            assert sc[i] == o["score"] and np.array_equal(cg[i], o["cigar"])


def test_hifi_sized_properties_and_oracle_spotcheck(gpu_ctx, pkg, oracle):
    """BASELINE config 4 geometry (15 kbp, w = 1000): CIGAR consumes both sequences; identical sequences give nM;
    a few pairs are checked against the oracle (it needs ~0.3 s per pair at this size)."""
    gdo, lib = oracle
    rng = np.random.default_rng(5)
    qs, ts = [], []
    for i in range(24):
        n = int(np.clip(rng.normal(15000, 2000), 5000, 25000))
        t = rng.integers(0, 4, size=n, dtype=np.uint8)
        q = t.copy() if i == 0 else gdo.mutate(rng, t, 0.002, 0.001, 0.001)
        qs.append(q), ts.append(t)
    sc, cg = gpu_ctx.ksw_extd2_batch(qs, ts, 1000, pkg.KswScore.from_preset("hifi"))
    assert sc[0] == len(qs[0]) and list(cg[0]) == [len(qs[0]) << 4]
    for i in range(len(qs)):
        ops, lens = cg[i] & 0xf, cg[i] >> 4
        assert lens[(ops == 0) | (ops == 1)].sum() == len(qs[i])
        assert lens[(ops == 0) | (ops == 2)].sum() == len(ts[i])
    a, b, q, e, q2, e2 = gdo.PRESETS["hifi"]
    for i in (1, 2, 3):
        o = gdo.oracle_extd2(lib, qs[i], ts[i], gdo.score_matrix(a, b), q, e, q2, e2, 1000)
        assert sc[i] == o["score"] and np.array_equal(cg[i], o["cigar"])


def test_band_exhaustion_reports_neg_inf(gpu_ctx, pkg):
    rng = np.random.default_rng(1)
    t = rng.integers(0, 4, size=400, dtype=np.uint8)
    q = t[:100].copy()
    sc, cg = gpu_ctx.ksw_extd2_batch([q], [t], 20, pkg.KswScore.from_preset("sr"))
    assert sc[0] == pkg.hip_abi.NEG_INF and len(cg[0]) == 0


def test_short_read_quartets_match_oracle(gpu_ctx, pkg, oracle):
    """the 16-lane kernel (four alignments of identical geometry per wavefront): a block of 150 x 150, w = 150 pairs (the
    short-read shape: SR/map.c:925 passes len, len, bw), incomplete quartets, mixed geometries, Ns, exact-match rows inside a
    quartet -- all against the oracle"""
    gdo, lib = oracle
    rng = np.random.default_rng(77)
    qs, ts, ws = [], [], []
    for i in range(403):  # 403 = 100 quartets + 3: the last quartet is padded
        q, t = gdo.make_pair(rng, 150, 0.02, 0.004, 0.004, n_frac=0.01 if i % 7 == 0 else 0.0)
        n = min(len(q), len(t), 150)
        if i % 5:
            q, t = q[:n], t[:n]  # len x len as the short-read path calls it; every 5th keeps qlen != tlen
        if i % 11 == 0:
            q = t.copy()  # exact match row: answered by the pre-filter, its quartet row must stay silent
        qs.append(q), ts.append(t), ws.append(150)
    for i in range(120):  # other geometries: singles and pairs end up in padded quartets
        n = int(rng.integers(40, 230))
        q, t = gdo.make_pair(rng, n, 0.03, 0.01, 0.01)
        qs.append(q), ts.append(t), ws.append(int(rng.integers(30, 200)))
    ex = np.array([len(q) * 2 if len(q) == len(t) else pkg.hip_abi.NEG_INF for q, t in zip(qs, ts)], np.int32)
    sc, cg = gpu_ctx.ksw_extd2_batch(qs, ts, ws, pkg.KswScore.from_preset("sr"), exact_score=ex)
    assert gpu_ctx.last_kernel_mask() & 4
    a, b, q_, e, q2, e2 = gdo.PRESETS["sr"]
    mat = gdo.score_matrix(a, b)
    n_exact = 0
    for i in range(len(qs)):
        if len(qs[i]) == len(ts[i]) and np.array_equal(qs[i], ts[i]):
            assert sc[i] == ex[i] and list(cg[i]) == [len(qs[i]) << 4]
            n_exact += 1
            continue
        o = gdo.oracle_extd2(lib, qs[i], ts[i], mat, q_, e, q2, e2, ws[i])
        assert sc[i] == o["score"], (i, len(qs[i]), len(ts[i]), ws[i], sc[i], o["score"])
        assert np.array_equal(cg[i], o["cigar"]), (i, len(qs[i]), len(ts[i]), ws[i])
    assert n_exact >= 30
