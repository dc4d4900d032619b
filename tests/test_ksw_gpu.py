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
