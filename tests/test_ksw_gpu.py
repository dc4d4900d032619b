"""GPU parity: the HIP ksw_extd2 batch (through the C ABI) against the reference's golden vectors and against the
oracle on seeded fuzz; plus size-independent properties at HiFi-sized inputs."""
import os

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
    """the short-alignment kernels (several alignments of identical geometry per wavefront; 150 x 150 runs six to a wavefront in groups
    of ten lanes): a block of 150 x 150, w = 150 pairs (the
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


def test_short_read_pipelines_match_oracle(gpu_ctx, pkg, oracle):
    """full matrices of one geometry as skewed pipelines (ksw_extd2_pipe_kernel: a lane starts the group's next alignment as soon as its
    block has left the matrix): 2 to 16 lanes per alignment, qlen on both sides of tlen, runs that end inside a wavefront's share,
    exact-match rows (answered by the pre-filter: their slot of the pipe stays silent), Ns, and heavy error rates whose best paths
    use the whole matrix -- against the oracle; the other geometries of the batch stay on the grouped kernels"""
    gdo, lib = oracle
    rng = np.random.default_rng(515)
    qs, ts, ws = [], [], []
    geos = [(150, 150, 150, 333), (150, 150, 171, 40), (17, 17, 20, 70), (36, 49, 49, 77), (128, 114, 128, 90), (100, 100, 150, 64), (160, 160, 160, 29),
            (155, 145, 200, 50), (59, 66, 72, 130), (33, 48, 48, 25), (150, 151, 151, 13),
            (151, 151, 150, 100),  # a 151-base read at bw = 150: the narrowest band that never binds
            (250, 250, 250, 50), (241, 256, 256, 9), (245, 245, 244, 1),  # 16 blocks: beyond the grouped kernels -- pipelines however short the run
            (200, 200, 199, 30)]
    for ql, tl, w, cnt in geos:
        for i in range(cnt):
            err = (0.01, 0.002) if i % 3 == 0 else (0.05, 0.02) if i % 3 == 1 else (0.15, 0.06)
            q, t = gdo.make_pair(rng, max(ql, tl) + 40, err[0], err[1], err[1], n_frac=0.02 if i % 9 == 4 else 0.0)
            q, t = q[:ql], t[:tl]
            assert len(q) == ql and len(t) == tl
            if i % 13 == 5 and ql == tl:
                q = t.copy()
            qs.append(np.ascontiguousarray(q)), ts.append(np.ascontiguousarray(t)), ws.append(w)
    for i in range(60):  # narrow bands and lone geometries: not for the pipelines
        n = int(rng.integers(40, 200))
        q, t = gdo.make_pair(rng, n, 0.03, 0.01, 0.01)
        qs.append(q), ts.append(t), ws.append(int(rng.integers(20, 60)))
    order = rng.permutation(len(qs))
    qs, ts, ws = [qs[i] for i in order], [ts[i] for i in order], [ws[i] for i in order]
    ex = np.array([len(q) * 2 if len(q) == len(t) else pkg.hip_abi.NEG_INF for q, t in zip(qs, ts)], np.int32)
    sc, cg = gpu_ctx.ksw_extd2_batch(qs, ts, ws, pkg.KswScore.from_preset("sr"), exact_score=ex)
    assert gpu_ctx.last_kernel_mask() & 16
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
    assert n_exact >= 20


def test_pipeline_runs_the_prefilter_empties(gpu_ctx, pkg, oracle):
    """a run of one geometry in which the exact-match pre-filter answers EVERY alignment (the device-side compaction leaves its pipes
    empty: wavefronts with nothing to do) beside a run it leaves alone, and a run with a single pending alignment"""
    gdo, lib = oracle
    rng = np.random.default_rng(99)
    qs, ts, ws = [], [], []
    for i in range(80):  # 90 x 90: all exact
        t = rng.integers(0, 4, size=90, dtype=np.uint8)
        qs.append(t.copy()), ts.append(t), ws.append(90)
    for i in range(80):  # 110 x 110: one pending alignment among exact ones
        t = rng.integers(0, 4, size=110, dtype=np.uint8)
        q = t.copy()
        if i == 37:
            q[50] = (q[50] + 1) & 3
            q = np.concatenate([q[:20], q[22:], rng.integers(0, 4, size=2, dtype=np.uint8)])
        qs.append(np.ascontiguousarray(q)), ts.append(t), ws.append(110)
    for i in range(60):  # 130 x 130: nothing exact
        q, t = gdo.make_pair(rng, 170, 0.04, 0.01, 0.01)
        qs.append(np.ascontiguousarray(q[:130])), ts.append(np.ascontiguousarray(t[:130])), ws.append(130)
    ex = np.array([len(q) * 2 for q in qs], np.int32)
    sc, cg = gpu_ctx.ksw_extd2_batch(qs, ts, ws, pkg.KswScore.from_preset("sr"), exact_score=ex)
    assert gpu_ctx.last_kernel_mask() & 16
    a, b, q_, e, q2, e2 = gdo.PRESETS["sr"]
    mat = gdo.score_matrix(a, b)
    for i in range(len(qs)):
        if np.array_equal(qs[i], ts[i]):
            assert sc[i] == ex[i] and list(cg[i]) == [len(qs[i]) << 4], i
            continue
        o = gdo.oracle_extd2(lib, qs[i], ts[i], mat, q_, e, q2, e2, ws[i])
        assert sc[i] == o["score"] and np.array_equal(cg[i], o["cigar"]), (i, len(qs[i]), sc[i], o["score"])


def test_pipelines_and_grouped_kernels_agree_at_scale():
    """120 000 short-read-shaped pairs (a short-read batch's size per GPU wavefront slot; a quarter exact matches, which the device-side
    compaction drops from the pipes) through the skewed pipelines, in a second process with GDIET_SR_PIPE=0 through the grouped
    kernels, and in a third with GDIET_DIAG_SHORTCUT=0 (no alignment answered from its main diagonal's score: every one through DP and walk):
    one digest over every score and CIGAR (tests/pipe_digest_check.py).  The grouped kernels are pinned to the oracle and the
    reference's goldens by the tests around this one; this is the same comparison at a size the oracle would need minutes for."""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pipe_digest_check.py")
    outs = []
    for env in ({}, {"GDIET_SR_PIPE": "0"}, {"GDIET_DIAG_SHORTCUT": "0"}):  # (the last: pipelines, but every alignment through the DP and the walk)
        r = subprocess.run([sys.executable, script], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        assert r.returncode == 0 and "digest" in r.stdout, r.stdout[-1000:] + r.stderr[-3000:]
        outs.append(r.stdout.strip().split("\n")[-1].split())
    assert int(outs[0][1]) & 16 and not int(outs[1][1]) & 16 and int(outs[2][1]) & 16, outs  # the second run did without the pipelines
    assert outs[0][-1] == outs[1][-1] == outs[2][-1], outs


def test_short_read_group_widths_match_oracle(gpu_ctx, pkg, oracle):
    """the short-alignment kernels pack 8 / 6 / 4 alignments of one geometry into a wavefront (groups of 8 / 10 / 16 lanes for
    targets of <= 128 / <= 160 / more bases): targets on both sides of each limit, full and padded groups, narrow and full bands,
    qlen != tlen, Ns -- against the oracle; and GDIET_GROUP_LANES-independent (the same pairs through the generic kernel)"""
    gdo, lib = oracle
    rng = np.random.default_rng(4242)
    qs, ts, ws = [], [], []
    for tlen in (33, 112, 127, 128, 129, 144, 159, 160, 161, 176):
        for w in (24, 150):
            base_q, base_t = gdo.make_pair(rng, tlen + 8, 0.0, 0.0, 0.0)
            for i in range(19):  # 19 = two full groups of 8 + 3, three of 6 + 1, four of 4 + 3
                q, t = gdo.make_pair(rng, tlen + 8, 0.03, 0.006, 0.006, n_frac=0.02 if i == 4 else 0.0)
                t = t[:tlen] if len(t) >= tlen else np.concatenate([t, base_t[:tlen - len(t)]])
                ql = tlen - (i % 3)  # a few query lengths per target length: several geometries per width
                q = q[:ql] if len(q) >= ql else np.concatenate([q, base_q[:ql - len(q)]])
                qs.append(q), ts.append(t), ws.append(w)
    sc, cg = gpu_ctx.ksw_extd2_batch(qs, ts, ws, pkg.KswScore.from_preset("sr"))
    assert gpu_ctx.last_kernel_mask() & 4
    gpu_ctx.set_kernel_mode(1)
    try:
        sc_g, cg_g = gpu_ctx.ksw_extd2_batch(qs, ts, ws, pkg.KswScore.from_preset("sr"))
    finally:
        gpu_ctx.set_kernel_mode(0)
    a, b, q_, e, q2, e2 = gdo.PRESETS["sr"]
    mat = gdo.score_matrix(a, b)
    for i in range(len(qs)):
        o = gdo.oracle_extd2(lib, qs[i], ts[i], mat, q_, e, q2, e2, ws[i])
        assert sc[i] == o["score"] == sc_g[i], (i, len(qs[i]), len(ts[i]), ws[i], sc[i], o["score"], sc_g[i])
        assert np.array_equal(cg[i], o["cigar"]) and np.array_equal(cg_g[i], o["cigar"]), (i, len(qs[i]), len(ts[i]), ws[i])


def test_device_pointer_entry_matches_host_entry(gpu_ctx, pkg, oracle):
    """gdiet_hip_ksw_extd2_batch_dev: everything resident in HBM (torch tensors), launched on a torch stream, never synchronises"""
    import torch
    gdo, _ = oracle
    rng = np.random.default_rng(9)
    qs, ts = [], []
    for i in range(40):
        q, t = gdo.make_pair(rng, int(rng.integers(100, 3000)), 0.01, 0.003, 0.003)
        qs.append(q), ts.append(t)
    w = np.full(len(qs), 300, np.int32)
    score = pkg.KswScore.from_preset("hifi")
    want_sc, want_cg = gpu_ctx.ksw_extd2_batch(qs, ts, 300, score)
    qbuf, qoff = pkg.pack(qs)
    tbuf, toff = pkg.pack(ts)
    coff = np.zeros(len(qs) + 1, np.int64)
    coff[1:] = np.cumsum([len(q) + len(t) for q, t in zip(qs, ts)])
    dev = torch.device("cuda", 0)
    d_q, d_t = torch.from_numpy(qbuf).to(dev), torch.from_numpy(tbuf).to(dev)
    d_coff = torch.from_numpy(coff).to(dev)
    d_sc = torch.zeros(len(qs), dtype=torch.int32, device=dev)
    d_nc = torch.zeros(len(qs), dtype=torch.int32, device=dev)
    d_cg = torch.zeros(int(coff[-1]) + 1, dtype=torch.int32, device=dev)
    st = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(st):
        gpu_ctx.ksw_extd2_batch_dev(len(qs), d_q.data_ptr(), d_t.data_ptr(), None, score, d_sc.data_ptr(), d_nc.data_ptr(), d_cg.data_ptr(),
                                    d_coff.data_ptr(), qoff, toff, w, stream=st.cuda_stream)
    st.synchronize()
    sc, nc, cg = d_sc.cpu().numpy(), d_nc.cpu().numpy(), d_cg.cpu().numpy().view(np.uint32)
    for i in range(len(qs)):
        assert sc[i] == want_sc[i] and np.array_equal(cg[coff[i]:coff[i] + nc[i]], want_cg[i])
    dp_ms, bt_ms = gpu_ctx.last_kernel_ms()
    assert dp_ms > 0 and bt_ms > 0
    cells, alg = gpu_ctx.last_dp_work()
    assert cells == sum((len(q) + len(t) - 1) * min(301, len(q), len(t)) for q, t in zip(qs, ts)) and alg > cells


def test_error_behaviour_of_the_abi(gpu_ctx, pkg):
    """argument errors are reported, never answered approximately (no CPU fallback, no silent truncation)"""
    import ctypes as C
    lib = pkg.load_library()
    score = pkg.KswScore.from_preset("sr")
    q = np.zeros(50, np.uint8)
    assert gpu_ctx.ksw_extd2_batch([], [], 10, score)[0].size == 0  # n = 0 is fine
    with pytest.raises(pkg.GdietError):  # an empty sequence: the reference returns without aligning
        gpu_ctx.ksw_extd2_batch([q, np.zeros(0, np.uint8)], [q, q], 10, score)
    bad = pkg.KswScore.from_preset("sr")
    bad.flag = 0x40  # KSW_EZ_EXTZ_ONLY: not a mode of the live path
    with pytest.raises(pkg.GdietError):
        gpu_ctx.ksw_extd2_batch([q], [q], 10, bad)
    bad = pkg.KswScore.from_preset("sr")
    bad.q, bad.e = 1, 1
    bad.mismatch = -100  # -min_sc > 2(q+e): ksw_extd2 returns without aligning
    with pytest.raises(pkg.GdietError):
        gpu_ctx.ksw_extd2_batch([q], [q], 10, bad)
    # CIGAR capacity too small: error code -5 and the needed size in n_cigar
    rng = np.random.default_rng(2)
    t = rng.integers(0, 4, size=200, dtype=np.uint8)
    qq = np.concatenate([t[:80], t[90:150], rng.integers(0, 4, size=7, dtype=np.uint8), t[150:]])
    qoff, toff = np.array([0, len(qq)], np.int64), np.array([0, len(t)], np.int64)
    w, sc, nc = np.array([100], np.int32), np.zeros(1, np.int32), np.zeros(1, np.int32)
    cg, coff = np.zeros(4, np.uint32), np.array([0, 2], np.int64)
    p = lambda a, ty: a.ctypes.data_as(C.POINTER(ty))
    rc = lib.gdiet_hip_ksw_extd2_batch(gpu_ctx._h, 1, p(qq, C.c_uint8), p(qoff, C.c_int64), p(t, C.c_uint8), p(toff, C.c_int64), p(w, C.c_int32),
                                       None, C.byref(score), p(sc, C.c_int32), p(nc, C.c_int32), p(cg, C.c_uint32), p(coff, C.c_int64))
    assert rc == -5 and nc[0] > 2
    assert lib.gdiet_hip_ksw_extd2_batch(None, 1, None, None, None, None, None, None, None, None, None, None, None) == -3
    ctx2 = C.c_void_p()
    assert lib.gdiet_hip_init(C.byref(ctx2), 99) == -1  # no such device: GDIET_E_NODEVICE


@pytest.mark.parametrize("two_waves", ["0", "1", "ckpt"])
def test_wide_band_kernels_match_oracle(pkg, oracle, monkeypatch, two_waves):
    """ONT bands (w = 1300: more than 64 blocks in flight) on the wide-band kernels -- two blocks per lane in one wavefront,
    two wavefronts per alignment exchanging their boundary through LDS, and the checkpointed form of the first (no stored
    backtrace: snapshots every 480 anti-diagonals, the cone of the walk recomputed chunk by chunk -- one half block per lane behind the
    96-block-ring first pass, one block per lane for bands wider than 95 blocks; alignments of 1 to 20 chunks here) -- against the oracle"""
    gdo, lib = oracle
    if two_waves == "ckpt":
        monkeypatch.setenv("GDIET_WIDE_CKPT", "1")
    else:
        monkeypatch.setenv("GDIET_WIDE_TWO_WAVES", two_waves)
        monkeypatch.setenv("GDIET_WIDE_CKPT", "0")
    ctx = pkg.Context(0)
    try:
        rng = np.random.default_rng(40 + (2 if two_waves == "ckpt" else int(two_waves)))
        qs, ts, ws = [], [], []
        for i in range(24):
            n = int(rng.integers(1400, 5000))
            if two_waves == "ckpt" and i % 6 == 0:
                n = [240, 480, 481, 720][i // 6 % 4]  # rend + 1 around multiples of the chunk size (480 anti-diagonals)
            q, t = gdo.make_pair(rng, n, 0.03, 0.02, 0.02, n_frac=0.01 if i % 5 == 0 else 0.0)
            if i % 4 == 1:
                q = q.copy()
                q[q == 4] = 7  # N of a reverse-complemented read
            qs.append(q), ts.append(t), ws.append(1300 if i % 3 else int(rng.integers(1050, 1900)))
        sc, cg = ctx.ksw_extd2_batch(qs, ts, ws, pkg.KswScore.from_preset("ont"))
        assert ctx.last_kernel_mask() & 8
        a, b, q_, e, q2, e2 = gdo.PRESETS["ont"]
        mat = gdo.score_matrix(a, b)
        for i in range(len(qs)):
            o = gdo.oracle_extd2(lib, qs[i], ts[i], mat, q_, e, q2, e2, ws[i], flag=gdo.EZ_APPROX_MAX | gdo.EZ_AVX512_SC)
            assert sc[i] == o["score"], (i, len(qs[i]), len(ts[i]), ws[i], sc[i], o["score"])
            assert np.array_equal(cg[i], o["cigar"]), (i, len(qs[i]), len(ts[i]), ws[i])
    finally:
        ctx.close()


def test_checkpointed_kernel_on_a_very_long_alignment(pkg, oracle, monkeypatch):
    """one 120 kbp x 120 kbp ONT-like pair (w = 1300: 500 chunks of 480 anti-diagonals, 319 MB of backtrace in the reference's layout,
    9.6 MB here) through the checkpointed wide-band kernel against the oracle, plus its consistency with the stored-backtrace kernel"""
    gdo, lib = oracle
    rng = np.random.default_rng(77)
    q, t = gdo.make_pair(rng, 120000, 0.03, 0.02, 0.02)
    q2, t2 = gdo.make_pair(rng, 30000, 0.03, 0.02, 0.02)
    a, b, q_, e, qq2, e2 = gdo.PRESETS["ont"]
    o = gdo.oracle_extd2(lib, q, t, gdo.score_matrix(a, b), q_, e, qq2, e2, 1300, flag=gdo.EZ_APPROX_MAX | gdo.EZ_AVX512_SC)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GDIET_WIDE_CKPT", mode)
        ctx = pkg.Context(0)
        try:
            sc, cg = ctx.ksw_extd2_batch([q, q2], [t, t2], 1300, pkg.KswScore.from_preset("ont"))
            assert ctx.last_kernel_mask() & 8
            res[mode] = (list(sc), [c.copy() for c in cg])
        finally:
            ctx.close()
    assert res["1"][0][0] == o["score"] and np.array_equal(res["1"][1][0], o["cigar"])
    assert res["1"][0] == res["0"][0] and all(np.array_equal(x, y) for x, y in zip(res["1"][1], res["0"][1]))
