"""GPU parity of the whole per-read path (B1: gdiet_hip_map_batch through the C ABI): SAM records identical to the
reference's golden SAM for the HiFi and ONT presets; the seeding/voting kernels, the gather kernel, the DP/backtrack
kernels and the host post-processing are all on the path."""
import os

import pytest

from fixture_io import LR, OVERRIDES, PAF_KINDS, SETS, golden_paf, golden_sam, read_fasta, read_fastq, reads_of

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["hifi", "ont", "sr", "sr_var", "hifi_w1", "hifi_edge", "ont_edge", "sr_edge", "hifi_sv", "ont_sv",
                                  "hifi_rep", "ont_rep", "sr_rep", "sr_rep_f60"])
def test_map_batch_matches_golden_sam(gpu_ctx, pkg, kind):
    base, stem, preset = SETS[kind]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
    reads = reads_of(kind)
    m = pkg.Mapper(gpu_ctx, names, seqs, preset=preset, **OVERRIDES.get(kind, {}))
    try:
        res = m.map([r[1] for r in reads])
        got = []
        for i, (qn, sq, ql) in enumerate(reads):
            got += m.sam(res, i, qn, sq, ql)
        want = golden_sam(kind)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a == b, (a[:300], b[:300])
        assert m.sam_batch(res, reads) == "".join(l + "\n" for l in want)  # the batch formatter (host threads) prints the same bytes
        # every DP of the HiFi fixture must have gone through the register-resident kernel
        if kind in ("hifi", "hifi_sv"):
            assert gpu_ctx.last_kernel_mask() & 1
        if kind in ("ont", "ont_sv"):  # w = 1300: the two-blocks-per-lane kernel
            assert gpu_ctx.last_kernel_mask() & 8
        if kind == "sr":  # 150 x 150, w = 150 boxes: the 16-lane kernel
            assert gpu_ctx.last_kernel_mask() & 4
    finally:
        m.close()


@pytest.mark.parametrize("kind", list(PAF_KINDS))
def test_map_batch_matches_golden_paf(gpu_ctx, pkg, kind):
    """gdiet_hip_paf_batch (mm_write_paf3) on the records of the GPU path: the reference's PAF lines, -c --paf-no-hit"""
    base, stem, preset = SETS[kind]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
    reads = reads_of(kind)
    m = pkg.Mapper(gpu_ctx, names, seqs, preset=preset, **OVERRIDES.get(kind, {}))
    try:
        res = m.map([r[1] for r in reads])
        assert m.paf_batch(res, reads, flag=0x20 | 0x8000000) == "".join(l + "\n" for l in golden_paf(kind))
    finally:
        m.close()


def test_map_uploaded_is_idempotent(gpu_ctx, pkg):
    """size-independent property: mapping the same resident batch twice gives identical records"""
    names, seqs = read_fasta(os.path.join(LR, "ref.fa.gz"))
    reads = read_fastq(os.path.join(LR, "hifi.fq.gz"))
    m = pkg.Mapper(gpu_ctx, names, seqs, preset="hifi")
    try:
        b = m.upload([r[1] for r in reads])
        r1 = m.map_uploaded(b)
        s1 = [m.sam(r1, i, reads[i][0], reads[i][1], reads[i][2]) for i in range(len(reads))]
        r2 = m.map_uploaded(b)
        s2 = [m.sam(r2, i, reads[i][0], reads[i][1], reads[i][2]) for i in range(len(reads))]
        assert s1 == s2
        # ... and slicing the batch over pipeline lanes must not change a byte either
        b3 = m.upload([r[1] for r in reads] * 12)  # 336 reads -> 2 lanes x 2 slices
        m.set_lanes(3)
        r3 = m.map_uploaded(b3)
        m.set_lanes(1)
        for rep in range(12):
            s3 = [m.sam(r3, rep * len(reads) + i, reads[i][0], reads[i][1], reads[i][2]) for i in range(len(reads))]
            assert s3 == s1
        m.free_batch(b3)
        m.free_batch(b)
    finally:
        m.close()


def test_index_import_route_gives_identical_sam(gpu_ctx, pkg):
    """gdiet_hip_index_import (what the reference-side stub of INTEGRATION.md feeds from mm_idx_t::B[]) against
    gdiet_hip_index_build: export -> shuffle the key order (a khash walk has no particular order) -> import -> same SAM"""
    import numpy as np
    names, seqs = read_fasta(os.path.join(LR, "ref.fa.gz"))
    reads = read_fastq(os.path.join(LR, "hifi.fq.gz"))
    m = pkg.Mapper(gpu_ctx, names, seqs, preset="hifi")
    try:
        flat = m.export_index()
        assert len(flat["keys"]) == m.n_keys() and int(flat["cnt"].sum()) == len(flat["pos"])
        start = np.concatenate([[0], np.cumsum(flat["cnt"].astype(np.int64))])
        perm = np.random.default_rng(5).permutation(len(flat["keys"]))
        pos = np.concatenate([flat["pos"][start[j]:start[j + 1]] for j in perm])
        for j in perm[:200]:  # every list ascending, as mm_idx_get returns it (LR/index.c:255)
            lst = flat["pos"][start[j]:start[j + 1]]
            assert (np.diff(lst.astype(np.int64)) > 0).all()
        flat2 = dict(keys=flat["keys"][perm], cnt=flat["cnt"][perm], pos=pos, S=flat["S"], offsets=flat["offsets"])
        m2 = pkg.Mapper.from_flat(gpu_ctx, names, [len(s) for s in seqs], flat2, preset="hifi")
        try:
            assert m2.mid_occ == m.mid_occ
            ra, rb = m.map([r[1] for r in reads]), m2.map([r[1] for r in reads])
            for i, (qn, sq, ql) in enumerate(reads):
                assert m.sam(ra, i, qn, sq, ql) == m2.sam(rb, i, qn, sq, ql)
        finally:
            m2.close()
    finally:
        m.close()


def test_two_batches_in_flight_give_the_same_records(gpu_ctx, pkg):
    """gdiet_hip_map_submit / _wait: two resident batches in flight (different read sets), several rounds; every record equals
    what the synchronous call returns"""
    names, seqs = read_fasta(os.path.join(LR, "ref.fa.gz"))
    reads = read_fastq(os.path.join(LR, "hifi.fq.gz"))
    ra, rb = reads[: len(reads) // 2], reads[len(reads) // 2:]
    m = pkg.Mapper(gpu_ctx, names, seqs, preset="hifi")
    try:
        ba, bb = m.upload([r[1] for r in ra]), m.upload([r[1] for r in rb])

        def sam(res, rs):
            return [m.sam(res, i, rs[i][0], rs[i][1], rs[i][2]) for i in range(len(rs))]

        want_a, want_b = sam(m.map_uploaded(ba), ra), sam(m.map_uploaded(bb), rb)
        for _ in range(3):
            ta = m.submit(ba)
            tb = m.submit(bb)
            with pytest.raises(pkg.GdietError):  # a third ticket must be refused, not queued silently
                m.submit(ba)
            assert sam(m.wait(ta), ra) == want_a
            assert sam(m.wait(tb), rb) == want_b
        assert sam(m.map_uploaded(ba), ra) == want_a  # the synchronous path still works afterwards
        m.free_batch(ba)
        m.free_batch(bb)
    finally:
        m.close()


def test_mmi_index_files(gpu_ctx, pkg, tmp_path):
    """gdiet_hip_index_load_mmi on an index file written by the reference itself (`GDiet_avx -ax sr ... -d`, committed under
    tests/golden/sr) gives the golden SAM; gdiet_hip_index_dump_mmi -> load_mmi round-trips"""
    import gzip
    import shutil
    from fixture_io import SR
    names, seqs = read_fasta(os.path.join(SR, "ref.fa.gz"))
    reads = reads_of("sr")[:600]
    want = [l for l in golden_sam("sr") if l.split("\t")[0] in {r[0] for r in reads}]
    theirs = str(tmp_path / "theirs.mmi")
    with gzip.open(os.path.join(SR, "ref.k21w11.mmi.gz"), "rb") as src, open(theirs, "wb") as dst:
        shutil.copyfileobj(src, dst)

    def run(m):
        res = m.map([r[1] for r in reads])
        return [line for i, (qn, sq, ql) in enumerate(reads) for line in m.sam(res, i, qn, sq, ql)]

    m = pkg.Mapper.from_mmi(gpu_ctx, theirs, names, [len(s) for s in seqs], preset="sr")
    try:
        assert run(m) == want
        ours = str(tmp_path / "ours.mmi")
        m.dump_mmi(ours)
        m2 = pkg.Mapper.from_mmi(gpu_ctx, ours, names, [len(s) for s in seqs], preset="sr")
        try:
            assert m2.n_keys() == m.n_keys() and run(m2) == want
        finally:
            m2.close()
    finally:
        m.close()
    # the index built on the DEVICE from the FASTA sequences, dumped: the very bytes `GDiet_avx -ax sr ... -d` wrote (mm_idx_dump, khash
    # slot order included)
    m3 = pkg.Mapper(gpu_ctx, names, seqs, preset="sr")
    try:
        built = str(tmp_path / "built.mmi")
        m3.dump_mmi(built)
        assert open(built, "rb").read() == open(theirs, "rb").read()
    finally:
        m3.close()
    with pytest.raises(pkg.GdietError):
        pkg.Mapper.from_mmi(gpu_ctx, os.path.join(SR, "sr.cmd"), names, [len(s) for s in seqs], preset="sr")  # not an index file
    # an index whose window size exceeds what the sketch kernels hold (w = 100: header word 0) is refused, not mapped with
    big = str(tmp_path / "w100.mmi")
    raw = bytearray(open(theirs, "rb").read())
    raw[4:8] = (100).to_bytes(4, "little")
    open(big, "wb").write(bytes(raw))
    with pytest.raises(pkg.GdietError):
        pkg.Mapper.from_mmi(gpu_ctx, big, names, [len(s) for s in seqs], preset="sr")
    with pytest.raises(pkg.GdietError):
        pkg.Mapper(gpu_ctx, names, seqs, preset="sr", w=65)


@pytest.mark.parametrize("kind,seed_kernel", [("hifi_rep", "thread"), ("ont_rep", "thread"), ("sr_rep", "wave"), ("sr_rep_f60", "wave")])
def test_rep_sets_through_the_other_seed_executor(pkg, kind, seed_kernel, monkeypatch):
    """the repeat-rich sets through the executor their read length does NOT select by default (GDIET_SEED_KERNEL: long reads one
    per thread on the shared stage code, short reads one per wavefront with the LDS pre-check / wave compactions / LDS-run merge):
    both executors of S4-S7 give the reference's SAM where minimizers are dropped, rescued and cut at max_max_occ"""
    import torch  # noqa: F401
    monkeypatch.setenv("GDIET_SEED_KERNEL", seed_kernel)
    if seed_kernel == "thread":
        monkeypatch.setenv("GDIET_VOTE_WAVE", "0")
    ctx = pkg.Context(0)  # the executor is chosen when the context is created
    base, stem, preset = SETS[kind]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
    reads = reads_of(kind)
    m = pkg.Mapper(ctx, names, seqs, preset=preset, **OVERRIDES.get(kind, {}))
    try:
        res = m.map([r[1] for r in reads])
        assert m.sam_batch(res, reads) == "".join(l + "\n" for l in golden_sam(kind))
    finally:
        m.close()
        ctx.close()


@pytest.mark.parametrize("tag,kind", [("k19w19", "hifi_rep"), ("k15w10", "ont_rep"), ("k21w11", "sr_rep")])
def test_device_built_rep_mmi_is_the_reference_s(gpu_ctx, pkg, tmp_path, tag, kind):
    """the index of the repeat-rich reference BUILT ON THE DEVICE and dumped: size and sha256 of the file `GDiet_avx -d` wrote for it
    (tests/golden/rep/mmi.sha256.json) -- thousands of multi-occurrence position lists (p[] arrays), whose order comes from the two
    stable device sorts, and khash buckets with non-singleton keys"""
    import hashlib
    import json
    from fixture_io import REP
    want = json.load(open(os.path.join(REP, "mmi.sha256.json")))[tag]
    names, seqs = read_fasta(os.path.join(REP, "ref.fa.gz"))
    m = pkg.Mapper(gpu_ctx, names, seqs, preset=SETS[kind][2])
    try:
        out = str(tmp_path / "built.mmi")
        m.dump_mmi(out)
        data = open(out, "rb").read()
        assert len(data) == want["size"] and hashlib.sha256(data).hexdigest() == want["sha256"]
    finally:
        m.close()


@pytest.mark.parametrize("kind", ["hifi", "ont", "sr", "hifi_rep", "ont_rep", "sr_rep"])
def test_device_built_index_equals_host_built(gpu_ctx, pkg, kind, monkeypatch):
    """gdiet_hip_index_build on the device (sketch slices, radix sorts, run-length encode, CAS table) against the host builder:
    same keys, same counts, same position lists in the same order, same mid_occ"""
    import numpy as np
    base, stem, preset = SETS[kind]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))

    def flat(env):
        monkeypatch.setenv("GDIET_INDEX_BUILD", env)
        ctx = pkg.Context(0)  # the builder is chosen when the context is created
        m = pkg.Mapper(ctx, names, seqs, preset=preset)
        try:
            f = m.export_index()
            order = np.argsort(f["keys"], kind="stable")
            start = np.concatenate([[0], np.cumsum(f["cnt"].astype(np.int64))])
            pos = np.concatenate([f["pos"][start[j]:start[j + 1]] for j in order]) if len(order) else f["pos"]
            return f["keys"][order], f["cnt"][order], pos, f["S"], m.mid_occ, m.n_keys()
        finally:
            m.close()
            ctx.close()

    h, d = flat("host"), flat("device")
    assert h[5] == d[5] and h[4] == d[4]
    for a, b in zip(h[:4], d[:4]):
        assert np.array_equal(a, b)


@pytest.fixture(scope="module")
def host_emulator(tmp_path_factory):
    """the CPU driver of tests/test_map_host.py (the product's stage code compiled for the host + the oracle's DP), pinned to the
    reference binary by the CPU suite; here it supplies expected SAM for read sets that have no committed golden file"""
    import subprocess
    from conftest import ROOT
    d = tmp_path_factory.mktemp("emu")
    exe = str(d / "map_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-w", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"),
                           "-I", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "emul", "map_host_main.cpp"),
                           "-x", "c", os.path.join(ROOT, "oracle", "gdo_ksw2.c"), "-o", exe])
    return exe, str(d)


@pytest.mark.parametrize("kind,n,length", [("hifi", 160, 3000), ("ont", 20, 24000), ("sr", 4000, 150)])
def test_fresh_reads_match_host_emulator(gpu_ctx, pkg, host_emulator, kind, n, length):
    """a few thousand alignments per preset on reads drawn at test time (fixed seed): GPU path == host emulator, byte for byte"""
    import gzip
    import subprocess
    import numpy as np
    from fixture_io import cmd_of
    exe, d = host_emulator
    base, stem, preset = SETS[kind]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
    rng = np.random.default_rng({"hifi": 101, "ont": 102, "sr": 103}[kind])
    sub, ind = {"hifi": (0.004, 0.002), "ont": (0.03, 0.02), "sr": (0.01, 0.001)}[kind]
    comp = str.maketrans("ACGTN", "TGCAN")
    reads = []
    for i in range(n):
        ln = length if kind == "sr" else int(rng.integers(length // 2, 2 * length))
        c = int(rng.choice([j for j in range(len(seqs)) if len(seqs[j]) > ln + 10]))
        st = int(rng.integers(0, len(seqs[c]) - ln))
        s = np.frombuffer(seqs[c][st:st + ln].encode(), np.uint8).copy()
        m = np.flatnonzero(rng.random(ln) < sub)
        s[m] = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=len(m))]
        s = s[rng.random(len(s)) >= ind]
        ip = np.flatnonzero(rng.random(len(s)) < ind)
        s = np.insert(s, ip, np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, size=len(ip))])
        q = s.tobytes().decode()
        if rng.random() < 0.5:
            q = q.translate(comp)[::-1]
        reads.append(("f%d" % i, q, "I" * len(q)))
    ref_fa, fq = os.path.join(d, kind + ".fa"), os.path.join(d, kind + ".fq")
    with gzip.open(os.path.join(base, "ref.fa.gz"), "rb") as src, open(ref_fa, "wb") as dst:
        dst.write(src.read())
    with open(fq, "w") as f:
        for nm, q, ql in reads:
            f.write("@%s\n%s\n+\n%s\n" % (nm, q, ql))
    want = subprocess.run([exe] + cmd_of(kind) + [ref_fa, fq], capture_output=True, text=True, check=True).stdout
    m = pkg.Mapper(gpu_ctx, names, seqs, preset=preset, **OVERRIDES.get(kind, {}))
    try:
        res = m.map([r[1] for r in reads])
        assert m.sam_batch(res, reads) == want
        mapped = sum(1 for i in range(n) if res.n_regs[i] > 0)
        assert mapped > 0.5 * n
    finally:
        m.close()


def test_three_batches_in_flight_short_reads(gpu_ctx, pkg):
    """gdiet_hip_set_inflight(3) with the ShortReads variant: three tickets open at once, several rounds, records identical to the
    synchronous call; afterwards the depth can be changed back"""
    from fixture_io import SR
    names, seqs = read_fasta(os.path.join(SR, "ref.fa.gz"))
    reads = reads_of("sr")
    parts = [reads[0:700], reads[700:1300], reads[1300:2000]]
    m = pkg.Mapper(gpu_ctx, names, seqs, preset="sr")
    try:
        batches = [m.upload([r[1] for r in p]) for p in parts]
        want = [m.sam_batch(m.map_uploaded(b), p) for b, p in zip(batches, parts)]
        assert "".join(want) == "".join(l + "\n" for l in golden_sam("sr"))
        m.set_inflight(3)
        for _ in range(3):
            tickets = [m.submit(b) for b in batches]
            got = [m.sam_batch(m.wait(t), p) for t, p in zip(tickets, parts)]
            assert got == want
        m.set_inflight(2)
        t = m.submit(batches[0])
        assert m.sam_batch(m.wait(t), parts[0]) == want[0]
        # small batches: every lane had a backtrace arena of its own above; now the lanes take turns in the shared one
        os.environ["GDIET_LANE_ARENA_GB"] = "0"
        try:
            m.set_inflight(3)
        finally:
            del os.environ["GDIET_LANE_ARENA_GB"]
        tickets = [m.submit(b) for b in batches]
        assert [m.sam_batch(m.wait(t), p) for t, p in zip(tickets, parts)] == want
        for b in batches:
            m.free_batch(b)
    finally:
        m.close()


@pytest.mark.parametrize("kind", ["hifi", "sr"])
def test_file_to_sam_through_the_library(gpu_ctx, pkg, kind):
    """steps 0-2 of the reference's worker_pipeline through the C ABI alone: gdiet_hip_fastx_read (mini-batches of the FASTQ file)
    -> map -> gdiet_hip_sam_batch; the concatenation is the golden SAM body, whatever the mini-batch size"""
    base, stem, preset = SETS[kind]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
    m = pkg.Mapper(gpu_ctx, names, seqs, preset=preset, **OVERRIDES.get(kind, {}))
    try:
        for chunk in (60000, 10 ** 9):
            out = []
            with pkg.FastxReader(os.path.join(base, stem + ".fq.gz")) as r:
                while True:
                    b = r.read(chunk)
                    if not b:
                        break
                    res = m.map([x[1] for x in b])
                    out.append(m.sam_batch(res, [(x[0], x[1], x[2]) for x in b]))
            assert "".join(out) == "".join(l + "\n" for l in golden_sam(kind))
    finally:
        m.close()


def test_file_to_sam_with_batches_in_flight(gpu_ctx, pkg, tmp_path):
    """tools/map_file.py's route: detached reader batches (C arrays, no Python object per read), three mini-batches in flight, SAM
    written batch by batch -- the file is the golden SAM body"""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import map_file
    from fixture_io import SR
    names, seqs = read_fasta(os.path.join(SR, "ref.fa.gz"))
    m = pkg.Mapper(gpu_ctx, names, seqs, preset="sr")
    try:
        for chunk, threads in ((30000, 1), (45000, 3)):
            out = str(tmp_path / ("out%d.sam" % chunk))
            with open(out, "wb") as f:
                n, _ = map_file.map_file(pkg, m, os.path.join(SR, "sr.fq.gz"), f, chunk, 3, threads)
            assert n == 2000
            assert open(out).read() == "".join(l + "\n" for l in golden_sam("sr"))
    finally:
        m.close()


def test_file_to_sam_failure_does_not_hang(gpu_ctx, pkg, tmp_path):
    """tools/map_file.py when a stage fails (here: the output sink raises on its second write): the call raises instead of hanging,
    every open ticket has been waited for, and the mapper maps the next file as if nothing had happened"""
    import sys
    import threading
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import map_file
    from fixture_io import SR
    names, seqs = read_fasta(os.path.join(SR, "ref.fa.gz"))
    m = pkg.Mapper(gpu_ctx, names, seqs, preset="sr")

    class Sink:
        def __init__(self):
            self.n = 0

        def write(self, b):
            self.n += 1
            if self.n == 2:
                raise IOError("disk full")
            return len(b)

    try:
        box = {}

        def run():
            try:
                map_file.map_file(pkg, m, os.path.join(SR, "sr.fq.gz"), Sink(), 15000, 3, 2)
                box["r"] = "no error"
            except IOError as e:
                box["r"] = str(e)
        th = threading.Thread(target=run, daemon=True)
        th.start()
        th.join(120)
        assert not th.is_alive(), "map_file hangs after a failed write"
        assert box.get("r") == "disk full"
        out = str(tmp_path / "after.sam")
        with open(out, "wb") as f:
            n, _ = map_file.map_file(pkg, m, os.path.join(SR, "sr.fq.gz"), f, 45000, 3, 2)
        assert n == 2000 and open(out).read() == "".join(l + "\n" for l in golden_sam("sr"))
    finally:
        m.close()


@pytest.mark.parametrize("kind,n_ctx", [("hifi_sv", 2), ("sr", 2), ("hifi_rep", 3)])
def test_single_process_fan_out_over_contexts(pkg, kind, n_ctx):
    """gdiet_hip_map_batch_multi (one process, one context per GPU, the mini-batch cut into contiguous ranges of equal DP
    cost, records gathered in input order): n_ctx contexts on the one device of the test box, each with its own copy of the index;
    the records are the golden SAM, byte for byte, exactly as from one context"""
    import torch  # noqa: F401
    base, stem, preset = SETS[kind]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
    reads = reads_of(kind)
    ctxs = [pkg.Context(0) for _ in range(n_ctx)]
    ms = [pkg.Mapper(c, names, seqs, preset=preset, **OVERRIDES.get(kind, {})) for c in ctxs]
    try:
        for m in ms:
            m.set_host_threads(max(1, pkg.effective_cpus() // n_ctx))
        res = pkg.map_multi(ms, [r[1] for r in reads])
        assert ms[0].sam_batch(res, reads) == "".join(l + "\n" for l in golden_sam(kind))
        bounds = pkg.read_ranges_by_cost_c([len(r[1]) for r in reads], n_ctx, band=1000 if preset != "sr" else 150)
        assert all(hi > lo for lo, hi in zip(bounds, bounds[1:]))  # every context had work
        del res
        # errors are joined: a bad option fails the call with a text that names the context, and hands nothing back
        bad = pkg.MapOpt.from_buffer_copy(ms[0].opt)
        ms[0].opt, keep = bad, ms[0].opt
        ms[0].opt.q2 = 120
        with pytest.raises(pkg.GdietError):
            pkg.map_multi(ms, [r[1] for r in reads[:8]])
        ms[0].opt = keep
    finally:
        for m in ms:
            m.close()
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("kind", ["hifi", "sr"])
def test_degenerate_box_fails_its_read_not_the_batch(kind):
    """a read whose DP box is degenerate (fault injection, GDIET_FAULT_BOX=<read index>: no read built so far produces one) comes
    back unmapped and is counted; every other read of the batch gets its golden records -- synchronously and with a batch in flight
    (tests/fault_box_check.py, in a process of its own: the library reads the variable once)"""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fault_box_check.py")
    r = subprocess.run([sys.executable, script, kind], capture_output=True, text=True, env=dict(os.environ, GDIET_FAULT_BOX="5"), timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


@pytest.mark.parametrize("env,kinds", [({"GDIET_POST_WAVE": "0"}, ["hifi_sv", "ont_sv", "hifi_rep"]),  # long alignments through the one-thread-per-alignment P1 kernel
                                       ({"GDIET_POST_WAVE": "1"}, ["sr", "sr_var", "sr_edge", "hifi_edge"]),  # short ones through the wave-parallel form
                                       ({"GDIET_FUSE_BT_GROUPS": "1"}, ["sr", "sr_var", "sr_edge", "sr_rep"]),  # grouped DP kernels walking their own alignments back
                                       ({"GDIET_SR_BOXES": "host"}, ["sr", "sr_var", "sr_edge"]),  # ShortReads candidate geometry on host threads
                                       ({"GDIET_SR_PIPE": "0"}, ["sr", "sr_var", "sr_edge", "sr_rep"]),  # short alignments on the grouped kernels only (no skewed pipelines)
                                       ({"GDIET_DIAG_SHORTCUT": "0"}, ["sr", "sr_var", "sr_rep"])])  # every short alignment walked back, also those whose score equals the main diagonal's
def test_golden_sam_under_the_other_post_kernel(env, kinds):
    """switches that select between two implementations of a stage (P1: one thread / one wavefront per alignment; the short-alignment
    DP kernels with / without their own walk; the ShortReads box stage on the device / the host; full-matrix short alignments as skewed pipelines / on the grouped kernels): each kind through the one it does
    not take by default: same SAM, ms:i / NM / de:f / dp_max tags included (tests/golden_env_check.py)"""
    import subprocess
    import sys
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_env_check.py")
    r = subprocess.run([sys.executable, script] + kinds, capture_output=True, text=True, env=dict(os.environ, **env), timeout=900)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_dp_kernel_at_four_wavefronts_per_simd(gpu_ctx, pkg):
    """gdiet_hip_set_dp_waves(4): the 64-lane DP kernel capped at four wavefronts per SIMD (room for the next batch's seeding / voting
    kernels): the same records, synchronously and with two batches in flight"""
    base, stem, preset = SETS["hifi_sv"]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
    reads = reads_of("hifi_sv")
    want = "".join(l + "\n" for l in golden_sam("hifi_sv"))
    m = pkg.Mapper(gpu_ctx, names, seqs, preset=preset)
    try:
        gpu_ctx.set_dp_waves(4)
        assert m.sam_batch(m.map([r[1] for r in reads]), reads) == want
        b = m.upload([r[1] for r in reads])
        m.set_inflight(2)
        t1, t2 = m.submit(b), m.submit(b)
        assert m.sam_batch(m.wait(t1), reads) == want and m.sam_batch(m.wait(t2), reads) == want
        m.free_batch(b)
        with pytest.raises(pkg.GdietError):
            gpu_ctx.set_dp_waves(3)
    finally:
        gpu_ctx.set_dp_waves(5)
        m.close()
