"""CPU: the read-input entry points (gdiet_hip_fastx_*; SURVEY 8f rank 2, input half) against the reference's own parser.

The reference prints an unmapped read as `name 4 * 0 0 * * 0 0 SEQ QUAL rl:i:0 [comment]` (LR/format.c), so mapping awkward FASTA /
FASTQ files against an unrelated contig with the reference binary (oracle/_ref, built from /root/reference by oracle/Makefile.ref)
shows exactly which name, sequence, quality string and comment its kseq_read / mm_bseq_read3 produced for every record.  A
committed fixture holds that view: tests/golden/fastx/*.expected.json, written by oracle/make_golden.py (never by a test) from the
reference's output; where oracle/_ref is built the test also checks that the reference still says what the fixture says."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_pkg
from fastx_inputs import awkward_inputs, reference_view

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "gdiet_lr_avx")
GOLD = os.path.join(ROOT, "tests", "golden", "fastx")


def _ours(pkg, path, chunk, with_comment):
    rows, sizes = [], []
    with pkg.FastxReader(path) as r:
        while True:
            b = r.read(chunk, with_qual=True, with_comment=with_comment)
            if not b and not r.truncated_now:
                break
            sizes.append(([len(x[1]) for x in b], r.truncated_now))
            rows += b
        trunc = r.truncated
    return rows, sizes, trunc


def _norm(rows, with_comment):
    # (the reference drops the comment column unless asked: -y copies it verbatim)
    return [(n, s, q, (c if with_comment else None)) for n, s, q, c in rows]


@pytest.mark.parametrize("with_comment", [False, True])
def test_reader_matches_the_reference_parser(tmp_path, with_comment):
    pkg = load_pkg()
    files = awkward_inputs(np.random.default_rng(31))
    for name, data in files.items():
        for gz in (False, True):
            path = os.path.join(str(tmp_path), name + (".gz" if gz else ""))
            with (gzip.open(path, "wb") if gz else open(path, "wb")) as f:
                f.write(data)
            ours, sizes, trunc = _ours(pkg, path, 1500, with_comment)
            gold = os.path.join(GOLD, "%s.%s.expected.json" % (name, "y" if with_comment else "n"))
            with open(gold) as f:
                want = [tuple(x.encode("latin1") if x is not None else None for x in row) for row in json.load(f)]
            if os.path.exists(REF_BIN):  # the committed fixture is the reference's view (also of the gzip'd file)
                assert [tuple(r) for r in reference_view(REF_BIN, path, tmp_path, with_comment)] == want, (name, gz)
            assert _norm(ours, with_comment) == [tuple(w) for w in want], (name, gz)
            assert trunc == (name in ("truncated.fq", "broken_mid.fq"))
            # mm_bseq_read3's batching rule: a batch closes with the record that brings its bases to chunk_size
            for b, closed_early in sizes[:-1]:  # (a malformed record closes its batch early, as in the reference)
                assert closed_early or (sum(b) >= 1500 and sum(b[:-1]) < 1500)


def test_fragment_mode_keeps_mates_together(tmp_path):
    pkg = load_pkg()
    path = os.path.join(str(tmp_path), "pairs.fq")
    with open(path, "wb") as f:
        for i in range(50):
            for m in (1, 2):
                f.write(b"@p%d/%d\n" % (i, m) + b"ACGT" * 25 + b"\n+\n" + b"I" * 100 + b"\n")
    with pkg.FastxReader(path) as r:
        names = []
        while True:
            b = r.read(250, frag_mode=True)  # 250 bases: closes after the third read, i.e. in the middle of a pair
            if not b:
                break
            assert b[-1][0].endswith(b"/2"), [x[0] for x in b]
            names += [x[0] for x in b]
    assert names == [b"p%d/%d" % (i, m) for i in range(50) for m in (1, 2)]


def test_reader_feeds_the_mapper_like_the_reference(tmp_path):
    """end to end on the committed fixture: reader -> names / sequences / qualities of the golden SAM (printed by GDiet_avx)"""
    from fixture_io import SR, golden_sam
    pkg = load_pkg()
    with pkg.FastxReader(os.path.join(SR, "sr.fq.gz")) as r:
        rows = []
        while True:
            b = r.read(100000)
            if not b:
                break
            rows += b
    primary = {}
    for line in golden_sam("sr"):
        f = line.split("\t")
        flag = int(f[1])
        if flag & 0x900:
            continue
        primary[f[0]] = (f[9], f[10], flag)
    comp = bytes.maketrans(b"ACGTN", b"TGCAN")
    assert len(rows) == len(primary)
    for n, s, q, c in rows:
        sam_seq, sam_qual, flag = primary[n.decode()]
        if flag & 16:
            s, q = s.translate(comp)[::-1], q[::-1]
        assert sam_seq == s.decode() and sam_qual == q.decode()


@pytest.mark.parametrize("block", [4096, 65536, 8 << 20])
def test_parallel_parse_equals_sequential_parse(tmp_path, block, monkeypatch):
    """several parser threads on speculative cuts give the records of the sequential grammar: a file of four-line FASTQ whose
    quality lines start with '@' and '+', with stretches of multi-line FASTQ, Windows line ends, FASTA and a malformed record in
    between (cuts that are no boundaries must be detected), parsed in blocks of 4 KB / 64 KB / 8 MB"""
    pkg = load_pkg()
    rng = np.random.default_rng(5)
    parts = []
    for i in range(6000):
        n = int(rng.integers(20, 300))
        s = bytes(rng.choice(list(b"ACGT"), size=n).tolist())
        q = bytearray(rng.integers(35, 74, size=n, dtype=np.uint8).tolist())
        if i % 3 == 0:
            q[0] = ord("@")
        if i % 5 == 0:
            q[0] = ord("+")
        zone = (i // 500) % 6
        if zone == 2 and n > 80:  # multi-line
            parts.append(b"@ml%d x\n" % i + s[:40] + b"\n" + s[40:] + b"\n+\n" + bytes(q[:40]) + b"\n" + bytes(q[40:]) + b"\n")
        elif zone == 3:
            parts.append(b"@cr%d\r\n" % i + s + b"\r\n+\r\n" + bytes(q) + b"\r\n")
        elif zone == 4:
            parts.append(b">fa%d some text\n" % i + s + b"\n")
        else:
            parts.append(b"@r%d c=%d\n" % (i, i) + s + b"\n+\n" + bytes(q) + b"\n")
        if i == 2750:
            parts.append(b"@broken\nACGTACGT\n+\nIII\n")
    path = os.path.join(str(tmp_path), "mixed.fq")
    with open(path, "wb") as f:
        f.write(b"".join(parts))
    monkeypatch.setenv("GDIET_FASTX_BLOCK", str(block))

    def all_batches(threads, chunk):
        out = []
        with pkg.FastxReader(path, threads=threads) as r:
            while True:
                b = r.read(chunk, with_qual=True, with_comment=True)
                if not b and not r.truncated_now:
                    break
                out.append((b, r.truncated_now))
        return out

    for chunk in (3000, 10 ** 7):
        want = all_batches(1, chunk)
        assert sum(len(b) for b, _ in want) == 6000
        assert sum(1 for _, t in want if t) == 1  # the malformed record closes exactly one batch early
        for threads in (2, 5):
            assert all_batches(threads, chunk) == want


def test_reader_threads_under_thread_sanitizer(tmp_path):
    """the parser threads and the read-ahead thread of csrc/fastx_reader.h, built alone with -fsanitize=thread"""
    exe = str(tmp_path / "fastx_test")
    subprocess.check_call(["g++", "-fsanitize=thread", "-O1", "-g", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"),
                           os.path.join(ROOT, "tests", "emul", "fastx_test.cpp"), "-o", exe, "-lz"])
    rng = np.random.default_rng(9)
    path = os.path.join(str(tmp_path), "t.fq")
    with open(path, "wb") as f:
        for i in range(3000):
            n = int(rng.integers(20, 200))
            s = bytes(rng.choice(list(b"ACGT"), size=n).tolist())
            q = bytearray(rng.integers(35, 74, size=n, dtype=np.uint8).tolist())
            if i % 4 == 0:
                q[0] = ord("@")
            if 1000 <= i < 1300 and n > 60:
                f.write(b"@m%d\n" % i + s[:30] + b"\n" + s[30:] + b"\n+\n" + bytes(q[:30]) + b"\n" + bytes(q[30:]) + b"\n")
            else:
                f.write(b"@r%d k\n" % i + s + b"\n+\n" + bytes(q) + b"\n")
    r = subprocess.run([exe, path], capture_output=True, text=True, timeout=600, env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0, r.stdout + r.stderr
    assert "ok 3000 records" in r.stdout and "ThreadSanitizer" not in r.stderr
