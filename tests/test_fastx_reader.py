"""CPU: the read-input entry points (gdiet_hip_fastx_*; SURVEY 8f rank 2, input half) against the reference's own parser.

The reference prints an unmapped read as `name 4 * 0 0 * * 0 0 SEQ QUAL rl:i:0 [comment]` (LR/format.c), so mapping awkward FASTA /
FASTQ files against an unrelated contig with the reference binary (oracle/_ref, built from /root/reference by oracle/Makefile.ref)
shows exactly which name, sequence, quality string and comment its kseq_read / mm_bseq_read3 produced for every record.  A
committed fixture pins the same thing where the reference is absent: tests/golden/fastx/*.expected were written by this test's
generator from the reference's output (tests/golden/fastx/README)."""
import gzip
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, load_pkg

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "gdiet_lr_avx")
GOLD = os.path.join(ROOT, "tests", "golden", "fastx")


def _awkward_inputs(rng):
    """name -> bytes of a FASTA/FASTQ file exercising the record grammar"""
    def seq(n, alphabet=b"ACGT"):
        return bytes(rng.choice(list(alphabet), size=n).tolist())

    def wrap(s, w):
        return b"\n".join(s[i:i + w] for i in range(0, len(s), w))

    files = {}
    # 1. plain 4-line FASTQ, comments after the name, one read with U and lower case
    recs = []
    for i in range(40):
        s = seq(int(rng.integers(30, 400)), b"ACGTacgtUuN" if i % 7 == 0 else b"ACGT")
        q = bytes(rng.integers(33, 74, size=len(s), dtype=np.uint8).tolist())
        recs.append(b"@r%d comment %d\tmore\n" % (i, i) + s + b"\n+\n" + q + b"\n")
    files["plain.fq"] = b"".join(recs)
    # 2. multi-line FASTQ: wrapped sequence and quality, quality lines that start with '@' and '+', '+name' separator lines,
    #    empty lines between records
    recs = []
    for i in range(30):
        s = seq(int(rng.integers(61, 500)))
        q = bytearray(rng.integers(35, 74, size=len(s), dtype=np.uint8).tolist())
        q[0] = ord("@")
        if len(q) > 60:
            q[60] = ord("+") if i % 2 else ord("@")
        recs.append(b"@m%d_%d\n" % (i // 2, 1 + i % 2) + wrap(s, 60) + b"\n+m%d\n" % i + wrap(bytes(q), 60) + b"\n" + (b"\n" if i % 3 == 0 else b""))
    files["multiline.fq"] = b"".join(recs)
    # 3. Windows line ends, with and without a comment
    recs = []
    for i in range(20):
        s = seq(int(rng.integers(30, 200)))
        q = bytes(rng.integers(40, 74, size=len(s), dtype=np.uint8).tolist())
        recs.append(b"@w%d%s\r\n" % (i, b" c" if i % 2 else b"") + s + b"\r\n+\r\n" + q + b"\r\n")
    files["crlf.fq"] = b"".join(recs)
    # 4. multi-line FASTA, text before the first header, a record with an empty sequence, no newline at the end
    recs = [b"leading text that is not a record\n"]
    for i in range(25):
        s = b"" if i == 11 else seq(int(rng.integers(30, 700)))
        recs.append(b">f%d desc=%d\n" % (i, i) + (wrap(s, 70) + b"\n" if s else b""))
    files["multi.fa"] = b"".join(recs).rstrip(b"\n")
    # 5. FASTQ whose last record has a short quality string
    files["truncated.fq"] = files["plain.fq"][:2000].rsplit(b"@r", 1)[0] + b"@last\nACGTACGTACGT\n+\nIIII\n"
    # 6. a malformed record in the middle: the lines after it are swallowed / rescanned exactly as kseq_read does
    body = files["plain.fq"].split(b"\n@r")
    files["broken_mid.fq"] = b"\n@r".join(body[:12]) + b"\n@bad\nACGTACGTAC\n+\nII\n@r" + b"\n@r".join(body[12:])
    return files


def _reference_view(path, tmp_path, with_comment):
    """[(name, seq, qual or None, comment or None)] as the reference binary parsed them"""
    contig = os.path.join(str(tmp_path), "unrelated.fa")
    if not os.path.exists(contig):
        with open(contig, "w") as f:
            f.write(">c\n" + "ACGTTGCA" * 400 + "\n")
    cmd = [REF_BIN, "-t", "1", "-ax", "map-hifi", "-Z", "10", "-W", "2", "-i", "0.2", "-k", "19", "-w", "19", "-a"] + (["-y"] if with_comment else []) + [contig, path]
    out = subprocess.run(cmd, capture_output=True, check=True).stdout
    rows = []
    for line in out.split(b"\n"):
        if not line or line.startswith(b"@"):
            continue
        f = line.split(b"\t")
        assert f[1] == b"4", "a read of the parser test mapped: " + line[:80].decode()
        assert f[11].startswith(b"rl:i:")  # the one tag an unmapped record carries; -y appends the comment after it
        comment = b"\t".join(f[12:]) if len(f) > 12 else None
        rows.append((f[0], f[9], None if f[10] == b"*" else f[10], comment))
    return rows


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
    files = _awkward_inputs(np.random.default_rng(31))
    os.makedirs(GOLD, exist_ok=True)
    for name, data in files.items():
        for gz in (False, True):
            path = os.path.join(str(tmp_path), name + (".gz" if gz else ""))
            with (gzip.open(path, "wb") if gz else open(path, "wb")) as f:
                f.write(data)
            ours, sizes, trunc = _ours(pkg, path, 1500, with_comment)
            gold = os.path.join(GOLD, "%s.%s.expected.json" % (name, "y" if with_comment else "n"))
            if os.path.exists(REF_BIN):
                want = _reference_view(path, tmp_path, with_comment)
                if not gz:
                    with open(gold, "w") as f:  # refreshed wherever the reference is available; committed
                        json.dump([[x.decode("latin1") if x is not None else None for x in row] for row in want], f)
            else:
                with open(gold) as f:
                    want = [tuple(x.encode("latin1") if x is not None else None for x in row) for row in json.load(f)]
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
