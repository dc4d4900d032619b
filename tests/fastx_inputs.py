"""inputs of the read-parser tests (seeded awkward FASTA / FASTQ files) and the probe that shows how the reference parsed a file.
Shared by tests/test_fastx_reader.py (reads the committed tests/golden/fastx/*.expected.json) and oracle/make_golden.py (the only
writer of those files)."""
import os
import subprocess

import numpy as np


def awkward_inputs(rng):
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


def reference_view(ref_bin, path, tmp_path, with_comment):
    """[(name, seq, qual or None, comment or None)] as the reference binary parsed them"""
    contig = os.path.join(str(tmp_path), "unrelated.fa")
    if not os.path.exists(contig):
        with open(contig, "w") as f:
            f.write(">c\n" + "ACGTTGCA" * 400 + "\n")
    cmd = [ref_bin, "-t", "1", "-ax", "map-hifi", "-Z", "10", "-W", "2", "-i", "0.2", "-k", "19", "-w", "19", "-a"] + (["-y"] if with_comment else []) + [contig, path]
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
