"""readers for the committed LongReads fixtures (tests/golden/lr/*; made by tools/synth.py + the reference binary)"""
import gzip
import os

LR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr")


def read_fasta(path):
    names, seqs, cur = [], [], []
    op = gzip.open if path.endswith(".gz") else open
    for line in op(path, "rt"):
        line = line.rstrip()
        if line.startswith(">"):
            if names:
                seqs.append("".join(cur))
            names.append(line[1:].split()[0])
            cur = []
        else:
            cur.append(line)
    seqs.append("".join(cur))
    return names, seqs


def read_fastq(path):
    op = gzip.open if path.endswith(".gz") else open
    lines = [l.rstrip() for l in op(path, "rt")]
    return [(lines[i][1:].split()[0], lines[i + 1], lines[i + 3]) for i in range(0, len(lines) - 3, 4)]


def golden_sam(kind):
    return [l.rstrip("\n") for l in gzip.open(os.path.join(LR, kind + ".golden.sam.gz"), "rt")]
