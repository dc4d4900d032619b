#!/usr/bin/env python3
"""Deterministic synthetic inputs of SURVEY.md 8d (our own generator; fixed PRNG seeds).

    python tools/synth.py ref  out.fa  --contigs 500000,300000,200000 [--repeats 20] [--seed 2]
    python tools/synth.py reads out.fq --ref out.fa --kind hifi|ont|sr --n 100 [--seed 5]
    python tools/synth.py reads out.fq --ref out.fa --kind hifi_sv|ont_sv --n 150 [--seed 11] [--mean-len 9000]

The *_sv kinds carry one structural difference from the reference per read at a random breakpoint -- deletion, insertion,
chimera, tandem duplication, inversion (round-robin), 0.7-4.5 kbp -- so that the second voting round (vote_2), candidate
linking, concatenate_cigars and the secondary / supplementary records of the LongReads path are exercised.
"""
import argparse
import sys

import numpy as np

BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = {65: 84, 67: 71, 71: 67, 84: 65, 78: 78}


def make_ref(contig_lens, seed=2, n_families=20, repeat_cov=0.05, div=0.02, n_frac=0.0001):
    rng = np.random.default_rng(seed)
    contigs = [BASES[rng.integers(0, 4, size=l)] for l in contig_lens]
    total = sum(contig_lens)
    # repeat layer: families of length U[300,6000] re-inserted to cover ~repeat_cov of the genome at `div` divergence
    fams = [BASES[rng.integers(0, 4, size=int(rng.integers(300, 6000)))] for _ in range(n_families)]
    covered = 0
    while covered < repeat_cov * total and fams:
        f = fams[int(rng.integers(0, len(fams)))]
        c = int(rng.integers(0, len(contigs)))
        if len(contigs[c]) <= len(f) + 10:
            continue
        pos = int(rng.integers(0, len(contigs[c]) - len(f)))
        cp = f.copy()
        m = rng.random(len(cp)) < div
        cp[m] = BASES[rng.integers(0, 4, size=int(m.sum()))]
        if rng.random() < 0.5:
            cp = np.array([COMP[x] for x in cp[::-1]], dtype=np.uint8)
        contigs[c][pos:pos + len(cp)] = cp
        covered += len(cp)
    # N runs
    n_runs = max(1, int(total * n_frac / 50))
    for _ in range(n_runs):
        c = int(rng.integers(0, len(contigs)))
        if len(contigs[c]) < 200:
            continue
        pos = int(rng.integers(0, len(contigs[c]) - 100))
        contigs[c][pos:pos + int(rng.integers(10, 90))] = 78
    return contigs


def write_fasta(path, contigs):
    with open(path, "wb") as f:
        for i, c in enumerate(contigs):
            f.write(b">chr%d\n" % (i + 1))
            for j in range(0, len(c), 80):
                f.write(c[j:j + 80].tobytes() + b"\n")


def read_fasta(path):
    names, seqs, cur = [], [], []
    for line in open(path, "rb"):
        line = line.rstrip()
        if line.startswith(b">"):
            if cur:
                seqs.append(np.frombuffer(b"".join(cur), dtype=np.uint8))
            names.append(line[1:].split()[0].decode())
            cur = []
        else:
            cur.append(line)
    if cur:
        seqs.append(np.frombuffer(b"".join(cur), dtype=np.uint8))
    return names, seqs


def mutate(rng, s, sub, ins, dele):
    s = s.copy()
    m = rng.random(len(s)) < sub
    idx = np.flatnonzero(m & (s != 78))
    if len(idx):
        cur = np.searchsorted(BASES, s[idx])
        s[idx] = BASES[(cur + rng.integers(1, 4, size=len(idx))) & 3]
    s = s[~(rng.random(len(s)) < dele)]
    ipos = np.flatnonzero(rng.random(len(s)) < ins)
    s = np.insert(s, ipos, BASES[rng.integers(0, 4, size=len(ipos))])
    return s


def make_reads(contigs, kind, n, seed):
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.float64)
    out = []
    for i in range(n):
        if kind == "hifi":
            ln = int(np.clip(rng.normal(15000, 2000), 5000, 25000))
            sub, ins, dele = 0.002, 0.001, 0.001
        elif kind == "ont":
            ln = int(np.clip(rng.lognormal(np.log(50000), 0.35), 5000, 150000))
            sub, ins, dele = 0.03, 0.02, 0.02
        else:
            ln = 150
            sub, ins, dele = 0.01, 0.0005, 0.0005
        c = int(rng.choice(len(contigs), p=lens / lens.sum()))
        ln = min(ln, len(contigs[c]) - 2)
        st = int(rng.integers(0, len(contigs[c]) - ln))
        s = mutate(rng, contigs[c][st:st + ln], sub, ins, dele)
        rev = rng.random() < 0.5
        if rev:
            s = np.array([COMP[x] for x in s[::-1]], dtype=np.uint8)
        out.append(("%s_%d_c%d_%d_%s" % (kind, i, c + 1, st, "-" if rev else "+"), s))
    return out


def revcomp(s):
    return np.array([COMP[x] for x in s[::-1]], dtype=np.uint8)


SV_KINDS = ("del", "ins", "chim", "dup", "inv")


def make_sv_reads(contigs, kind, n, seed, mean_len=None):
    """reads with one structural variant each (kinds round-robin, gap U[700,4500] at a breakpoint 20-80 % into the read)"""
    rng = np.random.default_rng(seed)
    lens = np.array([len(c) for c in contigs], dtype=np.float64)
    if kind == "hifi_sv":
        mu, sd, lo, hi = (mean_len or 15000), 0.13, 5000, 25000
        sub, ins, dele = 0.002, 0.0005, 0.0005
    else:
        mu, sd, lo, hi = (mean_len or 20000), 0.2, 8000, 60000
        sub, ins, dele = 0.03, 0.02, 0.02
    out = []
    for i in range(n):
        sv = SV_KINDS[i % len(SV_KINDS)]
        ln = int(np.clip(rng.normal(mu, mu * sd), lo, hi))
        gap = int(rng.integers(700, 4501))
        c = int(rng.choice(len(contigs), p=lens / lens.sum()))
        ln = min(ln, len(contigs[c]) - gap - 2)
        st = int(rng.integers(0, len(contigs[c]) - ln - gap))
        bp = int(ln * rng.uniform(0.2, 0.8))
        g = contigs[c]
        if sv == "del":
            s = np.concatenate([g[st:st + bp], g[st + bp + gap:st + ln + gap]])
        elif sv == "ins":
            s = np.concatenate([g[st:st + bp], BASES[rng.integers(0, 4, size=gap)], g[st + bp:st + ln]])
        elif sv == "chim":
            c2 = int(rng.choice(len(contigs), p=lens / lens.sum()))
            l2 = min(ln - bp, len(contigs[c2]) - 2)
            st2 = int(rng.integers(0, len(contigs[c2]) - l2))
            tail = contigs[c2][st2:st2 + l2]
            if rng.random() < 0.5:
                tail = revcomp(tail)
            s = np.concatenate([g[st:st + bp], tail])
        elif sv == "dup":
            gap = min(gap, bp)
            s = np.concatenate([g[st:st + bp], g[st + bp - gap:st + ln]])
        else:
            gap = min(gap, ln - bp)
            s = np.concatenate([g[st:st + bp], revcomp(g[st + bp:st + bp + gap]), g[st + bp + gap:st + ln]])
        s = mutate(rng, s, sub, ins, dele)
        rev = rng.random() < 0.5
        if rev:
            s = revcomp(s)
        out.append(("%s_%d_%s%d_c%d_%d_%s" % (kind, i, sv, gap, c + 1, st, "-" if rev else "+"), s))
    return out


def write_fastq(path, reads):
    with open(path, "wb") as f:
        for name, s in reads:
            f.write(b"@" + name.encode() + b"\n" + s.tobytes() + b"\n+\n" + b"I" * len(s) + b"\n")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["ref", "reads"])
    ap.add_argument("out")
    ap.add_argument("--contigs", default="500000,300000,200000")
    ap.add_argument("--repeats", type=int, default=20)
    ap.add_argument("--ref")
    ap.add_argument("--kind", default="hifi")
    ap.add_argument("--n", type=int, default=100)
    ap.add_argument("--seed", type=int, default=None)
    ap.add_argument("--mean-len", type=int, default=None)
    a = ap.parse_args()
    if a.what == "ref":
        write_fasta(a.out, make_ref([int(x) for x in a.contigs.split(",")], seed=2 if a.seed is None else a.seed, n_families=a.repeats))
    else:
        _, contigs = read_fasta(a.ref)
        if a.kind.endswith("_sv"):
            write_fastq(a.out, make_sv_reads(contigs, a.kind, a.n, 11 if a.seed is None else a.seed, a.mean_len))
        else:
            write_fastq(a.out, make_reads([c.copy() for c in contigs], a.kind, a.n, 5 if a.seed is None else a.seed))
