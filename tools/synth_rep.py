#!/usr/bin/env python3
"""Repeat-rich synthetic inputs (our own generator; fixed PRNG seeds) for the fixtures under tests/golden/rep/.

    python tools/synth_rep.py ref   out.fa [--seed 21]
    python tools/synth_rep.py reads out.fq --ref out.fa --layout out.layout.json --kind hifi_rep|ont_rep|sr_rep [--n N] [--seed S]

Every other committed reference is repeat-free (index keys are >= 99.97 % singletons), so the high-occurrence machinery of the
seeding stage never runs on them.  This reference is built so that it does (pattern "10": only even periods repeat in the
sparsified sequence with their full copy number):
  * three contigs of 1.5 / 1.0 / 0.5 Mbp, i.i.d. ACGT;
  * dispersed families: A 2.5 kbp x 300 copies at 1 % divergence (both strands), B 700 bp x 80 at 3 %, C 300 bp x 900 at 1 %
    (nine in ten on the forward strand: a 150 bp read inside it collects several thousand hits on one strand);
  * tandem satellites: 64 bp x 4 600 (index keys above max_max_occ = 4095: never used as seeds), 40 bp x 1 500 (keys above every
    mid_occ and below max_max_occ: a read inside it is ONE streak of high-occurrence seeds, mm_seed_select's rescue heap with
    replacement), 171 bp x 700 (odd period: half the copy number per key);
  * microsatellites (AC)n, poly-A, (AG)n, (AAT)n of 2-4 kbp: their sparsified sequence is one repeated base or a period of 3, so every
    step of the winnowing automaton emits the same hash -- thousands of identical minimizers in one read, which is what
    mm_seed_mz_flt drops;
  * a few N runs.
Reads are drawn uniformly, centred on family copies, inside / across the satellites, and across the microsatellites; the read
name says which.  ont_rep reads carry a third of the usual ONT error rate (a k = 15 seed of the sparsified read spans 29 bases: at
7 % error almost no seed inside a microsatellite survives, and the query-side filter would have nothing to count)."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import synth  # noqa: E402

BASES = synth.BASES
CONTIGS = (1500000, 1000000, 500000)


def _mut(rng, s, div):
    s = s.copy()
    m = np.flatnonzero(rng.random(len(s)) < div)
    s[m] = BASES[(np.searchsorted(BASES, s[m]) + rng.integers(1, 4, size=len(m))) & 3]
    return s


def make_ref(seed=21):
    rng = np.random.default_rng(seed)
    contigs = [BASES[rng.integers(0, 4, size=l)] for l in CONTIGS]
    layout = {"families": {}, "satellites": [], "micro": []}
    # reserved intervals (satellites, microsatellites) are laid first and never overwritten by family copies
    reserved = [[] for _ in contigs]

    def reserve(c, pos, ln):
        reserved[c].append((pos, pos + ln))

    def free(c, pos, ln):
        return all(pos + ln <= a or pos >= b for a, b in reserved[c])

    for name, unit_len, copies, div, c, pos in (("sat64", 64, 4600, 0.0003, 0, 300000), ("sat40", 40, 1500, 0.001, 1, 200000), ("sat171", 171, 700, 0.002, 2, 100000)):
        unit = BASES[rng.integers(0, 4, size=unit_len)]
        arr = _mut(rng, np.tile(unit, copies), div)
        contigs[c][pos:pos + len(arr)] = arr
        reserve(c, pos, len(arr))
        layout["satellites"].append(dict(name=name, contig=c, pos=pos, len=len(arr), unit=unit_len))
    micro = (("AC", 3200, 0, 900000), ("AC", 2600, 1, 700000), ("AC", 3600, 2, 350000), ("A", 2200, 0, 1200000), ("A", 2000, 1, 50000),
             ("AG", 3000, 2, 420000), ("AAT", 3000, 0, 100000))
    for unit, ln, c, pos in micro:
        u = np.frombuffer(unit.encode(), np.uint8)
        arr = _mut(rng, np.tile(u, ln // len(u) + 1)[:ln], 0.0005)
        contigs[c][pos:pos + ln] = arr
        reserve(c, pos, ln)
        layout["micro"].append(dict(unit=unit, contig=c, pos=pos, len=ln))
    for name, ln, copies, div, p_rev in (("A", 2500, 300, 0.01, 0.5), ("B", 700, 80, 0.03, 0.5), ("C", 300, 900, 0.01, 0.1)):
        fam = BASES[rng.integers(0, 4, size=ln)]
        where = []
        while len(where) < copies:
            c = int(rng.choice(len(contigs), p=np.array(CONTIGS) / sum(CONTIGS)))
            pos = int(rng.integers(1000, len(contigs[c]) - ln - 1000))
            if not free(c, pos, ln):
                continue
            cp = _mut(rng, fam, div)
            rev = bool(rng.random() < p_rev)
            if rev:
                cp = synth.revcomp(cp)
            contigs[c][pos:pos + ln] = cp
            reserve(c, pos, ln)  # copies do not overwrite each other: the copy number is what the index sees
            where.append((c, pos, int(rev)))
        layout["families"][name] = dict(len=ln, copies=where)
    for c, pos, ln in ((0, 650000, 60), (1, 400000, 25), (2, 250000, 80)):
        if free(c, pos, ln):
            contigs[c][pos:pos + ln] = 78
    return contigs, layout


def _draw(rng, contigs, layout, ln, i):
    """(contig, start, tag) of read i: a fixed rotation over the kinds of region"""
    mode = i % 20
    L = [len(c) for c in contigs]
    if mode < 7:  # uniform
        c = int(rng.choice(len(contigs), p=np.array(L) / sum(L)))
        return c, int(rng.integers(0, L[c] - ln)), "u"
    if mode < 12:  # centred on (or hanging over the edge of) a family copy
        fam = "ABC"[int(rng.integers(0, 3))] if mode != 11 else "C"
        f = layout["families"][fam]
        c, pos, _ = f["copies"][int(rng.integers(0, len(f["copies"])))]
        st = pos + f["len"] // 2 - ln // 2 + int(rng.integers(-ln // 3, ln // 3 + 1))
        return c, min(max(st, 0), L[c] - ln), "f" + fam
    if mode < 16:  # inside a satellite or across one of its ends
        s = layout["satellites"][int(rng.integers(0, len(layout["satellites"])))]
        where = int(rng.integers(0, 3))
        if where == 0 and s["len"] > ln:
            st = s["pos"] + int(rng.integers(0, s["len"] - ln))
        elif where == 1:
            st = s["pos"] - int(rng.integers(ln // 4, 3 * ln // 4))
        else:
            st = s["pos"] + s["len"] - int(rng.integers(ln // 4, 3 * ln // 4))
        return s["contig"], min(max(st, 0), L[s["contig"]] - ln), "s" + s["name"][3:]
    m = layout["micro"][int(rng.integers(0, len(layout["micro"])))]  # across (or inside) a microsatellite
    st = m["pos"] + m["len"] // 2 - ln // 2 + int(rng.integers(-ln // 4, ln // 4 + 1))
    return m["contig"], min(max(st, 0), L[m["contig"]] - ln), "m" + m["unit"]


def make_reads(contigs, layout, kind, n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        if kind == "hifi_rep":
            ln = int(np.clip(rng.normal(9000, 1500), 4000, 16000))
            sub, ins, dele = 0.002, 0.0005, 0.0005
        elif kind == "ont_rep":
            ln = int(np.clip(rng.lognormal(np.log(14000), 0.3), 6000, 40000))
            sub, ins, dele = 0.01, 0.006, 0.006
        else:
            ln = 150 if i % 23 else int(rng.choice([100, 250, 400, 900, 1400]))  # (23: coprime with the 20 region kinds)
            sub, ins, dele = 0.01, 0.0005, 0.0005
        c, st, tag = _draw(rng, contigs, layout, ln, i)
        s = synth.mutate(rng, contigs[c][st:st + ln], sub, ins, dele)
        rev = rng.random() < 0.5
        if rev:
            s = synth.revcomp(s)
        out.append(("%s_%d_%s_c%d_%d_%s" % (kind, i, tag, c + 1, st, "-" if rev else "+"), s))
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["ref", "reads"])
    ap.add_argument("out")
    ap.add_argument("--ref")
    ap.add_argument("--layout")
    ap.add_argument("--kind", default="hifi_rep")
    ap.add_argument("--n", type=int, default=120)
    ap.add_argument("--seed", type=int, default=None)
    a = ap.parse_args()
    if a.what == "ref":
        contigs, layout = make_ref(21 if a.seed is None else a.seed)
        synth.write_fasta(a.out, contigs)
        with open(a.layout or a.out + ".layout.json", "w") as f:
            json.dump(layout, f)
    else:
        _, contigs = synth.read_fasta(a.ref)
        layout = json.load(open(a.layout or a.ref + ".layout.json"))
        synth.write_fastq(a.out, make_reads(contigs, layout, a.kind, a.n, {"hifi_rep": 31, "ont_rep": 32, "sr_rep": 33}[a.kind] if a.seed is None else a.seed))
