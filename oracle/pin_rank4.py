#!/usr/bin/env python3
"""Pin the SURVEY 8f rank-4 oracles -- gdo_ksw_exts2 (oracle/gdo_ksw2.c) and gdo_lchain_dp (oracle/gdo_lchain.c) -- against the
reference's own ksw_exts2_sse and mg_lchain_dp (oracle/_ref/libgdiet_sr_avx.so, built from /root/reference by oracle/Makefile.ref)
and (re)generate the committed golden vectors tests/golden/ksw2_exts2.npz and tests/golden/lchain_dp.npz (inputs + the
REFERENCE's outputs).  Runs only where oracle/_ref exists (this container); TEST INFRASTRUCTURE.

    python oracle/pin_rank4.py                  # differential fuzz only
    python oracle/pin_rank4.py --write-golden   # also rewrite the two .npz files
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gdo  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(gdo.HERE), "tests", "golden")
EZ_KEYS = ("score", "zdropped", "max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "reach_end")


def splice_pair(rng, big=False):
    """a transcript-like query against a genome-like target: 1-4 exons separated by introns, most of them bounded by the
    canonical GT..AG (or CT..AC) signals, some with the GTr / yAG flanks; substitutions and small indels in the query"""
    exons = [rng.integers(0, 4, size=int(rng.integers(15, 400 if big else 120)), dtype=np.uint8) for _ in range(int(rng.integers(1, 5)))]
    parts = [exons[0]]
    for ex in exons[1:]:
        intron = rng.integers(0, 4, size=int(rng.integers(20, 900 if big else 300)), dtype=np.uint8)
        c = rng.random()
        if c < 0.6:
            intron[:2], intron[-2:] = [2, 3], [0, 2]  # GT .. AG
            if rng.random() < 0.5:
                intron[2] = rng.choice([0, 2])  # GTr
            if rng.random() < 0.5:
                intron[-3] = rng.choice([1, 3])  # yAG
        elif c < 0.8:
            intron[:2], intron[-2:] = [1, 3], [0, 1]  # CT .. AC (reverse strand)
        parts += [intron, ex]
    t = np.concatenate(parts)
    q = np.concatenate(exons).copy()
    m = rng.random(len(q)) < 0.03
    q[m] = (q[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
    q = q[rng.random(len(q)) >= 0.01]
    if len(q) == 0:
        q = t[:1].copy()
    return q, t


def exts2_cases(rng, n):
    for i in range(n):
        q, t = splice_pair(rng, big=(i % 10 == 9))
        if i % 7 == 0:
            t = t.copy()
            t[rng.integers(0, len(t), size=max(1, len(t) // 40))] = 4
        if i % 11 == 5:  # unrelated tail: z-drop / a maximum inside the matrix
            q = np.concatenate([q[:max(1, len(q) // 2)], rng.integers(0, 4, size=int(rng.integers(10, 120)), dtype=np.uint8)])
        a, b = (1, 2) if i % 2 else (2, 4)
        mat = gdo.score_matrix(a, b)
        if i % 5 == 0:
            mat = mat.copy()
            mat[24] = -1
        go, ge, go2, nc = (2, 1, 32, 9) if i % 3 else (4, 2, 24, 5)
        flag = [0, gdo.EZ_SPLICE_FOR, gdo.EZ_SPLICE_REV, gdo.EZ_SPLICE_FOR | gdo.EZ_SPLICE_REV][i % 4]
        if (i // 4) % 2:
            flag |= gdo.EZ_SPLICE_FLANK
        if (i // 8) % 2:
            flag |= gdo.EZ_APPROX_MAX
        flag |= [0, gdo.EZ_RIGHT, gdo.EZ_REV_CIGAR][(i // 16) % 3]
        if (i // 48) % 2:
            flag |= gdo.EZ_EXTZ_ONLY
        flag |= [0, gdo.EZ_APPROX_DROP, gdo.EZ_GENERIC_SC, gdo.EZ_SCORE_ONLY][(i // 96) % 4]
        zdrop = (-1, 30, 200)[(i // 3) % 3]
        junc, jb = None, 0
        if i % 6 == 1:
            junc = rng.integers(0, 16, size=len(t), dtype=np.uint8)
            junc[rng.random(len(t)) < 0.8] = 0
            jb = 3
        yield q, t, mat, go, ge, go2, nc, zdrop, jb, flag, junc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fuzz", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=14)
    ap.add_argument("--write-golden", action="store_true")
    args = ap.parse_args()
    gdo.build_oracle()
    ora = gdo.load_oracle()
    ref = gdo.load_ref("sr_avx")
    rng = np.random.default_rng(args.seed)
    bad, gold, n_skip_ops = 0, [], 0
    for i, (q, t, mat, go, ge, go2, nc, zdrop, jb, flag, junc) in enumerate(exts2_cases(rng, args.fuzz)):
        o = gdo.oracle_exts2(ora, q, t, mat, go, ge, go2, nc, zdrop, jb, flag, junc)
        r = gdo.ref_exts2(ref, q, t, mat, go, ge, go2, nc, zdrop, jb, flag, junc)
        if not gdo.same(o, r, EZ_KEYS):
            bad += 1
            if bad <= 5:
                print("MISMATCH exts2 oracle vs reference", i, len(q), len(t), hex(flag), zdrop, {k: (o[k], r[k]) for k in EZ_KEYS if o[k] != r[k]})
        if i < 384:
            gold.append((q, t, mat, go, ge, go2, nc, zdrop, jb, flag, junc, r))
            n_skip_ops += int(((r["cigar"] & 0xf) == 3).sum())
    print("ksw_exts2: pairs=%d oracle_vs_reference_mismatch=%d (golden: %d pairs, %d N_SKIP ops, %d z-dropped)"
          % (args.fuzz, bad, len(gold), n_skip_ops, sum(1 for g in gold if g[11]["zdropped"])))
    bad_lc = 0
    gold_lc = []
    if hasattr(gdo, "lchain_cases"):
        lref = ref
        for i, case in enumerate(gdo.lchain_cases(np.random.default_rng(args.seed + 1), max(200, args.fuzz // 5))):
            o = gdo.oracle_lchain(ora, *case)
            r = gdo.ref_lchain(lref, *case)
            if not gdo.same_lchain(o, r):
                bad_lc += 1
                if bad_lc <= 5:
                    print("MISMATCH lchain oracle vs reference", i, len(case[0]), len(o["u"]), len(r["u"]))
            if i < 64:
                gold_lc.append((case, r))
        print("mg_lchain_dp: cases=%d oracle_vs_reference_mismatch=%d (golden: %d cases, %d chains)"
              % (max(200, args.fuzz // 5), bad_lc, len(gold_lc), sum(len(g[1]["u"]) for g in gold_lc)))

    if args.write_golden:
        def pack(seqs, dt=np.uint8):
            offs = np.zeros(len(seqs) + 1, np.int64)
            offs[1:] = np.cumsum([len(s) for s in seqs])
            return (np.concatenate(seqs).astype(dt) if seqs else np.zeros(0, dt)), offs

        qs, qo = pack([g[0] for g in gold])
        ts, to = pack([g[1] for g in gold])
        js, _ = pack([g[10] if g[10] is not None else np.zeros(len(g[1]), np.uint8) for g in gold])
        params = np.array([[g[3], g[4], g[5], g[6], g[7], g[8], g[9], int(g[10] is not None)] for g in gold], np.int32)
        mats = np.array([g[2] for g in gold], np.int8)
        cg, co = pack([g[11]["cigar"].view(np.uint8) for g in gold])
        scal = np.array([[g[11][f] for f in EZ_KEYS] for g in gold], np.int64)
        np.savez_compressed(os.path.join(GOLDEN, "ksw2_exts2.npz"), q=qs, qo=qo, t=ts, to=to, junc=js, params=params, mat=mats,
                            cigar_bytes=cg, cigar_off=co, scalars=scal)
        if gold_lc:
            gdo.save_lchain_golden(os.path.join(GOLDEN, "lchain_dp.npz"), gold_lc)
        print("golden vectors written to", GOLDEN)
    return 1 if (bad or bad_lc) else 0


if __name__ == "__main__":
    sys.exit(main())
