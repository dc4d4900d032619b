#!/usr/bin/env python3
"""Pin the ksw2 oracle (oracle/gdo_ksw2.c) against the reference itself and (re)generate the committed golden
vectors tests/golden/ksw2_extd2.npz, ksw2_extz2.npz, exact_match.npz.

Runs only where oracle/_ref has been built from /root/reference (this container).  The golden vectors hold
inputs + the REFERENCE's outputs (ksw_extd2_sse / ksw_extz2_sse / exact_match_sse called through ctypes), so the
GPU box and later rounds can re-check both the oracle and the HIP kernels without the reference.

    python oracle/pin_ksw2.py --fuzz 3000          # differential fuzz only (oracle vs SSE vs AVX-512)
    python oracle/pin_ksw2.py --write-golden       # also rewrite tests/golden/*.npz
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gdo  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(gdo.HERE), "tests", "golden")


def cases(rng, n, heavy=True):
    """yield (tag, query, target, preset, w, zdrop, end_bonus, flag)"""
    for i in range(n):
        kind = i % 12
        preset = ("sr", "hifi", "ont")[i % 3]
        flag = gdo.EZ_APPROX_MAX
        zdrop, end_bonus = -1, 0
        if kind == 0:  # SR shape: 150x150, w=150
            q, t = gdo.make_pair(rng, 150, 0.01, 0.0005, 0.0005)
            preset, w = "sr", 150
        elif kind == 1:  # long-ish, moderate band
            tl = int(rng.integers(300, 2500 if heavy else 900))
            q, t = gdo.make_pair(rng, tl, 0.002, 0.001, 0.001)
            w = int(rng.integers(20, 400))
        elif kind == 2:  # band-edge stress: one indel of 0.4-0.6 w
            tl = int(rng.integers(400, 2000 if heavy else 900))
            w = int(rng.integers(40, 300))
            sz = int(w * rng.uniform(0.4, 0.62)) * (1 if rng.random() < 0.5 else -1)
            q, t = gdo.make_pair(rng, tl, 0.01, 0.002, 0.002, big_indel=sz)
        elif kind == 3:  # Ns in both
            tl = int(rng.integers(100, 1200))
            q, t = gdo.make_pair(rng, tl, 0.02, 0.005, 0.005, n_frac=0.02)
            w = int(rng.integers(10, 300))
        elif kind == 4:  # tiny
            tl = int(rng.integers(1, 40))
            q, t = gdo.make_pair(rng, tl, 0.05, 0.03, 0.03)
            w = int(rng.integers(0, 40))
        elif kind == 5:  # qlen != tlen, band may be exhausted (st > en)
            tl = int(rng.integers(50, 600))
            q, t = gdo.make_pair(rng, tl, 0.02, 0.01, 0.01, trim=int(rng.integers(0, tl // 2 + 1)))
            w = int(rng.integers(1, 120))
        elif kind == 6:  # unrelated sequences (path wanders to band edges)
            q = rng.integers(0, 4, size=int(rng.integers(20, 700)), dtype=np.uint8)
            t = rng.integers(0, 4, size=int(rng.integers(20, 700)), dtype=np.uint8)
            w = int(rng.integers(5, 200))
        elif kind == 7:  # w multiple of 16 and lengths multiple of 16 (alignment corner cases)
            tl = 16 * int(rng.integers(1, 60))
            q, t = gdo.make_pair(rng, tl, 0.01, 0.0, 0.0)
            w = 16 * int(rng.integers(1, 20))
        elif kind == 8:  # exact-max + z-drop / extension-only flags (oracle generality; not on the live path)
            tl = int(rng.integers(30, 500))
            q, t = gdo.make_pair(rng, tl, 0.05, 0.02, 0.02)
            w = int(rng.integers(10, 200))
            flag = int(rng.choice([0, gdo.EZ_EXTZ_ONLY, gdo.EZ_RIGHT, gdo.EZ_REV_CIGAR, gdo.EZ_SCORE_ONLY,
                                   gdo.EZ_APPROX_MAX | gdo.EZ_APPROX_DROP, gdo.EZ_APPROX_MAX | gdo.EZ_RIGHT,
                                   gdo.EZ_EXTZ_ONLY | gdo.EZ_RIGHT | gdo.EZ_REV_CIGAR, gdo.EZ_GENERIC_SC]))
            zdrop = int(rng.choice([-1, 20, 100, 400]))
            end_bonus = int(rng.choice([0, 5]))
        elif kind == 9:  # HiFi-like geometry at reduced length: w close to the length
            tl = int(rng.integers(800, 3000 if heavy else 1200))
            q, t = gdo.make_pair(rng, tl, 0.002, 0.001, 0.001)
            preset, w = "hifi", 1000
        elif kind == 10:  # ONT-like noise
            tl = int(rng.integers(300, 2000 if heavy else 900))
            q, t = gdo.make_pair(rng, tl, 0.03, 0.02, 0.02)
            preset, w = "ont", int(rng.integers(100, 1300))
        else:  # full matrix (w < 0)
            tl = int(rng.integers(10, 300))
            q, t = gdo.make_pair(rng, tl, 0.03, 0.02, 0.02)
            w = -1
        yield ("k%d" % kind, q, t, preset, w, zdrop, end_bonus, flag)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fuzz", type=int, default=1200)
    ap.add_argument("--seed", type=int, default=4)
    ap.add_argument("--write-golden", action="store_true")
    args = ap.parse_args()

    gdo.build_oracle()
    ora = gdo.load_oracle()
    ref = gdo.load_ref("lr_avx")
    rng = np.random.default_rng(args.seed)

    n_bad_sse = n_bad_avx = n_sse_vs_avx = 0
    gold = []
    gold_avx = []
    for idx, (tag, q, t, preset, w, zdrop, end_bonus, flag) in enumerate(cases(rng, args.fuzz)):
        a, b, go, ge, go2, ge2 = gdo.PRESETS[preset]
        mat = gdo.score_matrix(a, b)
        o = gdo.oracle_extd2(ora, q, t, mat, go, ge, go2, ge2, w, zdrop, end_bonus, flag)
        r = gdo.ref_extd2(ref, q, t, mat, go, ge, go2, ge2, w, zdrop, end_bonus, flag)
        keys = ("score", "zdropped", "max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "reach_end")
        if not gdo.same(o, r, keys):
            n_bad_sse += 1
            if n_bad_sse <= 5:
                print("MISMATCH oracle vs SSE", idx, tag, len(q), len(t), preset, w, flag, zdrop,
                      {k: (o[k], r[k]) for k in keys if o[k] != r[k]}, len(o["cigar"]), len(r["cigar"]))
        if flag == gdo.EZ_APPROX_MAX:  # the AVX-512 port is only used (and only valid) in the live-path mode
            # the parity target's kernel; the oracle reproduces its score table with GDO_EZ_AVX512_SC.  Every third
            # pair gets some query bytes 7 (= N of a reverse-complemented read, LR/map.c:1634), where SSE and AVX-512 differ.
            q7 = q.copy()
            if idx % 3 == 0 and len(q7) > 4:
                pos7 = rng.integers(0, len(q7), size=max(1, len(q7) // 100))
                q7[pos7] = 7
                run = int(rng.integers(0, len(q7) - 3))
                q7[run:run + int(rng.integers(1, 40))] = 7
            v = gdo.ref_extd2(ref, q7, t, mat, go, ge, go2, ge2, w, zdrop, end_bonus, flag, fn="ksw_extd2_avx512")
            o7 = gdo.oracle_extd2(ora, q7, t, mat, go, ge, go2, ge2, w, zdrop, end_bonus, flag | gdo.EZ_AVX512_SC)
            if not gdo.same(o7, v):
                n_bad_avx += 1
                if n_bad_avx <= 5:
                    print("MISMATCH oracle(AVX512_SC) vs ksw_extd2_avx512", idx, tag, len(q), len(t), preset, w, o7["score"], v["score"])
            r7 = gdo.ref_extd2(ref, q7, t, mat, go, ge, go2, ge2, w, zdrop, end_bonus, flag)
            if not gdo.same(r7, v):
                n_sse_vs_avx += 1
            if idx < 360:
                gold_avx.append((q7, t, preset, w, zdrop, end_bonus, flag, v))
        # extz2 (single affine) on the same pair
        oz = gdo.oracle_extz2(ora, q, t, mat, go, ge, w, zdrop, end_bonus, flag)
        rz = gdo.ref_extz2(ref, q, t, mat, go, ge, w, zdrop, end_bonus, flag)
        if not gdo.same(oz, rz, keys):
            n_bad_sse += 1
            if n_bad_sse <= 5:
                print("MISMATCH extz2 oracle vs SSE", idx, tag, len(q), len(t), preset, w, flag,
                      {k: (oz[k], rz[k]) for k in keys if oz[k] != rz[k]})
        if idx < 360:
            gold.append((q, t, preset, w, zdrop, end_bonus, flag, r, rz))
    # exact match
    n_bad_em = 0
    em = []
    for i in range(400):
        n = int(rng.integers(1, 300))
        t = rng.integers(0, 5, size=n, dtype=np.uint8)
        q = t.copy()
        if i % 3:
            q[int(rng.integers(0, n))] ^= 1
        mat = gdo.score_matrix(2, 8)
        if gdo.oracle_exact_match(ora, q, t) != gdo.ref_exact_match(ref, q, t, mat):
            n_bad_em += 1
        if i < 60:
            em.append((q, t, gdo.ref_exact_match(ref, q, t, mat)))
    # exact-maximum mode of ksw_extz2 (flag 0 / KSW_EZ_EXTZ_ONLY, with and without z-drop): what gdiet_hip_ksw_extz2_batch_ex answers
    n_bad_exact = 0
    gold_exact = []
    rng = np.random.default_rng(args.seed + 1000)  # (a stream of its own: these vectors do not depend on --fuzz)
    for i in range(1200 if args.fuzz >= 1000 else 300):
        preset = ("sr", "hifi", "ont")[i % 3]
        a, b, go, ge, _, _ = gdo.PRESETS[preset]
        mat = gdo.score_matrix(a, b)
        tlen = int(rng.integers(20, 420))
        t = rng.integers(0, 4, size=tlen, dtype=np.uint8)
        if i % 9 == 0:
            t[rng.integers(0, tlen, size=max(1, tlen // 50))] = 4
        q = t.copy()
        m = rng.random(tlen) < (0.02 if i % 2 else 0.08)
        q[m] = (q[m] + rng.integers(1, 4, size=int(m.sum()))) & 3
        q = q[rng.random(len(q)) >= 0.01]
        if i % 4 == 1 and len(q) > 60:  # a long indel or an unrelated tail: z-drop / a maximum inside the matrix
            cut = int(rng.integers(20, len(q) - 20))
            q = np.concatenate([q[:cut], rng.integers(0, 4, size=int(rng.integers(5, 80)), dtype=np.uint8), q[cut:]]) if i % 8 == 1 else \
                np.concatenate([q[:cut], rng.integers(0, 4, size=len(q) - cut, dtype=np.uint8)])
        if len(q) == 0:
            q = t[:1].copy()
        w = int(rng.integers(8, 200)) if i % 5 else -1
        zdrop = (-1, 20, 100, 400)[(i // 3) % 4]
        end_bonus = (0, 5)[(i // 12) % 2]
        flag = (0, gdo.EZ_EXTZ_ONLY)[(i // 24) % 2]
        oz = gdo.oracle_extz2(ora, q, t, mat, go, ge, w, zdrop, end_bonus, flag)
        rz = gdo.ref_extz2(ref, q, t, mat, go, ge, w, zdrop, end_bonus, flag)
        keys = ("score", "zdropped", "max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q", "reach_end")
        if not gdo.same(oz, rz, keys):
            n_bad_exact += 1
            if n_bad_exact <= 5:
                print("MISMATCH extz2 exact mode oracle vs SSE", i, len(q), len(t), preset, w, flag, zdrop, {k: (oz[k], rz[k]) for k in keys if oz[k] != rz[k]})
        if i < 300:
            gold_exact.append((q, t, preset, w, zdrop, end_bonus, flag, rz))
    n_bad_sse += n_bad_exact
    print("extz2 exact-maximum mode: pairs=%d oracle_vs_sse_mismatch=%d (z-dropped in %d of the %d golden pairs)"
          % (1200 if args.fuzz >= 1000 else 300, n_bad_exact, sum(1 for g in gold_exact if g[7]["zdropped"]), len(gold_exact)))
    print("pairs=%d oracle_vs_sse_mismatch=%d oracle_vs_avx512_mismatch=%d sse_vs_avx512_differ_on_byte7_inputs=%d exact_match_mismatch=%d"
          % (args.fuzz, n_bad_sse, n_bad_avx, n_sse_vs_avx, n_bad_em))

    if args.write_golden:
        os.makedirs(GOLDEN, exist_ok=True)

        def pack(seqs):
            offs = np.zeros(len(seqs) + 1, np.int64)
            offs[1:] = np.cumsum([len(s) for s in seqs])
            return (np.concatenate(seqs).astype(np.uint8) if seqs else np.zeros(0, np.uint8)), offs

        qs, qo = pack([g[0] for g in gold])
        ts, to = pack([g[1] for g in gold])
        params = np.array([[("sr", "hifi", "ont").index(g[2]), g[3], g[4], g[5], g[6]] for g in gold], np.int32)
        for name, k in (("ksw2_extd2", 7), ("ksw2_extz2", 8)):
            cg, co = pack([g[k]["cigar"].view(np.uint8) for g in gold])
            scal = np.array([[g[k][f] for f in ("score", "zdropped", "max", "max_q", "max_t", "mqe", "mqe_t", "mte",
                                                  "mte_q", "reach_end")] for g in gold], np.int64)
            np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), q=qs, qo=qo, t=ts, to=to, params=params,
                                cigar_bytes=cg, cigar_off=co, scalars=scal)
        qs7, qo7 = pack([g[0] for g in gold_avx])
        ts7, to7 = pack([g[1] for g in gold_avx])
        params7 = np.array([[("sr", "hifi", "ont").index(g[2]), g[3], g[4], g[5], g[6]] for g in gold_avx], np.int32)
        cg7, co7 = pack([g[7]["cigar"].view(np.uint8) for g in gold_avx])
        scal7 = np.array([[g[7][f] for f in ("score", "zdropped", "max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q",
                                             "reach_end")] for g in gold_avx], np.int64)
        np.savez_compressed(os.path.join(GOLDEN, "ksw2_extd2_avx512.npz"), q=qs7, qo=qo7, t=ts7, to=to7, params=params7,
                            cigar_bytes=cg7, cigar_off=co7, scalars=scal7)
        qsx, qox = pack([g[0] for g in gold_exact])
        tsx, tox = pack([g[1] for g in gold_exact])
        paramsx = np.array([[("sr", "hifi", "ont").index(g[2]), g[3], g[4], g[5], g[6]] for g in gold_exact], np.int32)
        cgx, cox = pack([g[7]["cigar"].view(np.uint8) for g in gold_exact])
        scalx = np.array([[g[7][f] for f in ("score", "zdropped", "max", "max_q", "max_t", "mqe", "mqe_t", "mte", "mte_q",
                                             "reach_end")] for g in gold_exact], np.int64)
        np.savez_compressed(os.path.join(GOLDEN, "ksw2_extz2_exact.npz"), q=qsx, qo=qox, t=tsx, to=tox, params=paramsx,
                            cigar_bytes=cgx, cigar_off=cox, scalars=scalx)
        eq, eqo = pack([e[0] for e in em])
        et, eto = pack([e[1] for e in em])
        np.savez_compressed(os.path.join(GOLDEN, "exact_match.npz"), q=eq, qo=eqo, t=et, to=eto,
                            expect=np.array([e[2] for e in em], np.int32))
        print("golden vectors written to", GOLDEN)
    return 1 if (n_bad_sse or n_bad_em) else 0


if __name__ == "__main__":
    sys.exit(main())
