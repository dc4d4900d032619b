#!/usr/bin/env python3
"""One-off soak comparison, larger than the committed fixtures (test infrastructure; not collected by pytest).

    python tests/soak.py make     # CPU box: synthesise read sets, run the reference (oracle/_ref/gdiet_*_avx) -> build_ub/soak/*.sam.gz
    python tests/soak.py check    # GPU box: map the same read sets through libgdiet_hip.so and compare every SAM line

Read sets: SV-style HiFi and ONT reads (tools/synth.py --kind *_sv: linked candidates, concatenate_cigars, supplementary records),
plain ONT reads long enough for the wide-band kernels (checkpointed form forced with GDIET_WIDE_CKPT=1), short reads of mixed lengths."""
import gzip
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
OUT = os.path.join(ROOT, "build_ub", "soak")
SETS = [  # name, variant, preset kind (fixture whose .cmd / reference is used), synth args, Mapper overrides
    ("hifi_sv", "lr", "hifi_sv", ["--kind", "hifi_sv", "--n", "1500", "--seed", "101", "--mean-len", "9000"], {}),
    ("ont_sv", "lr", "ont_sv", ["--kind", "ont_sv", "--n", "400", "--seed", "102", "--mean-len", "16000"], {"min_dp_max": 4000}),
    ("ont", "lr", "ont_sv", ["--kind", "ont", "--n", "150", "--seed", "103"], {"min_dp_max": 4000}),
    ("hifi", "lr", "hifi", ["--kind", "hifi", "--n", "1500", "--seed", "104"], {}),
    # short reads: 150 bases, a third of them cut to 36-149 (other geometries in the grouped short-alignment kernels)
    ("sr", "sr", "sr", ["--kind", "sr", "--n", "120000", "--seed", "105"], {}),
    ("sr_var", "sr", "sr_var", ["--kind", "sr", "--n", "60000", "--seed", "106"], None),  # (None: fixture_io.OVERRIDES of the kind)
]


def cut_some(fq, seed):
    import numpy as np
    rng = np.random.default_rng(seed)
    lines = open(fq).read().split("\n")
    for i in range(0, len(lines) - 3, 4):
        if rng.random() < 1 / 3:
            n = int(rng.integers(36, 150))
            lines[i + 1], lines[i + 3] = lines[i + 1][:n], lines[i + 3][:n]
    open(fq, "w").write("\n".join(lines))


def make():
    from fixture_io import SETS as FX, cmd_of
    os.makedirs(OUT, exist_ok=True)
    for name, variant, kind, synth, _ in SETS:
        d = FX[kind][0]
        if os.path.exists(os.path.join(OUT, name + ".sam.gz")) and os.path.exists(os.path.join(OUT, name + ".fq.gz")):
            continue
        ref_fa = os.path.join(OUT, variant + "_ref.fa")
        if not os.path.exists(ref_fa):
            with gzip.open(os.path.join(d, "ref.fa.gz"), "rb") as f, open(ref_fa, "wb") as g:
                g.write(f.read())
        fq = os.path.join(OUT, name + ".fq")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "synth.py"), "reads", fq, "--ref", ref_fa] + synth)
        if variant == "sr":
            cut_some(fq, int(synth[-1]))
        exe = os.path.join(ROOT, "oracle", "_ref", "gdiet_%s_avx" % variant)
        run = subprocess.run([exe, "-t", "8"] + cmd_of(kind) + [ref_fa, fq], capture_output=True, text=True, check=True)
        body = "".join(l + "\n" for l in run.stdout.split("\n") if l and not l.startswith("@"))
        with gzip.open(os.path.join(OUT, name + ".sam.gz"), "wt") as f:
            f.write(body)
        with open(fq, "rb") as f, gzip.open(fq + ".gz", "wb") as g:  # (only the .gz files travel to the GPU box)
            g.write(f.read())
        os.remove(fq)
        flags = {}
        for l in body.split("\n"):
            if l:
                flags[l.split("\t")[1]] = flags.get(l.split("\t")[1], 0) + 1
        print(name, "records", body.count("\n"), "flags", dict(sorted(flags.items(), key=lambda kv: int(kv[0]))))


def check():
    import torch  # noqa: F401
    from conftest import load_pkg
    from fixture_io import OVERRIDES, SETS as FX, read_fasta, read_fastq
    pkg = load_pkg()
    ctx = pkg.Context(0)
    bad = 0
    for name, variant, kind, _, over in SETS:
        d, _, preset = FX[kind]
        names, seqs = read_fasta(os.path.join(d, "ref.fa.gz"))
        reads = read_fastq(os.path.join(OUT, name + ".fq.gz"))
        want = gzip.open(os.path.join(OUT, name + ".sam.gz"), "rt").read()
        if over is None:
            over = OVERRIDES[kind]
        m = pkg.Mapper(ctx, names, seqs, preset=preset, **over)
        got = m.sam_batch(m.map([r[1] for r in reads]), reads)
        m.close()

        def norm(text):  # the reference's ms:i of reverse-strand records over reference Ns reads past its score matrix
            out = []
            for line in text.split("\n"):
                f = line.split("\t")
                if len(f) > 11 and int(f[1]) & 16 and "nn:i:0" not in f:
                    f = [x for x in f if not x.startswith("ms:i:")]
                out.append("\t".join(f))
            return out
        g, w = norm(got), norm(want)
        n_diff = sum(1 for a, b in zip(g, w) if a != b) + abs(len(g) - len(w))
        print(name, "reads", len(reads), "lines", len(w), "differing", n_diff, "kernel mask", ctx.last_kernel_mask())
        if n_diff:
            for a, b in zip(g, w):
                if a != b:
                    print("  got ", a[:200])
                    print("  want", b[:200])
                    break
        bad += n_diff
    ctx.close()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    {"make": make, "check": check}[sys.argv[1]]()
