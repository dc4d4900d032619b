#!/usr/bin/env python3
"""Test infrastructure: (re)makes the committed mapping fixtures under tests/golden/ with THE REFERENCE ITSELF
(oracle/_ref/gdiet_{lr,sr}_avx = GDiet_avx compiled from /root/reference by oracle/Makefile.ref).

    python oracle/make_golden.py            # check: every committed golden equals what the reference prints today
    python oracle/make_golden.py --write    # rewrite the goldens (and make the *_sv read sets if they are missing)
    python oracle/make_golden.py --only hifi_sv,ont_sv [--write]

What it makes, per kind of tests/fixture_io.py::SETS:
  <stem>.golden.sam.gz   the SAM body (no @ lines) of `gdiet_*_avx -t 4 <kind's .cmd> ref.fa reads.fq`
  <stem>.golden.paf.gz   (kinds in PAF_KINDS) the PAF lines of the same reads: `-x` instead of `-ax`, plus `-c --paf-no-hit`
  <stem>.trace.gz        (kinds in TRACED) the --print-seeds stage trace of the same run (LR/map.c:1328-1338,1447-1459,
                         1592-1602,1670-1675,1808-1810,1858-1863), reduced to the lines the stage test compares
  fastx/*.expected.json  what the reference's parser returns for the awkward FASTA/FASTQ files of tests/fastx_inputs.py
Read sets: hifi_sv.fq / ont_sv.fq come from tools/synth.py (--kind hifi_sv / ont_sv; seeds below); the older read sets
were made by tools/synth.py / tools/synth_sr_var.py and hand-written edge.fq files and are only read here.
Nothing of the product is involved: tests never write into tests/golden, this script is the only writer."""
import argparse
import gzip
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fixture_io import PAF_KINDS, SETS, TRACED, TRACE_PREFIXES, cmd_of, paf_cmd_of, reads_of  # noqa: E402

REF = {"lr": os.path.join(ROOT, "oracle", "_ref", "gdiet_lr_avx"), "sr": os.path.join(ROOT, "oracle", "_ref", "gdiet_sr_avx")}
SV_SETS = {"hifi_sv": dict(n=150, seed=11, mean_len=9000), "ont_sv": dict(n=60, seed=12, mean_len=14000)}


def _gunzip_to(src, dst):
    with gzip.open(src, "rb") as f, open(dst, "wb") as g:
        g.write(f.read())


def _write_gz(path, text):
    # mtime 0: identical content gives an identical file
    with open(path, "wb") as raw, gzip.GzipFile(filename="", mode="wb", fileobj=raw, mtime=0, compresslevel=9) as f:
        f.write(text.encode())


def make_sv_reads(kind, ref_fa):
    d, stem, _ = SETS[kind]
    out = os.path.join(d, stem + ".fq.gz")
    if os.path.exists(out):
        return
    p = SV_SETS[kind]
    with tempfile.TemporaryDirectory() as t:
        fq = os.path.join(t, "r.fq")
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "synth.py"), "reads", fq, "--ref", ref_fa, "--kind", kind,
                               "--n", str(p["n"]), "--seed", str(p["seed"]), "--mean-len", str(p["mean_len"])])
        _write_gz(out, open(fq).read())
    print("made", out)


def reference_run(kind, tmp):
    """(SAM body, reduced trace or None) of the reference on the kind's read set"""
    d, stem, _ = SETS[kind]
    variant = os.path.basename(d)
    ref_fa = os.path.join(tmp, variant + "_ref.fa")
    if not os.path.exists(ref_fa):
        _gunzip_to(os.path.join(d, "ref.fa.gz"), ref_fa)
    if kind in SV_SETS:
        make_sv_reads(kind, ref_fa)
    fq = os.path.join(tmp, kind + ".fq")
    with open(fq, "w") as f:
        for name, seq, qual in reads_of(kind):
            f.write("@%s\n%s\n+\n%s\n" % (name, seq, qual))
    run = subprocess.run([REF[variant], "-t", "4"] + cmd_of(kind) + [ref_fa, fq], capture_output=True, text=True, check=True)
    sam = "".join(l + "\n" for l in run.stdout.split("\n") if l and not l.startswith("@"))
    trace = None
    if kind in TRACED:
        run = subprocess.run([REF[variant]] + cmd_of(kind) + ["--print-seeds", ref_fa, fq], capture_output=True, text=True, check=True)
        assert "".join(l + "\n" for l in run.stdout.split("\n") if l and not l.startswith("@")) == sam
        trace = "".join(l + "\n" for l in run.stderr.split("\n") if l.startswith(TRACE_PREFIXES))
    paf = None
    if kind in PAF_KINDS:
        run = subprocess.run([REF[variant], "-t", "4"] + paf_cmd_of(kind) + [ref_fa, fq], capture_output=True, text=True, check=True)
        paf = "".join(l + "\n" for l in run.stdout.split("\n") if l)
    return sam, trace, paf


def fastx_expected(tmp):
    """{file name: json text}: the reference parser's view of tests/fastx_inputs.py's files (unmapped SAM records)"""
    import numpy as np
    from fastx_inputs import awkward_inputs, reference_view
    out = {}
    for name, data in awkward_inputs(np.random.default_rng(31)).items():
        path = os.path.join(tmp, name)
        with open(path, "wb") as f:
            f.write(data)
        for with_comment in (False, True):
            rows = reference_view(REF["lr"], path, tmp, with_comment)
            out["%s.%s.expected.json" % (name, "y" if with_comment else "n")] = json.dumps(
                [[x.decode("latin1") if x is not None else None for x in row] for row in rows])
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--write", action="store_true")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    for b in REF.values():
        if not os.path.exists(b):
            sys.exit("oracle/_ref is not built (make -f oracle/Makefile.ref needs /root/reference)")
    kinds = [k for k in a.only.split(",") if k] or list(SETS) + ["fastx"]
    bad = 0
    with tempfile.TemporaryDirectory() as tmp:
        for kind in kinds:
            if kind == "fastx":
                gold = os.path.join(ROOT, "tests", "golden", "fastx")
                for fn, text in fastx_expected(tmp).items():
                    p = os.path.join(gold, fn)
                    same = os.path.exists(p) and open(p).read() == text
                    if not same and a.write:
                        open(p, "w").write(text)
                    bad += not same
                    print("%-40s %s" % ("fastx/" + fn, "ok" if same else ("WRITTEN" if a.write else "DIFFERS")))
                continue
            d, stem, _ = SETS[kind]
            sam, trace, paf = reference_run(kind, tmp)
            for suffix, text in ((".golden.sam.gz", sam), (".trace.gz", trace), (".golden.paf.gz", paf)):
                if text is None:
                    continue
                p = os.path.join(d, stem + suffix)
                same = os.path.exists(p) and gzip.open(p, "rt").read() == text
                if not same and a.write:
                    _write_gz(p, text)
                bad += not same
                print("%-40s %s" % (os.path.relpath(p, ROOT), "ok" if same else ("WRITTEN" if a.write else "DIFFERS")))
            if kind in TRACED:
                flags = {}
                for l in sam.split("\n"):
                    if l:
                        flags[l.split("\t")[1]] = flags.get(l.split("\t")[1], 0) + 1
                print("    flags", dict(sorted(flags.items(), key=lambda kv: int(kv[0]))), "| CONQ events", trace.count("CONQ["))
    sys.exit(0 if a.write or not bad else 1)


if __name__ == "__main__":
    main()
