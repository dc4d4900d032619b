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
                         (kinds in SD_DIGESTED: every read's SD lines replaced by their count and sha1)
  fastx/*.expected.json  what the reference's parser returns for the awkward FASTA/FASTQ files of tests/fastx_inputs.py
  rep/mmi.sha256.json    size and sha256 of the index files `gdiet_*_avx -d` writes for the repeat-rich reference (k19w19, k15w10, k21w11)
Repeat-rich sets (tests/golden/rep/, kinds *_rep): reference and reads come from tools/synth_rep.py; for each of them the counts of the
branches they exist for (the product's host emulator, --stats) are printed next to the file names.
Read sets: hifi_sv.fq / ont_sv.fq come from tools/synth.py (--kind hifi_sv / ont_sv; seeds below); the older read sets
were made by tools/synth.py / tools/synth_sr_var.py and hand-written edge.fq files and are only read here.
Nothing of the product is involved in what is written: tests never write into tests/golden, this script is the only writer (the
--stats counts are printed, not stored)."""
import argparse
import gzip
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fixture_io import PAF_KINDS, REP, SD_DIGESTED, SETS, TRACED, TRACE_PREFIXES, cmd_of, digest_sd, paf_cmd_of, reads_of, variant_of  # noqa: E402

REF = {"lr": os.path.join(ROOT, "oracle", "_ref", "gdiet_lr_avx"), "sr": os.path.join(ROOT, "oracle", "_ref", "gdiet_sr_avx")}
SV_SETS = {"hifi_sv": dict(n=150, seed=11, mean_len=9000), "ont_sv": dict(n=60, seed=12, mean_len=14000)}
REP_SETS = {"hifi_rep": 120, "ont_rep": 60, "sr_rep": 3000}  # tools/synth_rep.py: reads per set (seeds are the tool's defaults)
# index files of the repeat-rich reference: (variant, the kind whose command line carries k / w / the pattern)
MMI_OF_REP = {"k19w19": ("lr", "hifi_rep"), "k15w10": ("lr", "ont_rep"), "k21w11": ("sr", "sr_rep")}


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


def make_rep_inputs():
    """tests/golden/rep/ref.fa.gz (+ ref.layout.json: where the repeats are) and the three read sets, if missing"""
    synth_rep = os.path.join(ROOT, "tools", "synth_rep.py")
    ref_gz, layout = os.path.join(REP, "ref.fa.gz"), os.path.join(REP, "ref.layout.json")
    with tempfile.TemporaryDirectory() as t:
        fa = os.path.join(t, "ref.fa")
        if not os.path.exists(ref_gz):
            subprocess.check_call([sys.executable, synth_rep, "ref", fa, "--layout", layout])
            _write_gz(ref_gz, open(fa).read())
            print("made", ref_gz)
        for kind, n in REP_SETS.items():
            out = os.path.join(REP, kind + ".fq.gz")
            if os.path.exists(out):
                continue
            if not os.path.exists(fa):
                _gunzip_to(ref_gz, fa)
            fq = os.path.join(t, kind + ".fq")
            subprocess.check_call([sys.executable, synth_rep, "reads", fq, "--ref", fa, "--layout", layout, "--kind", kind, "--n", str(n)])
            _write_gz(out, open(fq).read())
            print("made", out)


def rep_mmi_digests(tmp):
    """{"k19w19": {"size": .., "sha256": ..}, ..}: the index files `GDiet_avx -d` writes for the repeat-rich reference (3-5 MB each,
    thousands of multi-occurrence position lists); only their digests are committed (tests/golden/rep/mmi.sha256.json)"""
    import hashlib
    ref_fa = os.path.join(tmp, "rep_ref.fa")
    if not os.path.exists(ref_fa):
        _gunzip_to(os.path.join(REP, "ref.fa.gz"), ref_fa)
    out = {}
    for tag, (variant, kind) in MMI_OF_REP.items():
        mmi = os.path.join(tmp, tag + ".mmi")
        subprocess.run([REF[variant], "-t", "4"] + cmd_of(kind) + ["-d", mmi, ref_fa], capture_output=True, check=True)
        data = open(mmi, "rb").read()
        out[tag] = dict(size=len(data), sha256=hashlib.sha256(data).hexdigest())
    return out


def emulator_stats(kind, tmp, ref_fa, fq):
    """the [stats] lines of the product's host emulator (tests/emul/map_host_main.cpp --stats): how often each high-occurrence branch
    of the seeding stage fires on this set.  Printed for the record; tests/test_map_host.py asserts them."""
    exe = os.path.join(tmp, "map_host")
    if not os.path.exists(exe):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-w", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
                               os.path.join(ROOT, "tests", "emul", "map_host_main.cpp"), "-x", "c", os.path.join(ROOT, "oracle", "gdo_ksw2.c"), "-o", exe])
    err = subprocess.run([exe, "-t", "8"] + cmd_of(kind) + ["--stats", ref_fa, fq], capture_output=True, text=True, check=True).stderr
    return [l for l in err.split("\n") if l.startswith("[stats]")]


def reference_run(kind, tmp):
    """(SAM body, reduced trace or None) of the reference on the kind's read set"""
    d, stem, _ = SETS[kind]
    variant = variant_of(kind)
    ref_fa = os.path.join(tmp, os.path.basename(d) + "_ref.fa")
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
        lines = [l for l in run.stderr.split("\n") if l.startswith(TRACE_PREFIXES)]
        if kind in SD_DIGESTED:
            lines = digest_sd(lines)
        trace = "".join(l + "\n" for l in lines)
    paf = None
    if kind in PAF_KINDS:
        run = subprocess.run([REF[variant], "-t", "4"] + paf_cmd_of(kind) + [ref_fa, fq], capture_output=True, text=True, check=True)
        paf = "".join(l + "\n" for l in run.stdout.split("\n") if l)
    if kind in REP_SETS or kind == "sr_rep_f60":
        for l in emulator_stats(kind, tmp, ref_fa, fq):
            print("    " + l)
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
    kinds = [k for k in a.only.split(",") if k] or list(SETS) + ["fastx", "rep_mmi"]
    bad = 0
    if a.write:
        make_rep_inputs()
    with tempfile.TemporaryDirectory() as tmp:
        for kind in kinds:
            if kind == "rep_mmi":
                p = os.path.join(REP, "mmi.sha256.json")
                text = json.dumps(rep_mmi_digests(tmp), indent=1, sort_keys=True) + "\n"
                same = os.path.exists(p) and open(p).read() == text
                if not same and a.write:
                    open(p, "w").write(text)
                bad += not same
                print("%-40s %s" % (os.path.relpath(p, ROOT), "ok" if same else ("WRITTEN" if a.write else "DIFFERS")))
                continue
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
