#!/usr/bin/env python3
"""Test infrastructure (boundary proof): builds THE REFERENCE with this repo's binding compiled in.

    python oracle/build_ref_hip.py          -> oracle/_ref/gdiet_lr_hip  (and gdiet_sr_hip)

The reference's sources are copied to a temporary directory (never into the repo), two call sites are inserted (the edits below:
the whole reference-side change of INTEGRATION.md level B1), tests/integration/gdiet_hip_glue.c is compiled with the reference's own
headers, and everything is linked against genome-on-diet_amd/libgdiet_hip.so with -Wl,--no-undefined.  Compile flags as in
oracle/Makefile.ref (= the reference's Makefile for GDiet_avx).  With GDIET_HIP=1 the binary maps through the library (needs
an MI355X); without it, it is GDiet_avx unchanged."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("REF", "/root/reference")
OUT = os.path.join(ROOT, "oracle", "_ref")
LIBDIR = os.path.join(ROOT, "genome-on-diet_amd")
GLUE = os.path.join(ROOT, "tests", "integration", "gdiet_hip_glue.c")
DECL = ("int gdiet_glue_enabled(void); void gdiet_glue_index(const mm_idx_t *mi, const mm_mapopt_t *opt); void gdiet_glue_close(void);\n"
        "int gdiet_glue_map_step(int n_frag, const int *seg_off, const int *n_seg, const mm_bseq1_t *seq, int *n_reg, mm_reg1_t **reg, const mm_mapopt_t *opt);\n")

# (file, anchor regex, replacement): the complete reference-side change
EDITS = [
    # step 1 of worker_pipeline (LR/map.c:2132-2137, SR/map.c:1202-1207): the mini-batch goes to the library instead of kt_for(worker_for)
    ("map.c", r"kt_for\(p->n_threads, worker_for, in, \(\(step_t \*\)in\)->n_frag\);",
     "{ step_t *gs_ = (step_t *)in; if (!gdiet_glue_enabled() || gdiet_glue_map_step(gs_->n_frag, gs_->seg_off, gs_->n_seg, gs_->seq, gs_->n_reg, gs_->reg, p->opt)) "
     "kt_for(p->n_threads, worker_for, in, gs_->n_frag); }"),
    # after mm_mapopt_update (LR/main.c:643): hand the freshly built / loaded index to the library
    ("main.c", r"if \(argc != o\.ind \+ 1\) mm_mapopt_update\(&opt, mi\);",
     "if (argc != o.ind + 1) mm_mapopt_update(&opt, mi);\n\t\tif (gdiet_glue_enabled() && argc != o.ind + 1) gdiet_glue_index(mi, &opt);"),
]
EDITS.append(
    # before the [PROFILING] lines (LR/main.c:685): contexts and device indexes released, unmapped-by-failure reads reported
    ("main.c", r"print_profile\(\);", "if (gdiet_glue_enabled()) gdiet_glue_close();\n\tprint_profile();"))
COMMON = "kthread kalloc misc bseq sdust options index lchain align hit seed format pe esterr splitidx profile".split()


def build(variant):
    src = os.path.join(REF, "GDiet-LongReads" if variant == "lr" else "GDiet-ShortReads")
    if not os.path.isdir(src):
        sys.exit("no reference at " + src)
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        for f in os.listdir(src):
            if f.endswith((".c", ".h")):
                shutil.copy(os.path.join(src, f), d)
        for fn, pat, rep in EDITS:
            p = os.path.join(d, fn)
            text = open(p).read()
            new, n = re.subn(pat, lambda m: rep, text)
            if n != 1:
                sys.exit("%s: the anchor %r matched %d times" % (fn, pat, n))
            # the declarations go after the last #include of the file
            inc = list(re.finditer(r"^#include .*$", new, re.M))[-1]
            new = new[:inc.end()] + "\n" + ("#include \"bseq.h\"\n" if fn == "map.c" else "") + DECL + new[inc.end():]
            open(p, "w").write(new)
        cc = os.environ.get("CC", "gcc")
        base = [cc, "-c", "-g", "-O2", "-fPIC", "-w", "-DHAVE_KALLOC", "-DPROFILE"]
        objs = []

        def comp(name, srcfile, extra):
            o = os.path.join(d, name + ".o")
            subprocess.check_call(base + extra + [os.path.join(d, srcfile), "-o", o])
            objs.append(o)
        for c in COMMON + ["main"]:
            comp(c, c + ".c", [])
        comp("ksw2_ll_sse", "ksw2_ll_sse.c", ["-msse2"])
        for k in ("ksw2_extz2", "ksw2_extd2", "ksw2_exts2", "exact_match"):
            comp(k + "_sse41", k + "_sse.c", ["-msse4.1", "-DKSW_CPU_DISPATCH"])
            comp(k + "_sse2", k + "_sse.c", ["-msse2", "-mno-sse4.1", "-DKSW_CPU_DISPATCH", "-DKSW_SSE2_ONLY"])
        comp("ksw2_dispatch", "ksw2_dispatch.c", ["-msse4.1", "-DKSW_CPU_DISPATCH"])
        comp("sketch_avx", "sketch.c", ["-mavx512dq"])
        comp("map_avx", "map.c", ["-mavx512bw"])
        comp("ksw2_extd2_avx", "ksw2_extd2_avx.c", ["-mavx512bw"])
        shutil.copy(GLUE, os.path.join(d, "gdiet_hip_glue.c"))
        comp("gdiet_hip_glue", "gdiet_hip_glue.c", ["-I", os.path.join(ROOT, "include")] + (["-DGDIET_SHORTREADS"] if variant == "sr" else []))
        exe = os.path.join(OUT, "gdiet_%s_hip" % variant)
        subprocess.check_call([cc, "-g", "-O2"] + objs + ["-o", exe, "-L", LIBDIR, "-lgdiet_hip", "-Wl,--no-undefined",
                                                         "-Wl,-rpath,$ORIGIN/../../genome-on-diet_amd", "-Wl,-rpath-link," + os.environ.get("ROCM_LIB", "/opt/rocm/lib"),
                                                         "-lm", "-lz", "-lpthread"])
    return exe


if __name__ == "__main__":
    for v in (sys.argv[1:] or ["lr", "sr"]):
        print(build(v))
