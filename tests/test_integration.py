"""Boundary proof (INTEGRATION.md level B1): the REFERENCE itself, with this repo's binding compiled in
(tests/integration/gdiet_hip_glue.c + the two call sites oracle/build_ref_hip.py inserts), linked against libgdiet_hip.so with
-Wl,--no-undefined -> oracle/_ref/gdiet_{lr,sr}_hip.

CPU box: the tree compiles and links; without GDIET_HIP the binary is GDiet_avx (golden SAM); with GDIET_HIP=1 and no GPU it fails
loudly (no CPU fallback).  GPU box: with GDIET_HIP=1 the reference's own main(), FASTQ reader, index builder and mm_write_sam3 run
around this library's per-read path and print the golden SAM byte for byte."""
import gzip
import os
import subprocess

import pytest

from conftest import ROOT
from fixture_io import PAF_KINDS, SETS, cmd_of, golden_paf, golden_sam, paf_cmd_of, reads_of

HIP_BIN = {v: os.path.join(ROOT, "oracle", "_ref", "gdiet_%s_hip" % v) for v in ("lr", "sr")}


def _inputs(kind, tmp_path):
    d, stem, _ = SETS[kind]
    ref_fa, fq = str(tmp_path / "ref.fa"), str(tmp_path / "reads.fq")
    with gzip.open(os.path.join(d, "ref.fa.gz"), "rb") as src, open(ref_fa, "wb") as dst:
        dst.write(src.read())
    with open(fq, "w") as f:
        for name, seq, qual in reads_of(kind):
            f.write("@%s\n%s\n+\n%s\n" % (name, seq, qual))
    return ref_fa, fq


def _body(stdout):
    return [l for l in stdout.rstrip("\n").split("\n") if not l.startswith("@")]


def test_reference_compiles_and_links_against_the_library(tmp_path):
    if not os.path.isdir("/root/reference/GDiet-LongReads"):
        pytest.skip("no /root/reference on this machine (the binaries travel prebuilt)")
    r = subprocess.run(["python3", os.path.join(ROOT, "oracle", "build_ref_hip.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    for v, exe in HIP_BIN.items():
        assert os.path.exists(exe)
        und = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
        for sym in ("gdiet_hip_init", "gdiet_hip_index_import", "gdiet_hip_map_batch", "gdiet_hip_free_regs"):
            assert sym in und, (v, sym)  # bound to the shared library, not stubbed
    # without GDIET_HIP the patched binary is GDiet_avx
    ref_fa, fq = _inputs("hifi", tmp_path)
    env = {k: v for k, v in os.environ.items() if k != "GDIET_HIP"}
    out = subprocess.run([HIP_BIN["lr"], "-t", "4"] + cmd_of("hifi") + [ref_fa, fq], capture_output=True, text=True, env=env, check=True)
    assert _body(out.stdout) == golden_sam("hifi")


def test_glue_fails_loudly_without_a_gpu(tmp_path):
    import torch
    if not os.path.exists(HIP_BIN["lr"]):
        pytest.skip("oracle/_ref/gdiet_lr_hip not built")
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    ref_fa, fq = _inputs("hifi", tmp_path)
    r = subprocess.run([HIP_BIN["lr"], "-t", "2"] + cmd_of("hifi") + [ref_fa, fq], capture_output=True, text=True, env=dict(os.environ, GDIET_HIP="1"))
    assert r.returncode != 0 and "[gdiet_hip]" in r.stderr  # no CPU fallback once the GPU path was asked for
    assert not [l for l in _body(r.stdout) if l]  # ... and not a single record was printed


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["hifi", "hifi_sv", "ont_sv", "hifi_edge", "sr", "sr_var", "sr_edge", "hifi_rep", "ont_rep", "sr_rep", "sr_rep_f60"])
def test_patched_reference_binary_prints_the_golden_sam(kind, tmp_path):
    variant = "sr" if SETS[kind][2] == "sr" else "lr"
    exe = HIP_BIN[variant]
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/gdiet_%s_hip not built (needs /root/reference at build time)" % variant)
    ref_fa, fq = _inputs(kind, tmp_path)
    r = subprocess.run([exe, "-t", "4"] + cmd_of(kind) + [ref_fa, fq], capture_output=True, text=True, env=dict(os.environ, GDIET_HIP="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got, want = _body(r.stdout), golden_sam(kind)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a == b, (a[:300], b[:300])


@pytest.mark.gpu
@pytest.mark.parametrize("kind", list(PAF_KINDS))
def test_patched_reference_binary_prints_the_golden_paf(kind, tmp_path):
    """the same binary in PAF mode (-x, -c, --paf-no-hit): the records this library returns through the REFERENCE's mm_write_paf3"""
    variant = "sr" if SETS[kind][2] == "sr" else "lr"
    exe = HIP_BIN[variant]
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/gdiet_%s_hip not built (needs /root/reference at build time)" % variant)
    ref_fa, fq = _inputs(kind, tmp_path)
    r = subprocess.run([exe, "-t", "4"] + paf_cmd_of(kind) + [ref_fa, fq], capture_output=True, text=True, env=dict(os.environ, GDIET_HIP="1"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.rstrip("\n").split("\n") == golden_paf(kind)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["hifi_sv", "sr"])
def test_patched_reference_binary_fans_out_over_two_contexts(kind, tmp_path):
    """GDIET_HIP_DEVICES=0,0: the reference's one-process CLI with TWO contexts (here both on the test box's one device), every
    mini-batch cut into two read ranges by gdiet_hip_map_batch_multi -- the golden SAM, and [PROFILING] lines fed from the library's
    stage clocks (LR/profile.h:10-21)"""
    variant = "sr" if SETS[kind][2] == "sr" else "lr"
    exe = HIP_BIN[variant]
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/gdiet_%s_hip not built (needs /root/reference at build time)" % variant)
    ref_fa, fq = _inputs(kind, tmp_path)
    r = subprocess.run([exe, "-t", "4"] + cmd_of(kind) + [ref_fa, fq], capture_output=True, text=True,
                       env=dict(os.environ, GDIET_HIP="1", GDIET_HIP_DEVICES="0,0"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert _body(r.stdout) == golden_sam(kind)
    prof = {l.split(":")[0]: int(l.split(":")[1].split()[0]) for l in r.stderr.split("\n") if l.startswith("[PROFILING]")}
    assert prof["[PROFILING] seeding time"] > 0 and prof["[PROFILING] voting time"] > 0 and prof["[PROFILING] sequence alignment time"] > 0


@pytest.mark.gpu
def test_patched_reference_binary_survives_a_degenerate_box(tmp_path):
    """the glue reports a read the library gave up on and goes on: every other read keeps its golden records (GDIET_FAULT_BOX)"""
    exe = HIP_BIN["lr"]
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/gdiet_lr_hip not built")
    ref_fa, fq = _inputs("hifi", tmp_path)
    r = subprocess.run([exe, "-t", "4"] + cmd_of("hifi") + [ref_fa, fq], capture_output=True, text=True,
                       env=dict(os.environ, GDIET_HIP="1", GDIET_FAULT_BOX="3"), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    victim = reads_of("hifi")[3][0]
    got = _body(r.stdout)
    assert [l for l in got if l.split("\t")[0] != victim] == [l for l in golden_sam("hifi") if l.split("\t")[0] != victim]
    assert [l.split("\t")[1] for l in got if l.split("\t")[0] == victim] == ["4"]
    assert "degenerate DP box" in r.stderr
