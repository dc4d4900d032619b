"""CPU: the vectorised ASCII -> nt4 encoder of gdiet_hip_batch_upload (csrc/nt4_encode.h) against the byte-wise table
(seq_nt4_table, LR/sketch.c:11-18) on every byte value at every alignment and on random buffers."""
import os
import subprocess

from conftest import ROOT


def test_vector_encoder_equals_the_table(tmp_path):
    exe = str(tmp_path / "nt4_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"),
                           os.path.join(ROOT, "tests", "emul", "nt4_test.cpp"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " bad 0" in r.stdout
