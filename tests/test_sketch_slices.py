"""CPU: exact slicing of the winnowing automaton (the property the wave-parallel seed kernel relies on):
concatenated gd_sketch_slice() outputs == sequential gd_sketch_core() output, incl. Ns, tandem repeats, homopolymers."""
import os
import subprocess

from conftest import ROOT


def test_slices_equal_sequential_sketch(tmp_path):
    exe = str(tmp_path / "slice_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-w", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"),
                           os.path.join(ROOT, "tests", "emul", "sketch_slice_test.cpp"), "-o", exe])
    for seed in ("1", "2"):
        out = subprocess.run([exe, seed, "400"], capture_output=True, text=True)
        assert out.returncode == 0 and "mismatches=0" in out.stdout, out.stdout + out.stderr
