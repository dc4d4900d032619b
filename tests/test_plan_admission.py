"""CPU: the O(1) admission test of the register-resident DP kernels against its block-by-block definition (tests/emul/plan_test.cpp).
The planner asks it for every alignment of a batch -- kernel choice and backtrace geometry depend on it -- so the two forms must agree on
every geometry, not only on the ones the presets produce."""
import os
import subprocess

from conftest import ROOT


def test_fast_admission_test_equals_its_definition(tmp_path):
    exe = str(tmp_path / "plan_test")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"), os.path.join(ROOT, "tests", "emul", "plan_test.cpp"), "-o", exe])
    r = subprocess.run([exe, "1200000"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    cases, admitted, differ = (int(x) for x in r.stdout.split())
    assert cases == 7200000 and differ == 0 and 0.2 * cases < admitted < 0.8 * cases  # (both outcomes well represented)
