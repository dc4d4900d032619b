"""CPU: the host worker pool shared by the batches in flight (csrc/host_pool.h), hammered by several caller threads, plain and
under ThreadSanitizer."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.mark.parametrize("tsan", [False, True])
def test_shared_pool_runs_every_item_once(tmp_path, tsan):
    exe = str(tmp_path / "pool_test")
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"),
           os.path.join(ROOT, "tests", "emul", "pool_test.cpp"), "-o", exe]
    if tsan:
        cmd.insert(1, "-fsanitize=thread")
    subprocess.check_call(cmd)
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1")
    r = subprocess.run([exe, "4", "60" if tsan else "300", "6"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "bad 0" in r.stdout
    assert "WARNING: ThreadSanitizer" not in r.stderr
