"""Compile the HIP C-ABI library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgdiet_hip.so")
SRC = os.path.join(HERE, "csrc", "gdiet_hip.hip")


def _newest_source():
    m = 0.0
    for root in (os.path.join(HERE, "csrc"), os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def build_hip(force=False, verbose=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # max-ilp scheduling: the DP row loop is one long block of dependent packed-16 operations; the default scheduler leaves 40 hazard
    # s_nop in it (a VALU instruction reading the result of the VOP3P instruction right before it), this one 8: -1 % kernel time
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-o", LIB, SRC, "-lz"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
