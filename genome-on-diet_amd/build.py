"""Compile the HIP C-ABI library in-tree for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import re
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "libgdiet_hip.so")
SRC = os.path.join(HERE, "csrc", "gdiet_hip.hip")
RES = os.path.join(HERE, "build_resources.txt")  # the compiler's per-kernel resource remarks of the last build

# Register budgets the design depends on (checked against the compiler's own resource report after every build):
#   * the 64-lane DP kernel: 96 VGPRs = 5 wavefronts per SIMD, and nothing of its row loop in scratch memory;
#   * the kernels that must be able to start BESIDE a full house of those DP wavefronts (5 x 96 of a SIMD's 512 registers are taken):
#     at most 32 VGPRs (map_kernels.hip.h: map_post_wave_kernel).
BUDGET = {"_Z21ksw_extd2_wave_kernelILi64ELi0ELb1E": dict(vgprs=96, scratch=0),
          "_Z21ksw_extd2_pipe_kernelILb1E": dict(vgprs=128, scratch=320),  # four wavefronts per SIMD; the spills sit in the once-per-alignment staging code, none in the steps
          "_Z20map_post_wave_kernel": dict(vgprs=32, scratch=0),
          "_Z21map_pack_cigar_kernel": dict(vgprs=32, scratch=0)}


def _newest_source():
    m = 0.0
    for root in (os.path.join(HERE, "csrc"), os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def check_resources(text):
    """{kernel prefix: (VGPRs, scratch bytes per lane)} for the kernels in BUDGET; raises if one is over its budget"""
    found = {}
    blocks = re.split(r"remark: Function Name: ", text)
    for b in blocks[1:]:
        name = b.split()[0]
        for prefix, lim in BUDGET.items():
            if name.startswith(prefix):
                v = re.search(r"remark:\s+VGPRs: (\d+)", b)
                sc = re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b)
                if v and sc:
                    found[prefix] = (int(v.group(1)), int(sc.group(1)))
                    if int(v.group(1)) > lim["vgprs"] or int(sc.group(1)) > lim["scratch"]:
                        raise RuntimeError("%s: %d VGPRs / %d B scratch per lane, budget %d / %d (genome-on-diet_amd/build.py)"
                                           % (name, int(v.group(1)), int(sc.group(1)), lim["vgprs"], lim["scratch"]))
    missing = [p for p in BUDGET if p not in found]
    if missing:
        raise RuntimeError("no resource report for %s" % missing)
    return found


def build_hip(force=False, verbose=False):
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= _newest_source():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # max-ilp scheduling: the DP row loop is one long block of dependent packed-16 operations; the default scheduler leaves 40 hazard
    # s_nop in it (a VALU instruction reading the result of the VOP3P instruction right before it), this one 8: -1 % kernel time
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", "-mllvm", "-amdgpu-sched-strategy=max-ilp",
           "-Rpass-analysis=kernel-resource-usage", "-o", LIB, SRC, "-lz"]
    if verbose:
        print(" ".join(cmd))
    r = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
    remarks = "\n".join(l for l in r.stderr.split("\n") if "-Rpass-analysis=kernel-resource-usage" in l)
    other = "\n".join(l for l in r.stderr.split("\n") if "-Rpass-analysis=kernel-resource-usage" not in l)
    if r.returncode:
        raise RuntimeError("hipcc failed:\n" + other[-4000:])
    open(RES, "w").write(remarks)
    found = check_resources(remarks)
    if verbose:
        print("register budgets:", found)
    return LIB


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
