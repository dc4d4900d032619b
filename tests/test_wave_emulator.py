"""CPU: the per-lane code of the register-resident HIP kernel (genome-on-diet_amd/csrc/ksw_wave_core.h), driven by a
host lock-step emulator, reproduces the oracle's score and CIGAR bit for bit on seeded random pairs."""
import os
import subprocess

import pytest

from conftest import ROOT


CORE = [(1, 64, 400), (2, 64, 400), (3, 16, 400), (4, 128, 60), (5, 10, 400), (6, 8, 400)]
SINGLE = [16, 64, 10]
CONE = [(4, 60, 480), (7, 40, 320), (8, 30, 64)]
RING96 = [(4, 60, 480), (7, 40, 200)]


def _all_commands():
    cmds = {("core",) + c: [str(c[0]), str(c[2]), str(c[1])] for c in CORE}
    cmds.update({("single", l): ["9", "300", str(l), "single"] for l in SINGLE})
    cmds.update({("cone",) + c: [str(c[0]), str(c[1]), "128", "ckpt", str(c[2])] for c in CONE})
    cmds.update({("ring96",) + c: [str(c[0]), str(c[1]), "96", "ckpt96", str(c[2])] for c in RING96})
    return cmds


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    """builds the emulator and starts EVERY run of this module at once, a few at a time side by side (each is one single-threaded process of
    10-40 s; run one after the other they were four of the CPU suite's eight minutes); a test then waits for its own run"""
    from concurrent.futures import ThreadPoolExecutor
    exe = str(tmp_path_factory.mktemp("emul") / "wave_emul")
    subprocess.check_call(["g++", "-O2", "-w", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "emul", "wave_emul.cpp"), "-x", "c", os.path.join(ROOT, "oracle", "gdo_ksw2.c"), "-o", exe])
    pool = ThreadPoolExecutor(max_workers=max(1, min(6, (os.cpu_count() or 2) - 1)))
    runs = {key: pool.submit(subprocess.run, [exe] + args, capture_output=True, text=True) for key, args in _all_commands().items()}
    yield runs
    pool.shutdown(wait=True)


@pytest.mark.parametrize("seed,lanes,n", CORE)
def test_wave_core_matches_oracle(emul, seed, lanes, n):
    out = emul[("core", seed, lanes, n)].result()
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout


@pytest.mark.parametrize("lanes", SINGLE)
def test_single_affine_form_matches_extz2_oracle(emul, lanes):
    """K3: gdw_compute<false> (no X2 / Y2 half) against the oracle's ksw_extz2"""
    out = emul[("single", lanes)].result()
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout


@pytest.mark.parametrize("seed,n,rows", CONE)
def test_checkpointed_cone_pass_matches_oracle(emul, seed, n, rows):
    """the wide-band kernel without a stored backtrace: pass 1 on the 128-position ring with snapshots every `rows` anti-diagonals,
    pass 2 recomputing only the cone of the walk with one block per lane (gdw_cone_row); a cell the walk reads outside the
    recomputed cone is an error"""
    out = emul[("cone", seed, n, rows)].result()
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout


@pytest.mark.parametrize("seed,n,rows", RING96)
def test_checkpointed_96_block_ring_matches_oracle(emul, seed, n, rows):
    """first pass on the 96-block ring (one block + one half block per lane: gdw96_row, half-block core functions, the tracker
    hand-overs across halves, snapshots assembled from halves in the 128-position record format), second pass = the cone with one HALF block per lane
    (gdw_cone_row_half: chunks of at most 496 anti-diagonals)"""
    out = emul[("ring96", seed, n, rows)].result()
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout


@pytest.fixture(scope="module")
def pipe_emul(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emul") / "pipe_emul")
    subprocess.check_call(["g++", "-O2", "-w", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "emul", "pipe_emul.cpp"), "-x", "c", os.path.join(ROOT, "oracle", "gdo_ksw2.c"), "-o", exe])
    return exe


@pytest.mark.parametrize("seed,n,single", [(1, 120, False), (2, 120, False), (3, 80, True)])
def test_skewed_pipeline_matches_oracle(pipe_emul, seed, n, single):
    """ksw_extd2_pipe_kernel (full-matrix short alignments; a lane starts the group's next alignment as soon as its block has left
    the matrix): the device's loop statement by statement on 64 emulated lanes and an emulated LDS -- random geometries of 2..10 blocks,
    1..5 alignments per group, dead slots, Ns, error rates up to 15 % + 12 % indels, what a lane receives from a neighbour on another
    alignment -- score and CIGAR of every alignment against the oracle; every cell of a matrix must have been stored"""
    out = subprocess.run([pipe_emul, str(seed), str(n)] + (["single"] if single else []), capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout
