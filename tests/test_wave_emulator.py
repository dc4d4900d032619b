"""CPU: the per-lane code of the register-resident HIP kernel (genome-on-diet_amd/csrc/ksw_wave_core.h), driven by a
host lock-step emulator, reproduces the oracle's score and CIGAR bit for bit on seeded random pairs."""
import os
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def emul(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("emul") / "wave_emul")
    subprocess.check_call(["g++", "-O2", "-w", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"), "-I", os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "emul", "wave_emul.cpp"), "-x", "c", os.path.join(ROOT, "oracle", "gdo_ksw2.c"), "-o", exe])
    return exe


@pytest.mark.parametrize("seed,lanes,n", [(1, 64, 400), (2, 64, 400), (3, 16, 400), (4, 128, 60), (5, 10, 400), (6, 8, 400)])
def test_wave_core_matches_oracle(emul, seed, lanes, n):
    out = subprocess.run([emul, str(seed), str(n), str(lanes)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout


@pytest.mark.parametrize("lanes", [16, 64, 10])
def test_single_affine_form_matches_extz2_oracle(emul, lanes):
    """K3: gdw_compute<false> (no X2 / Y2 half) against the oracle's ksw_extz2"""
    out = subprocess.run([emul, "9", "300", str(lanes), "single"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout


@pytest.mark.parametrize("seed,n,rows", [(4, 60, 480), (7, 40, 320), (8, 30, 64)])
def test_checkpointed_cone_pass_matches_oracle(emul, seed, n, rows):
    """the wide-band kernel without a stored backtrace: pass 1 on the 128-position ring with snapshots every `rows` anti-diagonals,
    pass 2 recomputing only the cone of the walk with one block per lane (gdw_cone_row); a cell the walk reads outside the
    recomputed cone is an error"""
    out = subprocess.run([emul, str(seed), str(n), "128", "ckpt", str(rows)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout


@pytest.mark.parametrize("seed,n,rows", [(4, 60, 480), (7, 40, 200)])
def test_checkpointed_96_block_ring_matches_oracle(emul, seed, n, rows):
    """first pass on the 96-block ring (one block + one half block per lane: gdw96_row, half-block core functions, the tracker
    hand-overs across halves, snapshots assembled from halves in the 128-position record format), second pass = the cone with one HALF block per lane
    (gdw_cone_row_half: chunks of at most 496 anti-diagonals)"""
    out = subprocess.run([emul, str(seed), str(n), "96", "ckpt96", str(rows)], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout
