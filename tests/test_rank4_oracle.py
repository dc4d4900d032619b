"""CPU: the oracles of SURVEY 8f rank 4 (ksw_exts2, mg_lchain_dp -- kernels GDiet keeps in its tree and never calls) reproduce the
reference's own outputs stored in tests/golden/ (written by oracle/pin_rank4.py from ksw_exts2_sse / mg_lchain_dp)."""
import numpy as np

from golden_io import SCALARS, load_exts2


def test_exts2_oracle_matches_reference_golden(oracle):
    gdo, lib = oracle
    cases = load_exts2()
    assert len(cases) >= 300
    assert sum(int(((c["cigar"] & 0xf) == 3).sum()) for c in cases) >= 100  # introns (N_SKIP) are really found
    assert sum(c["zdropped"] for c in cases) >= 10 and len({c["flag"] for c in cases}) >= 40
    for c in cases:
        o = gdo.oracle_exts2(lib, c["q"], c["t"], c["mat"], c["go"], c["ge"], c["go2"], c["noncan"], c["zdrop"], c["junc_bonus"], c["flag"], c["junc"])
        for k in SCALARS:
            assert o[k] == c[k], (k, o[k], c[k], hex(c["flag"]))
        assert np.array_equal(o["cigar"], c["cigar"]), hex(c["flag"])
