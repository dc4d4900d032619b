"""The two finer levels of the drop-in boundary on the GPU:
B4 -- gdiet_hip_seed_batch (mm_sketch2 + mm_get_shift, mm_sketch3, mm_seed_mz_flt, mm_collect_matches2, LR/mmpriv.h:65-76) against THE
      REFERENCE's --print-seeds trace (committed as <kind>.trace.gz by oracle/make_golden.py): the pattern phase of every read ("Final
      shift" lines) and -- expanding every kept seed's occurrences as collect_seed_hits does (LR/map.c:861-955) -- its seed hits on both
      strands ("RS" counts and the SD lines, as a multiset; for the repeat-rich sets their count and sha1);
B2 -- gdiet_hip_map_frag (mm_map_frag's call shape, LR/minimap.h:390) against the golden SAM, read by read."""
import os

import pytest

from fixture_io import OVERRIDES, SD_DIGESTED, SETS, digest_sd, golden_sam, read_fasta, reads_of, trace_of

pytestmark = pytest.mark.gpu


def _per_read(lines):
    out, cur = [], None
    for l in lines:
        if l.startswith("Final shift"):
            cur = {"shift": l, "RS": None, "SD": []}
            out.append(cur)
        elif cur is not None and l.startswith("RS "):
            cur["RS"] = l
        elif cur is not None and l.startswith(("SD\t", "SDX\t")):
            cur["SD"].append(l)
    for r in out:
        r["SD"].sort()
    return out


@pytest.mark.parametrize("kind", ["hifi", "ont", "sr", "hifi_sv", "hifi_rep", "ont_rep", "sr_rep"])
def test_seed_batch_matches_the_reference_trace(gpu_ctx, pkg, kind):
    base, stem, preset = SETS[kind]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
    reads = reads_of(kind)
    want = _per_read(trace_of(kind))
    assert len(want) == len(reads)
    m = pkg.Mapper(gpu_ctx, names, seqs, preset=preset, **OVERRIDES.get(kind, {}))
    try:
        got = m.seed_batch([r[1] for r in reads])
    finally:
        m.close()
    n_sd = 0
    for i, (g, w) in enumerate(zip(got, want)):
        assert "Final shift: %d" % g["shift"] == w["shift"], (i, g["shift"], w["shift"])
        lines, nf, nr = [], 0, 0
        at = 0
        for n_occ, q_pos in g["seeds"]:
            qpos, qstrand = int(q_pos) >> 1, int(q_pos) & 1
            for y in g["occ"][at:at + int(n_occ)]:
                y = int(y)
                rid, loc, strand = y >> 32, (y & 0xffffffff) >> 1, y & 1
                if strand ^ qstrand:  # reverse: target = loc + qpos, printed as (uint32)target + 1   (LR/map.c:897-903, :1335)
                    lines.append("SD\t%s\t%d\t-\t%d" % (names[rid], ((loc + qpos) & 0xffffffff) + 1, qpos))
                    nr += 1
                else:  # forward: target = loc + tmp_extracted_len - qpos, printed as (int32)target + 1 - tmp_extracted_len   (:904-910, :1331)
                    t = (loc + g["tel"] - qpos) & 0xffffffff
                    t = t - (1 << 32) if t >= 1 << 31 else t
                    lines.append("SD\t%s\t%d\t+\t%d" % (names[rid], t + 1 - g["tel"], qpos))
                    nf += 1
            at += int(n_occ)
        assert at == len(g["occ"])
        assert "RS n_a_for: %d, n_a_rev: %d" % (nf, nr) == w["RS"], (i, nf, nr, w["RS"])
        mine = sorted(digest_sd(lines) if kind in SD_DIGESTED else lines)
        assert mine == w["SD"], (i, mine[:2], w["SD"][:2])
        n_sd += nf + nr
    assert n_sd > 1000


@pytest.mark.parametrize("kind", ["hifi", "sr_edge"])
def test_map_frag_matches_golden_sam(gpu_ctx, pkg, kind):
    base, stem, preset = SETS[kind]
    names, seqs = read_fasta(os.path.join(base, "ref.fa.gz"))
    reads = reads_of(kind)[:40]
    want = [l for l in golden_sam(kind) if l.split("\t")[0] in {r[0] for r in reads}]
    m = pkg.Mapper(gpu_ctx, names, seqs, preset=preset, **OVERRIDES.get(kind, {}))
    try:
        got = []
        for qn, sq, ql in reads:
            res = m.map_frag([sq])
            got += m.sam(res, 0, qn, sq, ql)
        if len({r[0] for r in reads}) == len(reads):  # (a read name that occurs twice would be counted twice in `want`)
            assert got == want
        res2 = m.map_frag([reads[0][1], reads[1][1]])  # two segments: only segment 0 is mapped, as in the reference
        assert res2.n_regs[1] == 0 and m.sam(res2, 0, *reads[0]) == m.sam(m.map_frag([reads[0][1]]), 0, *reads[0])
    finally:
        m.close()
