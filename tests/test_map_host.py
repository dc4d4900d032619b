"""CPU: the product's seeding/voting stage code (map_stages.h, compiled for the host) + host geometry and CIGAR
post-processing (map_host.h), with the oracle's ksw_extd2 standing in for the HIP kernel, reproduce the reference's
SAM byte for byte: against the committed golden SAM, and -- where oracle/_ref has been built -- against a fresh run of
the reference binary on a new random read set."""
import gzip
import os
import shutil
import subprocess

import pytest

from conftest import ROOT
from fixture_io import LR, PAF_KINDS, REP, SD_DIGESTED, SETS, SR, TRACE_PREFIXES, cmd_of, digest_sd, golden_paf, golden_sam, paf_cmd_of, reads_of, trace_of


@pytest.fixture(scope="module")
def host_driver(tmp_path_factory):
    d = tmp_path_factory.mktemp("maphost")
    exe = str(d / "map_host")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-w", "-I", os.path.join(ROOT, "genome-on-diet_amd", "csrc"),
                           "-I", os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "emul", "map_host_main.cpp"),
                           "-x", "c", os.path.join(ROOT, "oracle", "gdo_ksw2.c"), "-o", exe])
    for sub, base, files in (("lr", LR, ("ref.fa", "hifi.fq", "ont.fq")), ("sr", SR, ("ref.fa", "sr.fq", "var.fq")), ("rep", REP, ("ref.fa",))):
        os.makedirs(str(d / sub))
        for f in files:
            with gzip.open(os.path.join(base, f + ".gz"), "rb") as src, open(str(d / sub / f), "wb") as dst:
                shutil.copyfileobj(src, dst)
    return exe, str(d)


THREADS = ["-t", str(min(8, os.cpu_count() or 1))]
REP_KINDS = ("hifi_rep", "ont_rep", "sr_rep", "sr_rep_f60")
_rep_cache = {}


def _rep_run(host_driver, kind, tmp_path):
    """(SAM lines, stderr) of ONE run of the host driver on a repeat-rich set, with --print-seeds --stats: the golden-SAM, stage-trace
    and branch-count tests of these sets share it (a run is 10-15 s: the oracle's scalar DP on reads with tens of thousands of seed hits)"""
    if kind not in _rep_cache:
        exe, d = host_driver
        fq = str(tmp_path / "reads.fq")
        with open(fq, "w") as f:
            for name, seq, qual in reads_of(kind):
                f.write("@%s\n%s\n+\n%s\n" % (name, seq, qual))
        out = subprocess.run([exe] + THREADS + cmd_of(kind) + ["--print-seeds", "--stats", os.path.join(d, "rep", "ref.fa"), fq], capture_output=True, text=True, check=True)
        _rep_cache[kind] = (out.stdout.rstrip("\n").split("\n"), out.stderr)
    return _rep_cache[kind]


@pytest.mark.parametrize("kind", ["hifi", "ont", "sr", "sr_var", "hifi_w1", "hifi_edge", "ont_edge", "sr_edge", "hifi_sv", "ont_sv", "hifi_rep", "ont_rep", "sr_rep", "sr_rep_f60"])
def test_host_path_matches_golden_sam(host_driver, kind, tmp_path):
    exe, d = host_driver
    base = SETS[kind][0]
    if kind in REP_KINDS:
        got, want = _rep_run(host_driver, kind, tmp_path)[0], golden_sam(kind)
        assert len(got) == len(want)
        for a, b in zip(got, want):
            assert a == b, (a[:200], b[:200])
        return
    fq = str(tmp_path / "reads.fq")
    with open(fq, "w") as f:
        for name, seq, qual in reads_of(kind):
            f.write("@%s\n%s\n+\n%s\n" % (name, seq, qual))
    out = subprocess.run([exe] + THREADS + cmd_of(kind) + [os.path.join(d, os.path.basename(base), "ref.fa"), fq], capture_output=True, text=True, check=True)
    got = out.stdout.rstrip("\n").split("\n")
    want = golden_sam(kind)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a == b, (a[:200], b[:200])


@pytest.mark.parametrize("kind", list(PAF_KINDS))
def test_host_path_matches_golden_paf(host_driver, kind, tmp_path):
    """mm_write_paf3 restated (gd_write_paf): the PAF lines of the reference for the same reads -- -c (cg:Z: tag), --paf-no-hit (lines of
    unmapped reads), supplementary / secondary records, the negative de:f values the reference prints for long insertions"""
    exe, d = host_driver
    base = SETS[kind][0]
    fq = str(tmp_path / "reads.fq")
    with open(fq, "w") as f:
        for name, seq, qual in reads_of(kind):
            f.write("@%s\n%s\n+\n%s\n" % (name, seq, qual))
    out = subprocess.run([exe] + THREADS + paf_cmd_of(kind) + [os.path.join(d, os.path.basename(base), "ref.fa"), fq], capture_output=True, text=True, check=True)
    got, want = out.stdout.rstrip("\n").split("\n"), golden_paf(kind)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a == b, (a[:300], b[:300])


def test_host_path_matches_reference_binary_on_fresh_reads(host_driver, tmp_path):
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "gdiet_lr_avx")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    exe, d = host_driver
    fq = str(tmp_path / "fresh.fq")
    ref_fa = os.path.join(d, "lr", "ref.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "synth.py"), "reads", fq, "--ref", ref_fa, "--kind", "hifi", "--n", "20", "--seed", "77"])
    cmd = open(os.path.join(LR, "hifi.cmd")).read().split()
    want = subprocess.run([ref_bin, "-t", "4"] + cmd + [ref_fa, fq], capture_output=True, text=True, check=True).stdout
    want = [l for l in want.rstrip("\n").split("\n") if not l.startswith("@")]
    got = subprocess.run([exe] + cmd + [ref_fa, fq], capture_output=True, text=True, check=True).stdout.rstrip("\n").split("\n")

    def norm(line):
        # known, documented divergence (DESIGN.md "undefined behaviour in the reference"): for a reverse-strand record
        # whose window contains reference Ns, mm_update_extra indexes mat[4*5+7] -- two bytes past the 25-entry
        # matrix on mm_map_frag's stack -- so its ms:i value depends on the reference binary's stack layout.
        f = line.split("\t")
        if len(f) > 11 and f[1] in ("16", "272", "2064") and "nn:i:0" not in f:
            f = [x for x in f if not x.startswith("ms:i:")]
        return "\t".join(f)

    assert [norm(x) for x in got] == [norm(x) for x in want]


@pytest.mark.parametrize("extra", ["", "-N 5 -n 0.3,0.1 -s 40 --AF_max_loc 20", "-k 15 -w 10 -Z 110 -W 3 -i 0.5 -r 0.1,20,100 -n 2,1 -s 30 -N 3"])
def test_sr_host_path_matches_reference_binary_on_fresh_reads(host_driver, tmp_path, extra):
    """ShortReads variant on a fresh read set of mixed lengths (60-600 bp: both geometry branches, SR/map.c:796/809),
    reads flush with the contig ends, Ns -- against a run of the reference binary"""
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "gdiet_sr_avx")
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    exe, d = host_driver
    fq = str(tmp_path / "fresh.fq")
    ref_fa = os.path.join(d, "sr", "ref.fa")
    subprocess.check_call(["python3", os.path.join(ROOT, "tools", "synth_sr_var.py"), "41", "1500", fq, ref_fa])
    cmd = (open(os.path.join(SR, "sr.cmd")).read() + " " + extra).split()
    want = subprocess.run([ref_bin, "-t", "4"] + cmd + [ref_fa, fq], capture_output=True, text=True, check=True).stdout
    want = [l for l in want.rstrip("\n").split("\n") if not l.startswith("@")]
    got = subprocess.run([exe] + cmd + [ref_fa, fq], capture_output=True, text=True, check=True).stdout.rstrip("\n").split("\n")
    assert got == want


@pytest.mark.parametrize("kind", ["hifi", "sr"])
def test_mmi_files_are_exchangeable_with_the_reference(host_driver, tmp_path, kind):
    """the .mmi reader / writer behind gdiet_hip_index_load_mmi / _dump_mmi (map_index.h): (1) an index file written by the
    reference (`GDiet_avx -d`) gives the golden SAM through our path; (2) an index file written by us is accepted by the
    reference and gives the golden SAM through ITS path -- and is the very file `GDiet_avx -d` writes, byte for byte"""
    variant = "sr" if kind == "sr" else "lr"
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "gdiet_%s_avx" % variant)
    if not os.path.exists(ref_bin):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    exe, d = host_driver
    ref_fa = os.path.join(d, variant, "ref.fa")
    fq = str(tmp_path / "reads.fq")
    with open(fq, "w") as f:
        for name, seq, qual in reads_of(kind)[:400]:
            f.write("@%s\n%s\n+\n%s\n" % (name, seq, qual))
    cmd = cmd_of(kind)
    want = subprocess.run([ref_bin, "-t", "4"] + cmd + [ref_fa, fq], capture_output=True, text=True, check=True).stdout
    want = [l for l in want.rstrip("\n").split("\n") if not l.startswith("@")]
    theirs, ours = str(tmp_path / "theirs.mmi"), str(tmp_path / "ours.mmi")
    subprocess.run([ref_bin, "-t", "4"] + cmd + ["-d", theirs, ref_fa], capture_output=True, check=True)
    got = subprocess.run([exe] + cmd + ["--mmi=" + theirs, ref_fa, fq], capture_output=True, text=True, check=True).stdout.rstrip("\n").split("\n")
    assert got == want
    subprocess.run([exe] + cmd + ["--dump-mmi=" + ours, ref_fa, fq], capture_output=True, check=True)
    assert open(ours, "rb").read() == open(theirs, "rb").read()  # byte-identical to mm_idx_dump, khash slot order included
    back = subprocess.run([ref_bin, "-t", "4"] + cmd + [ours, fq], capture_output=True, text=True, check=True).stdout
    assert [l for l in back.rstrip("\n").split("\n") if not l.startswith("@")] == want


def _split_trace(lines):
    """per read: the --print-seeds lines by stage (a read starts at its "Final shift" line)"""
    per_read, cur = [], None
    for line in lines:
        if line.startswith("Final shift"):
            cur = {"shift": line, "RS": [], "SD": [], "VT": [], "AVT": [], "BE": [], "AL": [], "CON": []}
            per_read.append(cur)
        elif cur is not None:
            for key, prefix in (("RS", "RS "), ("SD", "SD\t"), ("SD", "SDX\t"), ("VT", "VT\t"), ("AVT", "AVT\t"), ("BE", "BE\t"), ("AL", "AL_SCORE"), ("CON", "CON")):
                if line.startswith(prefix):
                    cur[key].append(line)
    for r in per_read:
        r["SD"].sort()
    return per_read


@pytest.mark.parametrize("kind", ["hifi", "ont", "sr", "hifi_sv", "ont_sv", "hifi_rep", "ont_rep", "sr_rep"])
def test_stage_trace_matches_the_reference(host_driver, tmp_path, kind):
    """stage-level parity (the reference's --print-seeds trace, committed as <kind>.trace.gz by oracle/make_golden.py):
    chosen pattern phase, sorted seed hits of both strands, vote candidates before and after linking (second voting round
    included), DP boxes, DP scores and the CONQ / CONT line pairs of every concatenate_cigars call, line for line.  Hits with
    equal targets may be ordered differently by the two sorts, so the SD lines are compared as a multiset per read (for the
    repeat-rich sets, whose reads have up to 280 000 hits per strand: as their count and the sha1 of the sorted lines)."""
    exe, d = host_driver
    ref_fa = os.path.join(d, os.path.basename(SETS[kind][0]), "ref.fa")
    fq = str(tmp_path / "reads.fq")
    reads = reads_of(kind)
    if kind in REP_KINDS:
        err = _rep_run(host_driver, kind, tmp_path)[1]
    else:
        with open(fq, "w") as f:
            for name, seq, qual in reads:
                f.write("@%s\n%s\n+\n%s\n" % (name, seq, qual))
        err = subprocess.run([exe] + THREADS + cmd_of(kind) + ["--print-seeds", ref_fa, fq], capture_output=True, text=True, check=True).stderr
    mine = [l for l in err.split("\n") if l.startswith(TRACE_PREFIXES)]
    want, got = _split_trace(trace_of(kind)), _split_trace(digest_sd(mine) if kind in SD_DIGESTED else mine)
    assert len(want) == len(got) == len(reads)
    for a, b in zip(want, got):
        assert a == b
    assert sum(len(r["SD"]) for r in want) > (20 if kind in SD_DIGESTED else 100) and sum(len(r["VT"]) for r in want) > 0
    if kind.endswith("_sv"):  # the fixtures exist for these stages
        assert sum(len(r["CON"]) for r in want) >= 2 * 20


# what each repeat-rich set must reach, as counted by the host emulator's --stats (every number a lower bound on the count of READS
# on which the branch fires): mm_seed_mz_flt dropping minimizers (SR/seed.c:5-29); a streak of seeds above mid_occ with a rescue
# budget (SR/seed.c:79-98) and with more seeds than the budget, i.e. the heap with replacement (:89-94); a seed above max_max_occ
# (:101-102); a strand with more than 4096 hits (the wave vote kernel's LDS-run merge path, MAP_VOTE_CAP)
REP_MUST_REACH = {"hifi_rep": dict(mz_flt_drops=10, rescue=20, heap_replace=10, over_max_max_occ=5, big_strand=50),
                  "ont_rep": dict(mz_flt_drops=4, rescue=10, heap_replace=10, over_max_max_occ=5, big_strand=30),
                  "sr_rep": dict(mz_flt_drops=0, rescue=5, heap_replace=5, over_max_max_occ=100, big_strand=50),
                  "sr_rep_f60": dict(mz_flt_drops=5, rescue=10, heap_replace=10, over_max_max_occ=100, big_strand=1)}


@pytest.mark.parametrize("kind", list(REP_MUST_REACH))
def test_rep_fixtures_reach_the_high_occurrence_branches(host_driver, tmp_path, kind):
    """the repeat-rich sets exist to make the high-occurrence branches of the seeding stage fire; count them (and the index keys
    above mid_occ / max_max_occ) with the product's own stage code on the host"""
    import re
    err = _rep_run(host_driver, kind, tmp_path)[1]
    st = dict(re.findall(r"([a-z_>0-9()]+)=(\d+)", " ".join(l for l in err.split("\n") if l.startswith("[stats]"))))
    num = {k.split("(")[0]: int(v) for k, v in st.items()}
    assert num["keys>mid_occ"] >= 5 and num["keys>max_max_occ"] >= 3 and num["multi"] >= 2000 and num["max_count"] > 4095
    need = REP_MUST_REACH[kind]
    assert num["mz_flt_drops"] >= need["mz_flt_drops"] and num["rescue"] >= need["rescue"] and num["heap_replace"] >= need["heap_replace"]
    assert num["over_max_max_occ"] >= need["over_max_max_occ"] and num["strand>4096hits"] >= need["big_strand"]


def test_rep_mmi_files_are_the_reference_s(host_driver, tmp_path):
    """the .mmi writer on an index with thousands of multi-occurrence position lists (p[] arrays of up to 8 000 entries, khash
    buckets with non-singleton keys): the file has the size and sha256 of the one `GDiet_avx -d` wrote for the same reference and
    command line (tests/golden/rep/mmi.sha256.json, oracle/make_golden.py) -- the other references' p[] arrays are empty"""
    import hashlib
    import json
    exe, d = host_driver
    want = json.load(open(os.path.join(REP, "mmi.sha256.json")))
    fq = str(tmp_path / "none.fq")
    open(fq, "w").write("@r\nACGT\n+\nIIII\n")
    for tag, kind in (("k19w19", "hifi_rep"), ("k15w10", "ont_rep"), ("k21w11", "sr_rep")):
        ours = str(tmp_path / (tag + ".mmi"))
        subprocess.run([exe] + cmd_of(kind) + ["--dump-mmi=" + ours, os.path.join(d, "rep", "ref.fa"), fq], capture_output=True, check=True)
        data = open(ours, "rb").read()
        assert len(data) == want[tag]["size"] and hashlib.sha256(data).hexdigest() == want[tag]["sha256"], tag


def test_sv_fixtures_reach_concatenation_supplementary_and_secondary_records():
    """what the *_sv goldens hold (>= 50 concatenate_cigars events, >= 30 supplementary, >= 5 secondary records)"""
    flags, conq = {}, 0
    for kind in ("hifi_sv", "ont_sv"):
        for line in golden_sam(kind):
            fl = int(line.split("\t")[1])
            flags[fl] = flags.get(fl, 0) + 1
        conq += sum(1 for l in trace_of(kind) if l.startswith("CONQ"))
    assert conq >= 50 and flags.get(2048, 0) + flags.get(2064, 0) >= 30 and flags.get(256, 0) + flags.get(272, 0) >= 5


def test_committed_goldens_are_what_the_reference_prints():
    """oracle/make_golden.py in check mode: every committed golden SAM / trace / parser fixture is reproduced by the reference
    binaries compiled here (skipped where /root/reference, hence oracle/_ref, does not exist)"""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "gdiet_lr_avx")):
        pytest.skip("oracle/_ref not built (no /root/reference on this machine)")
    r = subprocess.run(["python3", os.path.join(ROOT, "oracle", "make_golden.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
