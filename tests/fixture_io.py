"""readers for the committed mapping fixtures (tests/golden/lr/*, tests/golden/sr/*; inputs made by tools/synth.py /
tools/synth_sr_var.py, golden SAM by the reference binaries oracle/_ref/gdiet_{lr,sr}_avx with the command in *.cmd)"""
import gzip
import hashlib
import os

LR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lr")
SR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sr")
REP = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rep")
# kind -> (directory, read set / golden stem, Mapper preset)
# "hifi_w1": the first 8 reads of hifi.fq with -k 15 -w 1 (every sparsified base is a minimizer: the per-read scratch of the
# seeding kernel overflows its first estimate and the batch is retried with the hard bound)
# "*_edge": reads shorter than k / k+w, all-N, poly-A, lower case, IUPAC codes, chimeras, duplications, lengths around the 300 bp
# switches of both variants, indels, reads flush with a contig start (golden: tools of the reference on these very files)
# "hifi_sv" / "ont_sv": reads with one structural difference each (deletion, insertion, chimera, tandem duplication, inversion;
# tools/synth.py --kind *_sv): second voting round, linked candidates, concatenate_cigars, supplementary and secondary records
# "*_rep": the repeat-rich reference of tools/synth_rep.py (tests/golden/rep/ref.fa.gz: dispersed families, tandem satellites with
# keys above mid_occ and above max_max_occ = 4095, microsatellites) and reads drawn across and inside the repeats: the drop branch of
# mm_seed_mz_flt, mm_seed_select's rescue heap, the max_max_occ cut, multi-occurrence position lists in the index and strands with
# far more than 4096 hits all fire (oracle/make_golden.py prints the counts; tests/test_map_host.py asserts them)
# "sr_rep_f60": the sr_rep reads with -f 60 (mid_occ = 60): at the preset's mid_occ = 1000 the query-side filter cannot fire in the
# ShortReads variant, whose mm_sketch3 stops at 800 minimizers (SR/map.c:621-622)
SETS = {"hifi_rep": (REP, "hifi_rep", "hifi"), "ont_rep": (REP, "ont_rep", "ont"), "sr_rep": (REP, "sr_rep", "sr"), "sr_rep_f60": (REP, "sr_rep_f60", "sr"),
        "hifi_sv": (LR, "hifi_sv", "hifi"), "ont_sv": (LR, "ont_sv", "ont"), "hifi_edge": (LR, "edge_hifi", "hifi"), "ont_edge": (LR, "edge_ont", "ont"), "sr_edge": (SR, "edge", "sr"),
        "hifi_w1": (LR, "w1", "hifi"), "hifi": (LR, "hifi", "hifi"), "ont": (LR, "ont", "ont"), "sr": (SR, "sr", "sr"), "sr_var": (SR, "var", "sr")}
# options of var.cmd that differ from the sr preset (README command): -N 5 -n 0.3,0.1 -s 40 --AF_max_loc 20
OVERRIDES = {"hifi_w1": dict(k=15, w=1), "ont_sv": dict(min_dp_max=4000),  # ont_sv.cmd: -s 4000 (reads of ~14 kbp)
             "ont_rep": dict(min_dp_max=4000), "sr_rep_f60": dict(mid_occ=60),
             "sr_var": dict(best_n=5, min_cnt=0.3, rec_threshold_frac=0.1, min_dp_max=40, AF_max_loc=20)}


# kinds whose --print-seeds stage trace is committed next to the golden SAM (<stem>.trace.gz, oracle/make_golden.py), reduced
# to the lines that start with one of TRACE_PREFIXES
TRACED = ("hifi", "ont", "hifi_sv", "ont_sv", "sr", "hifi_rep", "ont_rep", "sr_rep")
TRACE_PREFIXES = ("Final shift", "RS ", "SD\t", "VT\t", "AVT\t", "BE\t", "AL_SCORE", "CONQ", "CONT")
# kinds whose committed trace holds, per read, ONE line "SDX\t<number of SD lines>\t<sha1 of the sorted SD lines>" in place of the SD
# lines themselves (a read of the repeat-rich sets has up to 280 000 seed hits on a strand: millions of lines per set)
SD_DIGESTED = ("hifi_rep", "ont_rep", "sr_rep")


def variant_of(kind):
    """which of the reference's two trees maps this kind: "sr" = GDiet-ShortReads, "lr" = GDiet-LongReads"""
    return "sr" if SETS[kind][2] == "sr" else "lr"


def digest_sd(lines):
    """the trace lines of a run with every read's SD lines replaced by their count and digest (see SD_DIGESTED)"""
    out, sd = [], []

    def flush():
        if sd:
            out.append("SDX\t%d\t%s" % (len(sd), hashlib.sha1("\n".join(sorted(sd)).encode()).hexdigest()))
            del sd[:]
    for line in lines:
        if line.startswith("SD\t"):
            sd.append(line)
        else:
            flush()
            out.append(line)
    flush()
    return out


# kinds with a committed PAF golden as well (<stem>.golden.paf.gz: the same command with -x instead of -ax, no -a, plus -c
# --paf-no-hit: mm_write_paf3 with the cg:Z: tag and the unmapped lines)
PAF_KINDS = ("hifi_sv", "sr")


def paf_cmd_of(kind):
    out = []
    for tok in cmd_of(kind):
        if tok == "-a":
            continue
        out.append("-x" if tok == "-ax" else tok)
    return out + ["-c", "--paf-no-hit"]


def golden_paf(kind):
    d, stem, _ = SETS[kind]
    return [l.rstrip("\n") for l in gzip.open(os.path.join(d, stem + ".golden.paf.gz"), "rt")]


def read_fasta(path):
    names, seqs, cur = [], [], []
    op = gzip.open if path.endswith(".gz") else open
    for line in op(path, "rt"):
        line = line.rstrip()
        if line.startswith(">"):
            if names:
                seqs.append("".join(cur))
            names.append(line[1:].split()[0])
            cur = []
        else:
            cur.append(line)
    seqs.append("".join(cur))
    return names, seqs


def read_fastq(path):
    op = gzip.open if path.endswith(".gz") else open
    lines = [l.rstrip() for l in op(path, "rt")]
    return [(lines[i][1:].split()[0], lines[i + 1], lines[i + 3]) for i in range(0, len(lines) - 3, 4)]


def golden_sam(kind):
    d, stem, _ = SETS[kind]
    return [l.rstrip("\n") for l in gzip.open(os.path.join(d, stem + ".golden.sam.gz"), "rt")]


def trace_of(kind):
    d, stem, _ = SETS[kind]
    return [l.rstrip("\n") for l in gzip.open(os.path.join(d, stem + ".trace.gz"), "rt")]


def reads_of(kind):
    """the read set a golden SAM was made from"""
    d, stem, _ = SETS[kind]
    if kind == "sr_rep_f60":
        return read_fastq(os.path.join(d, "sr_rep.fq.gz"))
    if kind == "hifi_w1":
        return read_fastq(os.path.join(d, "hifi.fq.gz"))[:8]
    if kind.endswith("_edge"):
        return read_fastq(os.path.join(d, "edge.fq.gz"))
    return read_fastq(os.path.join(d, stem + ".fq.gz"))


def cmd_of(kind):
    d, stem, _ = SETS[kind]
    name = {"hifi_edge": "hifi", "ont_edge": "ont", "sr_edge": "sr"}.get(kind, stem)
    return open(os.path.join(d, name + ".cmd")).read().split()
