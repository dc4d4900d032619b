#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X Genome-on-Diet hot path (contract: see the task statement).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload = BASELINE.json configs[3], the configuration the metric is quoted on:
    GDiet-LongReads `-ax map-hifi -Z 10 -W 2 -i 0.2 -k 19 -w 19 -N 1 -r 1000 --vt_dis=650 --vt_nb_loc=5 --vt_df1=0.0106
    --vt_df2=0.2 -s 400 --vt_cov 0.04 --max_min_gap=4000 --vt_f=0.04 --secondary=yes` (reference README.md:44), synthetic HiFi
    reads ~15 kbp (N(15000,2000) in [5000,25000], sub .2 % / ins .1 % / del .1 %, 50 % reverse strand) against a synthetic
    GRCh38-sized reference (24 contigs with the GRCh38 primary-chromosome lengths, i.i.d. ACGT + a 5 % repeat layer at 2 %
    divergence + N runs).  `--ref-mbp` scales the reference down for quick runs (the JSON says which size ran).

A "step" = one pass of the whole per-read path (sketch2/shift/sketch3, seed filter + lookup, hit sort, vote/vote_2 on the
GPU; candidate geometry on host threads; window gather, exact-match, ksw_extd2 DP, backtrack on the GPU; mm_update_extra /
concatenate_cigars / mm_set_sam_params on host threads) over one batch of reads that is already resident in HBM.  Every step
maps a DIFFERENT batch: ceil(200 000 / batch) = 40 distinct batches of 5120 reads are synthesised and uploaded before the timed
region (configs[3]'s 200 k reads; the default --steps 40 goes through each of them once, more steps cycle).  With the
default --inflight 2 step i+1 is submitted (gdiet_hip_map_submit) before step i is waited for, so its seeding / voting / host
stages overlap the DP kernel of step i; all K batches are complete when the timed region ends (--inflight 1 runs them one at
a time).  value = bases of reads with >= 1 alignment / wall time, summed over ranks (reads are sharded over GPUs, index
replicated, no collective).  p50_read_latency_ms = median over the reads of the timed steps of (gdiet_hip_map_wait of their batch
returned - gdiet_hip_map_submit was called): every read of a batch completes with its batch.  config.with_upload: the same
steps once more, outside the timed region, with the reads handed over as HOST buffers (gdiet_hip_batch_upload = encode + H2D inside
the pipeline): the PCIe-inclusive rate, never `value`.

roofline: dominant kernel = ksw_extd2_wave_kernel<64, 0, true> (DP + its own backtrack); achieved = algorithmic bytes of the
launch (per alignment (qlen+tlen-1)*min(w+1,qlen,tlen) + (qlen+tlen) + qlen + ceil(tlen/2)) / its duration, the mean
over the K timed launches of HIP events recorded on the stream each launch went to; traffic = PMC bytes from profiles/.
cpu_baseline: the reference binary itself (oracle/_ref/gdiet_lr_avx = GDiet_avx) where it travelled with the repo, mapping
a bounded sample of the same kind of reads against the contig they were drawn from, best of several thread counts (rank 0,
N = 1 only); else the DP stage of the oracle port on one core.
"""
import argparse
import glob
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
W_HIFI = 1000
GRCH38 = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636, 138394717, 133797422,
          135086622, 133275309, 114364328, 107043718, 101991189, 90338345, 83257441, 80373285, 58617616, 64444167,
          46709983, 50818468, 156040895, 57227415]
BASES = np.frombuffer(b"ACGT", dtype=np.uint8)
COMP = np.zeros(256, np.uint8)
COMP[[65, 67, 71, 84, 78]] = [84, 71, 67, 65, 78]


def synth_reference(total_mbp, seed=2):
    """24 contigs in GRCh38 proportions, i.i.d. ACGT, repeat layer ~5 % at 2 % divergence, N runs"""
    rng = np.random.default_rng(seed)
    scale = total_mbp * 1e6 / sum(GRCH38)
    lens = [max(20000, int(l * scale)) for l in GRCH38]
    contigs = [BASES[rng.integers(0, 4, size=l, dtype=np.uint8)] for l in lens]
    total = sum(lens)
    fams = [BASES[rng.integers(0, 4, size=int(rng.integers(300, 6000)), dtype=np.uint8)] for _ in range(2000)]
    p = np.array(lens, np.float64) / total
    n_ins = int(0.05 * total / 3150)
    cs = rng.choice(len(contigs), size=n_ins, p=p)
    fi = rng.integers(0, len(fams), size=n_ins)
    for c, f in zip(cs, fi):
        fam = fams[f]
        if lens[c] <= len(fam) + 10:
            continue
        pos = int(rng.integers(0, lens[c] - len(fam)))
        cp = fam.copy()
        m = rng.random(len(cp)) < 0.02
        cp[m] = BASES[rng.integers(0, 4, size=int(m.sum()))]
        if rng.random() < 0.5:
            cp = COMP[cp[::-1]]
        contigs[c][pos:pos + len(cp)] = cp
    for _ in range(max(1, int(total * 1e-4 / 50))):
        c = int(rng.choice(len(contigs), p=p))
        pos = int(rng.integers(0, lens[c] - 100))
        contigs[c][pos:pos + int(rng.integers(10, 90))] = 78
    return ["chr%d" % (i + 1) for i in range(len(contigs))], contigs


def synth_hifi_reads(rng, contigs, n, only_contig=None, first=0):
    """SURVEY 8d HiFi reads: length N(15000, 2000) clipped to [5000, 25000], sub 0.2 % / ins 0.1 % / del 0.1 %, half of them
    reverse-complemented.  Error positions are drawn as counts (binomial) + positions, not as one uniform per base: 40 batches
    of 5120 reads have to be made before the timed region."""
    lens = np.array([len(c) for c in contigs], np.float64)
    pc = lens / lens.sum()
    out = []
    for i in range(n):
        ln = int(np.clip(rng.normal(15000, 2000), 5000, 25000))
        c = only_contig if only_contig is not None else int(rng.choice(len(contigs), p=pc))
        ln = min(ln, len(contigs[c]) - 2)
        st = int(rng.integers(0, len(contigs[c]) - ln))
        s = contigs[c][st:st + ln].copy()
        m = rng.integers(0, ln, size=rng.binomial(ln, 0.002))
        m = m[s[m] != 78]
        if len(m):
            s[m] = BASES[(np.searchsorted(BASES, s[m]) + rng.integers(1, 4, size=len(m))) & 3]
        d = rng.integers(0, ln, size=rng.binomial(ln, 0.001))
        if len(d):
            s = np.delete(s, d)
        ip = rng.integers(0, len(s), size=rng.binomial(len(s), 0.001))
        if len(ip):
            s = np.insert(s, ip, BASES[rng.integers(0, 4, size=len(ip))])
        if rng.random() < 0.5:
            s = COMP[s[::-1]]
        out.append(("r%d_c%d_%d" % (first + i, c + 1, st), s.tobytes()))
    return out


def shared_reference(total_mbp, local_rank, local_world, barrier, tag):
    """the synthetic reference once per NODE: local rank 0 synthesises it into a file in /dev/shm, the other ranks of the node map
    that file (np.memmap) instead of spending 6 s and 3 GB each on an identical copy.  Returns (names, contigs, seconds, how)"""
    t0 = time.time()
    if local_world <= 1:
        names, contigs = synth_reference(total_mbp, seed=2)
        return names, contigs, time.time() - t0, "synthesised in-process"
    path = "/dev/shm/gdiet_bench_ref_%s_%d.u8" % (tag, int(total_mbp))
    if local_rank == 0:
        names, contigs = synth_reference(total_mbp, seed=2)
        tmp = path + ".tmp"
        with open(tmp, "wb") as f:
            for c in contigs:
                f.write(c.tobytes())
        with open(path + ".json", "w") as f:
            json.dump({"names": names, "lens": [len(c) for c in contigs]}, f)
        os.replace(tmp, path)
    barrier()
    if local_rank != 0:
        meta = json.load(open(path + ".json"))
        mm = np.memmap(path, dtype=np.uint8, mode="r")
        off = np.concatenate([[0], np.cumsum(meta["lens"])])
        names, contigs = meta["names"], [mm[off[i]:off[i + 1]] for i in range(len(meta["lens"]))]
    barrier()
    if local_rank == 0:  # every rank has the file mapped: the name can go (the pages live as long as the mappings)
        for f in (path, path + ".json"):
            try:
                os.unlink(f)
            except OSError:
                pass
    return names, contigs, time.time() - t0, "synthesised by local rank 0, shared through /dev/shm"


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline_reference(names, contigs, rng, cores, which="whole", gpu_reads=None):
    """the reference's own GDiet_avx on the host: build a .mmi with -d (outside the timing), then time mapping only.
    which = "whole" (the default: the very reference the GPU leg maps against -- ~1 min of indexing and a 4 GB file for 3.1 Gbp --
    and the FIRST 2048 READS OF THE GPU LEG's first batch) | "largest" (chr1-sized contig only, reads drawn from it) | "smallest"."""
    exe = os.path.join(ROOT, "oracle", "_ref", "gdiet_lr_avx")
    lens = [len(x) for x in contigs]
    sel = list(range(len(contigs))) if which == "whole" else [int(np.argmax(lens) if which == "largest" else np.argmin(lens))]
    sub = [contigs[i] for i in sel]
    if which == "whole" and gpu_reads:
        reads = list(gpu_reads[:2048])
    else:
        reads = synth_hifi_reads(rng, sub, 2048, only_contig=None if which == "whole" else 0)  # ~31 Mbases: seconds of wall time on the host's cores
    hifi = ("-ax map-hifi -Z 10 -W 2 -i 0.2 -k 19 -w 19 -N 1 -r 1000 --vt_dis=650 --vt_nb_loc=5 --vt_df1=0.0106 --vt_df2=0.2 -s 400 "
            "--vt_cov 0.04 --max_min_gap=4000 --vt_f=0.04 --sort=merge --frag=no -F200,1 --secondary=yes -a").split()
    with tempfile.TemporaryDirectory() as d:
        fa, fq, mmi = os.path.join(d, "c.fa"), os.path.join(d, "r.fq"), os.path.join(d, "c.mmi")
        with open(fa, "wb") as f:
            for i in sel:
                f.write(b">" + names[i].encode() + b"\n" + np.asarray(contigs[i]).tobytes() + b"\n")
        with open(fq, "wb") as f:
            for nm, s in reads:
                f.write(b"@" + nm.encode() + b"\n" + s + b"\n+\n" + b"I" * len(s) + b"\n")
        t0 = time.time()
        subprocess.run([exe, "-t", str(cores)] + hifi + ["-d", mmi, fa], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True, timeout=420)
        os.unlink(fa)
        t_index = time.time() - t0
        # `cores` = the CPUs this container may use (cgroup quota / affinity, not the host's thread count).  The reference's thread
        # scaling is not perfect (its per-alignment 30 MB backtrace allocations serialise in the kernel), so the baseline is the BEST
        # of a few thread counts on the same sample, not simply -t <all cores>
        best, tried, sam_body = None, [], None
        for t in sorted({max(1, cores // 2), cores, 2 * cores}):
            r = subprocess.run([exe, "-t", str(t)] + hifi + [mmi, fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
            err = r.stderr.decode(errors="ignore")
            if sam_body is None:  # the reference's records for these very reads: diffed with the GPU path's outside the timed region
                sam_body = [l for l in r.stdout.decode().split("\n") if l and not l.startswith("@")]
            m = re.search(r"\[M::main::([0-9.]+)\*[0-9.]+\] loaded/built the index", err)
            load = float(m.group(1)) if m else 0.0
            m2 = re.search(r"Real time: ([0-9.]+) sec", err)
            if not m2:
                continue
            names_mapped = {l.split("\t")[0] for l in r.stdout.decode().split("\n") if l and not l.startswith("@") and l.split("\t")[1] != "4"}
            mapped = sum(len(s) for nm, s in reads if nm in names_mapped)
            v = mapped / max(1e-6, float(m2.group(1)) - load)
            tried.append("-t %d: %.2f Mbases/s" % (t, v / 1e6))
            if best is None or v > best[0]:
                best = (v, t)
        # the second half of the metric on the CPU side: the reference's own per-read probe (--print-qname: one "QT <name> <thread> <sec>"
        # line per read, the wall time of its mm_map_frag call, LR/map.c:2020-2021), at the best thread count found above
        qt_ms, qt_note = None, None
        try:
            r = subprocess.run([exe, "-t", str(best[1])] + hifi + ["--print-qname", mmi, fq], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=900)
            qt = [float(l.split("\t")[3]) for l in r.stderr.decode(errors="ignore").split("\n") if l.startswith("QT\t") and len(l.split("\t")) >= 4]
            if qt:
                qt_ms = 1e3 * float(np.median(qt))
                qt_note = ("median of the reference's %d QT lines (--print-qname, -t %d): wall time of one read's mm_map_frag call on one thread; "
                           "p10 / p90: %.1f / %.1f ms" % (len(qt), best[1], 1e3 * float(np.percentile(qt, 10)), 1e3 * float(np.percentile(qt, 90))))
        except Exception as ex:
            qt_note = "QT run failed: %r" % (ex,)
    what = {"largest": "the largest contig", "smallest": "the smallest contig", "whole": "the whole reference"}[which]
    return {"value": best[0], "unit": "mapped bases/s", "cores": best[1], "kind": "reference", "cpu_model": cpu_model(), "cpus_usable": cores,
            "p50_read_latency_ms": qt_ms, "p50_read_latency_note": qt_note,
            "_sam_body": sam_body if which == "whole" and gpu_reads else None, "_reads": reads if which == "whole" and gpu_reads else None,
            "sample": "%d HiFi reads (%d bases, " % (len(reads), sum(len(s) for _, s in reads)) + ("the first reads of the GPU leg's first batch" if which == "whole" and gpu_reads else "the GPU leg's generator") +
                      ") against %s (%s, %d bp), GDiet_avx with a .mmi of it prebuilt in %.0f s "
                      "(outside the timing), mapping wall time only (index load excluded); best of %s"
                      % (what, "+".join(names[i] for i in sel) if which != "whole" else "%d contigs" % len(sel),
                         sum(lens[i] for i in sel), t_index, "; ".join(tried))}


def cpu_baseline_port(reads_enc, budget_s=12.0):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gdo
    a, b, q, e, q2, e2 = gdo.PRESETS["hifi"]
    mat, lib = gdo.score_matrix(a, b), gdo.load_oracle()
    t0, bases, n = time.time(), 0, 0
    for s in reads_enc:
        gdo.oracle_extd2(lib, s, s, mat, q, e, q2, e2, W_HIFI, flag=gdo.EZ_APPROX_MAX | gdo.EZ_AVX512_SC)
        bases += len(s)
        n += 1
        if time.time() - t0 > budget_s:
            break
    return {"value": bases / (time.time() - t0), "unit": "mapped bases/s", "cores": 1, "kind": "port",
            "sample": "%d full-read candidate alignments (%d bases): DP + backtrack stage of the oracle port only, 1 thread" % (n, bases)}


def _ms_undefined(line):
    """for a reverse-strand record whose window holds reference Ns under read Ns, mm_update_extra reads
    mat[4*5+7], two bytes past the score matrix on mm_map_frag's stack: the ms:i tag of such a record is whatever the reference binary's
    stack holds.  Such records are compared without that tag (and counted)."""
    f = line.split("\t")
    return len(f) > 11 and (int(f[1]) & 16) and "nn:i:0" not in f


def parity_at_scale(mapper, reads, ref_sam):
    res = mapper.map([s for _, s in reads])
    mine = [l for l in mapper.sam_batch(res, [(nm, s, b"I" * len(s)) for nm, s in reads]).split("\n") if l]
    raw = sum(a != b for a, b in zip(mine, ref_sam)) + abs(len(mine) - len(ref_sam))

    def norm(l):
        return "\t".join(x for x in l.split("\t") if not x.startswith("ms:i:")) if _ms_undefined(l) else l
    diff = sum(norm(a) != norm(b) for a, b in zip(mine, ref_sam)) + abs(len(mine) - len(ref_sam))
    flags = {}
    for l in ref_sam:
        fl = l.split("\t")[1]
        flags[fl] = flags.get(fl, 0) + 1
    return {"reads": len(reads), "sam_lines": len(ref_sam), "differing_lines": diff, "differing_lines_raw": raw,
            "records_over_reference_N_on_the_reverse_strand": sum(1 for l in ref_sam if _ms_undefined(l)), "flags": flags,
            "note": "GDiet_avx (-t N, whole %s index) vs gdiet_hip_map_batch + gdiet_hip_sam_batch on the same reads; differing_lines ignores the ms:i tag of "
                    "reverse-strand records over reference Ns (undefined in the reference: an out-of-bounds read of its stack, DESIGN.md 5.2), "
                    "differing_lines_raw does not" % "GRCh38-sized"}


def latency_modes(mapper, ctx, batches, modes):
    """the throughput / latency trade (outside the timed region): synchronous calls (one batch in flight: a read's latency is its batch's
    step time) at several batch sizes, on reads of the first resident batches"""
    out = []
    pool = [r for b in batches[:2] for r in b[1]]
    for n_reads, steps in modes:
        n_reads = min(n_reads, len(pool))
        sets = []
        for k in range(steps):  # distinct reads per step where the pool allows
            o = (k * n_reads) % max(1, len(pool) - n_reads + 1)
            sets.append(pool[o:o + n_reads])
        ups = [mapper.upload([s for _, s in rs]) for rs in sets]
        mapper.map_uploaded(ups[0])  # buffers sized
        lat, bases = [], 0
        t0 = time.perf_counter()
        for u, rs in zip(ups, sets):
            t1 = time.perf_counter()
            res = mapper.map_uploaded(u)
            lat.append(time.perf_counter() - t1)
            nr = np.frombuffer(res.n_regs, dtype=np.int32, count=res.n)
            bases += int(sum(len(s) for (_, s), m in zip(rs, nr > 0) if m))
        dt = time.perf_counter() - t0
        for u in ups:
            mapper.free_batch(u)
        out.append({"reads_per_batch": n_reads, "batches_in_flight": 1, "steps": steps, "p50_read_latency_ms": 1e3 * float(np.median(lat)),
                    "bases_per_s": bases / dt, "dp_kernel_ms_last": ctx.last_kernel_ms()[0]})
    return out


def self_check(res, res_again, reads):
    n = len(reads)
    same, cig_ok, placed, mapped = True, 0, 0, 0
    for i in range(n):
        nr = res.n_regs[i]
        same &= nr == res_again.n_regs[i]
        if nr <= 0:
            continue
        mapped += 1
        for j in range(nr):
            a, b = res.regs[i][j], res_again.regs[i][j]
            same &= (a.rid, a.rs, a.re, a.qs, a.qe, a.score, a.mapq, a.n_cigar) == (b.rid, b.rs, b.re, b.qs, b.qe, b.score, b.mapq, b.n_cigar)
        r = res.regs[i][0]
        cg = np.ctypeslib.as_array(r.cigar, shape=(r.n_cigar,))
        ops, lens = cg & 0xf, cg >> 4
        cig_ok += int(lens[(ops == 0) | (ops == 1)].sum() == r.qe - r.qs and lens[(ops == 0) | (ops == 2)].sum() == r.re - r.rs)
        _, c, st = reads[i][0].split("_")
        placed += int(r.rid == int(c[1:]) - 1 and abs(r.rs - int(st)) < 2000)
    return {"reads": n, "mapped": mapped, "pipelined_equals_synchronous": bool(same), "cigar_spans_consistent": cig_ok, "primary_at_true_origin": placed}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40, help="timed steps; 40 x 5120 reads = configs[3]'s 200 k reads, every step a different batch")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=5120,
                    help="reads per step per GPU (5120 reads = ~9 400 alignments: just under two rounds of the 5 120 resident wavefront slots; "
                         "measured best of 4096..6144; its backtrace arena is 175 GB of the 288 GB)")
    ap.add_argument("--total-reads", type=int, default=200000, help="reads of the workload (configs[3]); ceil(total / batch) distinct batches are made, at most --steps")
    ap.add_argument("--ref-mbp", type=float, default=float(os.environ.get("GDIET_BENCH_REF_MBP", "3088")),
                    help="size of the synthetic reference in Mbp (default: GRCh38-sized)")
    ap.add_argument("--lanes", type=int, default=1, help="software-pipeline depth inside a step (gdiet_hip_set_map_lanes)")
    ap.add_argument("--inflight", type=int, default=2, choices=[1, 2, 3, 4],
                    help="batches in flight (gdiet_hip_map_submit/_wait): 2 (the library's default) overlaps the seeding / voting / host stages of step i+1 with the "
                         "DP kernel of step i -- since round 3 that keeps the DP kernels back to back (the side kernels run at a raised wave priority), at two step "
                         "times of latency; 3 (rounds 1-2) adds a step of latency for nothing")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = every rank maps its own --batch reads per step (the contract's default); strong = ONE read set, each "
                         "batch cut into contiguous ranges of equal DP cost (read_ranges_by_cost), one per rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-ref", default="whole", choices=["whole", "largest", "smallest"],
                    help="what the reference binary indexes for the CPU baseline (whole: the GPU leg's reference and reads; falls back to largest if that fails)")
    ap.add_argument("--no-upload-pass", action="store_true", help="skip the PCIe-inclusive pass (config.with_upload)")
    ap.add_argument("--no-latency-modes", action="store_true", help="skip the latency_mode passes (synchronous calls at smaller batch sizes, outside the timed region)")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL; gloo only to rehearse the N > 1 path on a one-GPU box)")
    ap.add_argument("--device", type=int, default=-1, help="HIP device of this rank (default: LOCAL_RANK)")
    ap.add_argument("--host-threads", type=int, default=0, help="host threads of the post-processing pool (default: the CPUs this container may use / ranks)")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    local_rank = local
    import torch.distributed as dist
    if args.device >= 0:
        local = args.device
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.dist_backend)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from __graft_entry__ import _load_pkg
    pkg = _load_pkg()
    pkg_cpus = pkg.effective_cpus()
    cores = max(1, pkg_cpus // max(1, local_world))  # CPUs this container may use (cgroup quota, affinity), shared by the ranks of the node
    if world > 1 and cores < 6:
        # several ranks on few CPUs (8 ranks on a 16-CPU quota: 2 each): every rank has two lane threads waiting for the GPU most of the time;
        # let them sleep on a blocking event instead of spinning in hipStreamSynchronize (read when the context is created)
        os.environ.setdefault("GDIET_SYNC", "block")
    ctx = pkg.Context(local)

    # the same reference on every rank (replicated index): synthesised once per node
    names, contigs, t_ref, ref_how = shared_reference(args.ref_mbp, local_rank, local_world, (lambda: dist.barrier()) if world > 1 else (lambda: None),
                                                      os.environ.get("MASTER_PORT", "0"))
    t1 = time.time()
    mapper = pkg.Mapper(ctx, names, contigs, preset="hifi", n_threads=cores)
    host_threads = args.host_threads or cores
    mapper.set_host_threads(host_threads)  # N ranks on one node share its cores
    t_index = time.time() - t1
    # ---- the read set: distinct batches, uploaded before the timed region (value = inputs resident in HBM)
    n_distinct = max(1, min(args.steps, -(-args.total_reads // args.batch)))
    t2 = time.time()
    batches = []  # (device batch, reads [(name, bytes)], read_lens)
    if args.scaling == "strong" and world > 1:
        rng = np.random.default_rng(5)  # ONE read set, identical on every rank; each rank keeps its cost-balanced contiguous share
        for b in range(n_distinct):
            allr = synth_hifi_reads(rng, contigs, args.batch, first=b * args.batch)
            bounds = pkg.read_ranges_by_cost([len(s) for _, s in allr], world, band=W_HIFI)
            mine = allr[bounds[rank]:bounds[rank + 1]]
            batches.append((mapper.upload([s for _, s in mine]), mine, np.array([len(s) for _, s in mine])))
    else:
        rng = np.random.default_rng(pkg.rank_seed(5, rank))  # HiFi reads seed 5 (+rank: read-sharded weak scaling)
        for b in range(n_distinct):
            reads = synth_hifi_reads(rng, contigs, args.batch, first=b * args.batch)
            batches.append((mapper.upload([s for _, s in reads]), reads, np.array([len(s) for _, s in reads])))
    t_reads = time.time() - t2

    clock = pkg.JobClock(dist if world > 1 else None, dev, lambda: torch.cuda.synchronize(dev))
    mapper.set_lanes(args.lanes)
    if args.inflight > 1:
        mapper.set_inflight(args.inflight)
    all_cells = [0, 0]  # DP cells / launches of EVERY step made by this process (the PMC summaries divide their counters by these)
    step_no = [0]

    def run_steps(k, rec=None, host_buffers=False):
        """k steps, each on the next batch of the cycle; with --inflight N the next N-1 steps are submitted before step i is waited
        for.  rec: dict of lists filled per completed step (stage seconds, kernel ms of THAT step's launch -- HIP events on the
        stream it was launched on --, its DP work, its mapped bases, submit -> wait-return latency).  host_buffers: the reads go in
        as host buffers (gdiet_hip_batch_upload inside the step) instead of the resident batch."""
        last, open_t = None, []

        def finish(item):
            ticket, bi, t_sub, tmp_batch = item
            res = mapper.wait(ticket) if ticket is not None else None
            return res, bi, t_sub, tmp_batch

        def done(res, bi, t_sub, tmp_batch):
            t_done = time.perf_counter()
            cells, alg = ctx.last_dp_work()
            all_cells[0] += cells
            all_cells[1] += 1
            if tmp_batch is not None:
                mapper.free_batch(tmp_batch)
            if rec is not None:
                nr = np.frombuffer(res.n_regs, dtype=np.int32, count=res.n) if res.n else np.zeros(0, np.int32)
                rec["stages"].append(mapper.stage_seconds())
                rec["kern"].append(ctx.last_kernel_ms())
                rec["work"].append((cells, alg))
                rec["mapped_bases"].append(int(batches[bi][2][nr > 0].sum()))
                rec["mapped_reads"].append(int((nr > 0).sum()))
                rec["latency"].append(t_done - t_sub)
                rec["aligns"].append(int(nr.sum()))
            return res

        ahead = None  # host_buffers: the next step's batch, encoded and copied while the lanes work on the ones in flight
        for it in range(k):
            bi = step_no[0] % len(batches)
            step_no[0] += 1
            t_sub = time.perf_counter()
            tmp_batch = None
            dbatch = batches[bi][0]
            if host_buffers:
                tmp_batch = dbatch = ahead if ahead is not None else mapper.upload([s for _, s in batches[bi][1]])
                ahead = None
            if args.inflight == 1:
                res = mapper.map_uploaded(dbatch)  # returns when every read of the batch has its records on the host
                last = (done(res, bi, t_sub, tmp_batch), bi)
            else:
                open_t.append((mapper.submit(dbatch), bi, t_sub, tmp_batch))
                if host_buffers and it + 1 < k:
                    ahead = mapper.upload([s for _, s in batches[step_no[0] % len(batches)][1]])
                if len(open_t) == args.inflight:
                    r = finish(open_t.pop(0))
                    last = (done(*r), r[1])
        while open_t:
            r = finish(open_t.pop(0))
            last = (done(*r), r[1])
        return last

    run_steps(args.inflight)  # set-up, like the index build: each lane allocates its device scratch on its first batch
    run_steps(args.warmup)
    clock.start()  # synchronize + barrier + synchronize
    rec = {k: [] for k in ("stages", "kern", "work", "mapped_bases", "mapped_reads", "latency", "aligns")}
    res, res_bi = run_steps(args.steps, rec)  # every record of all K batches is on the host when this returns
    elapsed = clock.stop()  # synchronize + barrier
    # the PCIe-inclusive figure (never `value`): the same kind of steps with the reads handed over as host buffers
    with_upload = None
    if not args.no_upload_pass:
        k2 = min(args.steps, 10)
        rec2 = {k: [] for k in rec}
        t3 = time.perf_counter()
        run_steps(k2, rec2, host_buffers=True)
        torch.cuda.synchronize(dev)
        dt2 = time.perf_counter() - t3
        with_upload = {"steps": k2, "ms_per_step": 1e3 * dt2 / k2, "bases_per_s_this_rank": sum(rec2["mapped_bases"]) / dt2,
                       "note": "gdiet_hip_batch_upload (host threads encode, one H2D copy) of the next batch while the lanes work, then submit; outside the timed region"}
    # for reference: the last launch once more with nothing else on the GPU (outside the timed region)
    mapper.set_lanes(1)
    res1 = mapper.map_uploaded(batches[res_bi][0])
    dp_alone = ctx.last_kernel_ms()[0]
    all_cells[0] += ctx.last_dp_work()[0]
    all_cells[1] += 1
    # size-independent self-checks at full size (outside the timed region): the pipelined steps and this synchronous pass give the
    # same records (idempotence); every CIGAR consumes exactly its query and reference interval; the primary record of a mapped
    # read lies where the read was drawn from (the read names carry contig and start)
    check = self_check(res, res1, batches[res_bi][1])
    del res1
    lat_modes = None
    if rank == 0 and world == 1 and not args.no_latency_modes:  # (one rank only: the other ranks of a multi-GPU run would wait in the final reduction meanwhile)
        # the same pipelined steps with fewer batches in flight: a batch then waits for fewer DP kernels ahead of it (p50 ~ depth x step)
        lat_modes = []
        # (gdiet_hip_set_dp_waves(4) -- a fifth of every SIMD's registers left to the next batch's seeding / voting kernels -- was measured
        # here as well: the DP kernel loses 10 %, 762 Mbases/s at 202 ms with two batches in flight; profiles/r03_latency_modes.json)
        for depth, dp_waves in [(d, 5) for d in sorted({2, 3} - {args.inflight})]:
            mapper.set_inflight(depth)
            keep, args.inflight = args.inflight, depth
            run_steps(depth)
            rec3 = {k: [] for k in rec}
            torch.cuda.synchronize(dev)
            t4 = time.perf_counter()
            run_steps(24, rec3)
            torch.cuda.synchronize(dev)
            dt4 = time.perf_counter() - t4
            args.inflight = keep
            lat_modes.append({"reads_per_batch": len(batches[0][1]), "batches_in_flight": depth, "dp_wavefronts_per_simd": dp_waves, "steps": 24,
                              "note": "24 steps incl. the fill and drain of the pipeline (one DP kernel's time extra: the rate of a long run is ~4 % higher)",
                              "p50_read_latency_ms": 1e3 * float(np.median(rec3["latency"])),
                              "bases_per_s": sum(rec3["mapped_bases"]) / dt4, "dp_kernel_ms_last": rec3["kern"][-1][0]})
        mapper.set_inflight(args.inflight)
        # one batch at a time: one round of the 5 120 resident wavefront slots is ~2 780 reads (1.84 alignments per read); then smaller
        lat_modes += latency_modes(mapper, ctx, batches, [(len(batches[0][1]), 3), (2780, 4), (1024, 6), (256, 8), (32, 8)])

    bases_timed = float(sum(rec["mapped_bases"]))
    elapsed, total_bases = clock.aggregate(elapsed, bases_timed)  # MAX over ranks, SUM over ranks

    if rank == 0:
        dp_ms = np.array([d for d, _ in rec["kern"]], np.float64)
        bt = float(np.mean([b for _, b in rec["kern"]]))
        cells = np.array([c for c, _ in rec["work"]], np.float64)
        alg = np.array([a for _, a in rec["work"]], np.float64)
        # algorithmic bytes of the DP launches (every candidate box the path aligned, counted by the library from the lengths) over
        # the time those launches took
        achieved = alg.sum() / (dp_ms.sum() * 1e-3) / 1e9
        dp = float(dp_ms.mean())
        cells_launch = float(cells.mean())
        # per-cell figures from the PMC passes of this very command (tools/pmc_traffic.py, tools/pmc_valu.py -> profiles/*.json)
        traffic, traffic_src, valu = None, None, {}
        try:
            sclk = ctx.last_dp_clock()  # (median MHz, min MHz, median wavefront ms) over the wavefronts of the last DP launches
        except Exception:
            sclk = None
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")), reverse=True):
            try:
                t = json.load(open(f))
                if "traffic_bytes_per_cell" in t:
                    traffic, traffic_src = t["traffic_bytes_per_cell"] * cells_launch, os.path.relpath(f, ROOT)
                    break
            except Exception:
                pass
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_valu.json")), reverse=True):
            try:
                t = json.load(open(f))
                if "valu_insts_per_cell" in t:
                    insts = t["valu_insts_per_cell"] * cells_launch
                    # MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on a SIMD-32; 1024 SIMDs at 2.4 GHz
                    # cycles per VALU instruction at the clock the kernel's own wavefronts measured (s_memtime / s_memrealtime stamps)
                    cyc = dp * 1e-3 * sclk[0] * 1e6 / (insts / 1024.0) if sclk else None
                    valu = {"valu_insts_per_cell": t["valu_insts_per_cell"], "salu_insts_per_cell": t.get("salu_insts_per_cell"),
                            "valu_insts_per_1024_cell_row": 1024 * t["valu_insts_per_cell"],
                            "sclk_cycles_per_valu_inst": cyc,
                            # profiles/r03_clock.md: a pure stream of the kernel's instruction class (v_pk_add/sub/max/min_*16, v_perm, v_bfi) issues at
                            # 4.30 (8 wavefronts per SIMD) .. 4.47 (5 per SIMD) shader cycles per wave-instruction per SIMD: the hardware's rate
                            "valu_issue_frac": (4.47 / cyc) if cyc else None,
                            "valu_util": insts * 2.0 / (1024 * (sclk[0] * 1e6 if sclk else 2.4e9) * dp * 1e-3),
                            "valu_util_note": "PMC SQ_INSTS_VALU of this kernel per cell (%s) x this run's cells per launch; sclk_cycles_per_valu_inst = kernel time x "
                                              "measured sclk / instructions per SIMD; valu_issue_frac = 4.47 cycles (pure stream of the same instruction class at 5 "
                                              "wavefronts per SIMD, tools/clock_probe.hip) / that; valu_util = against 2 cycles per wave64 instruction (SIMD-32 issue "
                                              "peak of MI355X_MICROARCH.md, which this class of instructions does not reach)" % os.path.relpath(f, ROOT),
                            "simd_cycles_per_valu_inst_pmc": t.get("simd_cycles_per_valu_inst"), "valu_active_frac_pmc": t.get("valu_active_frac")}
                    break
            except Exception:
                pass
        st = np.mean(np.array(rec["stages"]), axis=0)
        n_reads_step = int(np.mean([len(b[1]) for b in batches]))
        out = {
            "metric": "mapped bases/sec (whole node), HiFi map-hifi k19w19",
            "value": total_bases / elapsed,
            "unit": "bases/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": "int16",
            "data": "synthetic",
            "p50_read_latency_ms": 1e3 * float(np.median(rec["latency"])),
            "latency_mode": lat_modes,
            "config": {"workload": "BASELINE configs[3]: GDiet-LongReads -ax map-hifi k19 w19 -Z 10 -W 2 -i 0.2 -r 1000 ..., %d synthetic ~15 kbp HiFi reads in %d distinct "
                                   "batches vs synthetic reference of %.0f Mbp in 24 contigs (GRCh38-sized = 3088)" % (sum(len(b[1]) for b in batches), len(batches), args.ref_mbp),
                       "reads_per_step_per_gpu": n_reads_step, "bases_per_step_per_gpu": int(np.mean([b[2].sum() for b in batches])),
                       "distinct_batches": len(batches), "reads_timed_this_rank": int(sum(len(batches[i % len(batches)][1]) for i in range(args.steps))),
                       "mapped_fraction": float(sum(rec["mapped_reads"])) / max(1, sum(len(batches[i % len(batches)][1]) for i in range(args.steps))),
                       "alignments_per_step": float(np.mean(rec["aligns"])), "index_keys": int(mapper.n_keys()), "mid_occ": int(mapper.mid_occ),
                       "setup_s": {"reference": round(t_ref, 1), "reference_how": ref_how, "index_build_upload": round(t_index, 1), "reads_synth_upload": round(t_reads, 1)},
                       "stage_s_per_step": {"seed_kernel": st[0], "vote_kernel": st[1], "host_geometry": st[2], "gather_dp_backtrack": st[3],
                                            "host_postprocess": st[4], "other": st[5]},
                       "p50_read_latency_note": "median over the timed steps of (gdiet_hip_map_wait returned - gdiet_hip_map_submit called) of the read's batch; "
                                                "every read of a batch completes with its batch; p10 / p90: %.1f / %.1f ms"
                                                % (1e3 * float(np.percentile(rec["latency"], 10)), 1e3 * float(np.percentile(rec["latency"], 90))),
                       "batches_in_flight": args.inflight, "self_check": check, "with_upload": with_upload,
                       "parallelism": "reads sharded over %d GPU(s), index replicated, no collective" % world, "host_threads": host_threads,
                       "cpus_usable_on_node": pkg_cpus, "pipeline_lanes": args.lanes, "gpu_waits": os.environ.get("GDIET_SYNC", "spin")},
            "roofline": dict({"bound": "hbm", "kernel": "ksw_extd2_wave_kernel<64, 0, true>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                              "algorithmic_bytes_per_launch": float(alg.mean()), "dp_cells_per_launch": cells_launch,
                              "dp_cells_all_launches": int(all_cells[0]), "dp_launches": int(all_cells[1]),
                              "gcups": cells.sum() / (dp_ms.sum() * 1e-3) / 1e9, "kernel_ms": dp,
                              "kernel_ms_note": "mean over the K timed launches (HIP events on each launch's stream); the 64-lane kernel walks its own "
                                                "alignments back, so this is DP + backtrack; the other batches' seeding/voting kernels share the GPU",
                              "kernel_ms_alone": dp_alone, "backtrack_kernel_ms": bt,
                              "sclk_mhz": sclk[0] if sclk else None, "sclk_mhz_min": sclk[1] if sclk else None,
                              "sclk_note": "shader clock the DP kernel sustained: per wavefront ticks(s_memtime) / ticks(s_memrealtime) x 100 MHz around its DP rows, "
                                           "median / minimum over the wavefronts (gdiet_hip_last_dp_clock); 2 400 MHz is the part's peak: not throttled"}, **valu),
        }
        # N ranks share the node's CPUs: the host stages of a step must stay hidden behind its DP kernel
        if st[2] + st[4] > 0.8 * dp * 1e-3:
            out["config"]["host_bound_warning"] = ("host stages %.1f ms per step with %d threads against a %.1f ms DP kernel: this rank is host-bound"
                                                   % (1e3 * (st[2] + st[4]), host_threads, dp))
        if not args.no_cpu_baseline and world == 1:  # the CPU baseline is timed on rank 0 at N = 1 only
            port_reads = [np.searchsorted(BASES, np.frombuffer(s, np.uint8)).clip(0, 3).astype(np.uint8) for _, s in batches[0][1][:64]]
            try:  # the reference binary itself where it travelled with the repo (and runs on this host's CPU) ...
                if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "gdiet_lr_avx")):
                    raise FileNotFoundError("oracle/_ref/gdiet_lr_avx")
                try:
                    out["cpu_baseline"] = cpu_baseline_reference(names, contigs, np.random.default_rng(99), pkg_cpus, args.cpu_baseline_ref, batches[0][1])
                except Exception as ex_whole:  # e.g. no room for the 4 GB index file: the chr1-sized form
                    if args.cpu_baseline_ref != "whole":
                        raise
                    out["cpu_baseline"] = cpu_baseline_reference(names, contigs, np.random.default_rng(99), pkg_cpus, "largest")
                    out["cpu_baseline"]["sample"] += " (whole-reference baseline failed: %r)" % (ex_whole,)
            except Exception as ex:  # ... else the DP stage of the oracle port; the baseline must never take the benchmark line down
                try:
                    out["cpu_baseline"] = cpu_baseline_port(port_reads)
                    out["cpu_baseline"]["sample"] += " (reference binary unavailable: %r)" % (ex,)
                except Exception as ex2:
                    out["cpu_baseline"] = {"value": None, "unit": "mapped bases/s", "cores": 0, "kind": "port", "sample": "failed: %r / %r" % (ex, ex2)}
            if out["cpu_baseline"].get("value"):
                out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]  # (vs_baseline stays null: BASELINE.md publishes no number in this metric)
            # parity at the benchmark's scale (outside the timed region): the reference's SAM records for the shared reads, against the
            # whole GRCh38-sized index, diffed line by line with gdiet_hip_sam_batch of the GPU path's records for the same reads
            ref_sam, ref_reads = out["cpu_baseline"].pop("_sam_body", None), out["cpu_baseline"].pop("_reads", None)
            if ref_sam is not None and ref_reads:
                out["config"]["parity_at_scale"] = parity_at_scale(mapper, ref_reads, ref_sam)
            if out["cpu_baseline"].get("p50_read_latency_ms"):
                out["p50_read_latency_vs_cpu"] = out["p50_read_latency_ms"] / out["cpu_baseline"]["p50_read_latency_ms"]
        print(json.dumps(out))
    for b in batches:
        mapper.free_batch(b[0])
    mapper.close()
    if world > 1:
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
