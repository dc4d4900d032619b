#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into the per-launch instruction-issue figures of one kernel (bench.py's roofline.valu_*).

usage: pmc_valu.py KERNEL_SUBSTRING OUT.json PASS_DIR [PASS_DIR ...] [--bench-log FILE]
Each PASS_DIR holds the *_counter_collection.csv of one
    rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d PASS_DIR -- python3 bench.py ...
    rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d PASS_DIR -- python3 bench.py ...
run (program directly after `--`).  Units, per /opt/skills/guides/MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* count
quad-cycles; GRBM_GUI_ACTIVE is summed over the 8 XCDs, so the kernel lasted GRBM_GUI_ACTIVE / 8 shader cycles.
--bench-log: the output of one of the profiled bench.py runs; its dp_cells per launch is stored beside the counters, which is how
bench.py recognises a measurement of the same launch."""
import collections
import csv
import glob
import json
import os
import sys

N_SIMD = 256 * 4  # MI355X: 256 CUs x 4 SIMDs


def main():
    args = sys.argv[1:]
    bench_log = None
    if "--bench-log" in args:
        i = args.index("--bench-log")
        bench_log = args[i + 1]
        del args[i:i + 2]
    kern, out, dirs = args[0], args[1], args[2:]
    tot, calls = collections.Counter(), collections.Counter()
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                if kern in r["Kernel_Name"]:
                    tot[r["Counter_Name"]] += float(r["Counter_Value"])
                    calls[r["Counter_Name"]] += 1
    per = {k: tot[k] / calls[k] for k in tot}
    res = {"kernel": kern, "launches_seen": dict(calls), "per_launch_raw": per}
    if "GRBM_GUI_ACTIVE" in per:
        cyc = per["GRBM_GUI_ACTIVE"] / 8.0
        res["kernel_shader_cycles"] = cyc
        if "SQ_INSTS_VALU" in per:
            res["simd_cycles_per_valu_inst"] = cyc * N_SIMD / per["SQ_INSTS_VALU"]
            # MI355X_MICROARCH.md (Wave scheduling): a wave64 VALU instruction issues over 2 cycles on a SIMD-32
            res["valu_util_vs_simd32_issue_peak"] = per["SQ_INSTS_VALU"] * 2.0 / (cyc * N_SIMD)
        if "SQ_ACTIVE_INST_VALU" in per:
            res["valu_active_frac"] = per["SQ_ACTIVE_INST_VALU"] * 4.0 / (cyc * N_SIMD)  # quad-cycles -> cycles, per SIMD
        if "SQ_WAVE_CYCLES" in per:
            res["mean_resident_waves"] = per["SQ_WAVE_CYCLES"] * 4.0 / cyc
    if bench_log:
        for line in reversed(open(bench_log).read().splitlines()):
            if line.startswith("{") and '"roofline"' in line:
                r = json.loads(line)["roofline"]
                # every launch of the process was counted by the profiler; the bench line carries the cells of all of them
                cells, n_l = r["dp_cells_all_launches"], r["dp_launches"]
                res["dp_cells_all_launches"], res["dp_launches"] = cells, n_l
                for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU"):
                    if c in tot:
                        assert calls[c] == n_l, "the profile saw %d launches, the bench made %d" % (calls[c], n_l)
                if "SQ_INSTS_VALU" in tot:
                    res["valu_insts_per_cell"] = tot["SQ_INSTS_VALU"] / cells
                    res["valu_insts_per_1024_cell_row"] = 1024 * tot["SQ_INSTS_VALU"] / cells
                if "SQ_INSTS_SALU" in tot:
                    res["salu_insts_per_cell"] = tot["SQ_INSTS_SALU"] / cells
                break
    res["note"] = ("wave-level instruction counts (one VALU instruction = 64 lane-operations); SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* in quad-cycles, "
                   "GRBM_GUI_ACTIVE summed over 8 XCDs (MI355X_MICROARCH.md, constants table / DVFS section)")
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
