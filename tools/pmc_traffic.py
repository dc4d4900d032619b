#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes into the per-launch HBM traffic of one kernel (bench.py's roofline.traffic).

usage: pmc_traffic.py KERNEL_SUBSTRING OUT.json PASS_DIR [PASS_DIR ...] [--bench-log FILE]
(--bench-log: the output of one of the profiled bench.py runs; its dp_cells / algorithmic bytes per launch are stored beside the
traffic, which is how bench.py recognises a measurement of the same launch)
Each PASS_DIR holds the *_counter_collection.csv of one `rocprofv3 --pmc X --output-format csv -d PASS_DIR -- python3 bench.py ...`
run (WRITE_SIZE and FETCH_SIZE need separate passes: TCC has 4 slots, they cost 2 + 3).
Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KiB; on gfx950 FETCH_SIZE reports half
of the bytes of wide coalesced streaming reads, so the read side is doubled (an upper bound for narrow reads)."""
import collections
import csv
import glob
import json
import os
import sys


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
    wr = per.get("WRITE_SIZE", 0.0) * 1024
    rd = per.get("FETCH_SIZE", 0.0) * 1024 * 2
    res = {"kernel": kern, "launches_seen": dict(calls), "per_launch_raw": per, "write_bytes_per_launch": wr,
           "fetch_bytes_per_launch_x2_corrected": rd, "traffic_bytes_per_launch": wr + rd,
           "note": "WRITE_SIZE/FETCH_SIZE in KiB; FETCH_SIZE doubled (gfx950 correction, MI355X_MICROARCH.md HBM section)"}
    if bench_log:
        for line in reversed(open(bench_log).read().splitlines()):
            if line.startswith("{") and '"roofline"' in line:
                r = json.loads(line)["roofline"]
                res["dp_cells_per_launch"], res["algorithmic_bytes_per_launch"] = r["dp_cells_per_launch"], r["algorithmic_bytes_per_launch"]
                if "dp_cells_all_launches" in r and calls:  # every launch of the process was counted: bytes per DP cell
                    n_l = max(calls.values())
                    assert n_l == r["dp_launches"], "the profile saw %d launches, the bench made %d" % (n_l, r["dp_launches"])
                    res["traffic_bytes_per_cell"] = (tot.get("WRITE_SIZE", 0.0) * 1024 + tot.get("FETCH_SIZE", 0.0) * 1024 * 2) / r["dp_cells_all_launches"]
                break
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
