#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection.csv files into one JSON per kernel.

usage: summarize_pmc.py <kernel substring> <out.json> <dir with counter csvs> [<dir> ...]
Each directory is one separate `rocprofv3 --pmc ...` pass (FETCH_SIZE and WRITE_SIZE never share a pass:
MI355X_MICROARCH.md, rocprofv3 PMC slots).  Values are means per launch of the matching kernel."""
import collections
import csv
import glob
import json
import sys

kernel, out_path, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
counters = {}
for d in dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            counters[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
summary = {"kernel": kernel, "counters": counters}
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    f, w = counters["FETCH_SIZE"]["mean_per_launch"], counters["WRITE_SIZE"]["mean_per_launch"]
    summary["hbm_traffic_bytes_per_launch"] = {
        "fetch_raw": f * 1024, "fetch_corrected_x2": 2 * f * 1024, "write": w * 1024,
        "total_corrected": (2 * f + w) * 1024,
        "note": "FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream "
                "(MI355X_MICROARCH.md, HBM), so the corrected value doubles it; the access-width mix of this kernel is "
                "uncalibrated, the true value lies between raw and corrected",
    }
json.dump(summary, open(out_path, "w"), indent=1)
print(json.dumps(summary.get("hbm_traffic_bytes_per_launch", {})))
