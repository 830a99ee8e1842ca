#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection.csv files into one JSON per kernel.

usage: summarize_pmc.py <kernel substring> <out.json> [--searches N] [--kernel-ms T] <dir with counter csvs> [<dir> ...]

Each directory is one separate `rocprofv3 --pmc ...` pass (FETCH_SIZE and WRITE_SIZE never share a pass:
MI355X_MICROARCH.md, rocprofv3 PMC slots).  Values are means per launch of the matching kernel.
--searches: (structure, template) pairs one launch searches -> instructions per search.
--kernel-ms: the kernel's mean duration from the --kernel-trace --stats run of the same command -> share of the VALU
issue slots used (1024 SIMDs x 2.4 GHz x one wave64 VALU instruction per 2 cycles, MI355X_MICROARCH.md)."""
import argparse
import collections
import csv
import glob
import json

ap = argparse.ArgumentParser()
ap.add_argument("kernel")
ap.add_argument("out")
ap.add_argument("dirs", nargs="+")
ap.add_argument("--searches", type=float, default=0.0)
ap.add_argument("--kernel-ms", type=float, default=0.0)
args = ap.parse_args()

counters = {}
for d in args.dirs:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if args.kernel in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            counters[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
summary = {"kernel": args.kernel, "counters": counters}
mean = lambda name: counters[name]["mean_per_launch"] if name in counters else None
if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
    f, w = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    summary["hbm_traffic_bytes_per_launch"] = {
        "fetch_raw": f * 1024, "fetch_corrected_x2": 2 * f * 1024, "write": w * 1024,
        "total_corrected": (2 * f + w) * 1024,
        "note": "FETCH_SIZE/WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reads 1/2 of a wide coalesced stream "
                "(MI355X_MICROARCH.md, HBM), so the corrected value doubles it; the access-width mix of this kernel is "
                "uncalibrated, the true value lies between raw and corrected",
    }
# the informative ceiling of this kernel is instruction issue, not HBM: instructions per search and VALU issue share
secondary = {}
kinds = {"valu": "SQ_INSTS_VALU", "salu": "SQ_INSTS_SALU", "branch": "SQ_INSTS_BRANCH", "lds": "SQ_INSTS_LDS",
         "vmem_rd": "SQ_INSTS_VMEM_RD", "smem": "SQ_INSTS_SMEM"}
if args.searches and mean("SQ_INSTS_VALU") is not None:
    secondary["searches_per_launch"] = args.searches
    secondary["instructions_per_search"] = {k: round(mean(c) / args.searches, 1) for k, c in kinds.items()
                                            if mean(c) is not None}
if args.kernel_ms and mean("SQ_INSTS_VALU") is not None:
    slots = 1024 * 2.4e9 * 0.5 * args.kernel_ms * 1e-3
    secondary["kernel_ms"] = args.kernel_ms
    secondary["valu_issue_frac"] = round(mean("SQ_INSTS_VALU") / slots, 4)
    if mean("SQ_THREAD_CYCLES_VALU") is not None and mean("SQ_ACTIVE_INST_VALU"):
        # lanes active in the VALU instructions that issued (both counters in the same units)
        secondary["valu_lane_utilisation"] = round(mean("SQ_THREAD_CYCLES_VALU") / (64.0 * mean("SQ_ACTIVE_INST_VALU")), 4)
if mean("SQ_WAIT_ANY") is not None and mean("SQ_WAVE_CYCLES"):
    secondary["wave_cycles_waiting_frac"] = round(mean("SQ_WAIT_ANY") / mean("SQ_WAVE_CYCLES"), 4)
if secondary:
    summary["secondary"] = secondary
json.dump(summary, open(args.out, "w"), indent=1)
print(json.dumps({"hbm": summary.get("hbm_traffic_bytes_per_launch", {}), "secondary": secondary}))
