#!/usr/bin/env python3
"""Condense rocprofv3 --pmc counter_collection.csv files into one JSON per kernel.

usage: summarize_pmc.py <kernel substring> <out.json> [--searches N] [--kernel-ms T] <dir with counter csvs> [<dir> ...]

Each directory is one separate `rocprofv3 --pmc ...` pass (FETCH_SIZE and WRITE_SIZE never share a pass:
MI355X_MICROARCH.md, rocprofv3 PMC slots).  Values are means per launch of the matching kernel.
--searches: (structure, template) pairs one launch searches -> instructions per search.
--kernel-ms: the kernel's mean duration from the --kernel-trace --stats run of the same command -> share of the VALU
issue slots used.  The ceiling is MEASURED when --valu-ceiling names the JSON of profiles/valu_probe.hip (time one
SIMD needs per independent wave64 v_fma_f32 with 4 waves issuing -- k_search's occupancy); without it, the guide's
figure: 1024 SIMDs x 2.4 GHz x one wave64 VALU instruction per 2 cycles (MI355X_MICROARCH.md).
--head: git revision the profiled library was built from (recorded in the JSON)."""
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
ap.add_argument("--valu-ceiling", default="")
ap.add_argument("--head", default="")
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
if args.head:
    summary["git_head"] = args.head
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
if mean("TCC_HIT_sum") is not None and mean("TCC_MISS_sum") is not None:
    h, m = mean("TCC_HIT_sum"), mean("TCC_MISS_sum")
    summary["l2"] = {"hits": h, "misses": m, "hit_rate": round(h / (h + m), 4) if h + m else None,
                     "note": "requests that miss L2 go to the fabric: Infinity Cache (256 MiB) first, HBM behind it; "
                             "FETCH_SIZE counts both (MI355X_MICROARCH.md, HBM)"}
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
    secondary["valu_ceiling"] = "MI355X_MICROARCH.md: 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU instruction"
    if args.valu_ceiling:
        try:
            probe = json.load(open(args.valu_ceiling))
            ns = probe["v_fma_f32"]["4"]["ns_per_instruction_per_simd"]
            slots = probe["simds"] * args.kernel_ms * 1e6 / ns
            secondary["valu_ceiling"] = (f"measured ({args.valu_ceiling.split('/')[-1]}): {ns:.4f} ns per wave64 v_fma_f32 per SIMD "
                                         f"at 4 waves/SIMD x {probe['simds']} SIMDs")
        except (OSError, KeyError, ValueError):
            pass
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
