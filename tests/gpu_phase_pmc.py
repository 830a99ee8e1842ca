"""Per-phase instruction counters: parse a rocprofv3 --pmc run of tests/gpu_ablate.py. Not a test.

usage: gpu_phase_pmc.py <dir with counter csv> [searches]
gpu_ablate.py launches k_search 4x per mode in the order full / no dfs / no subtrees / staging only."""
import collections
import csv
import glob
import sys

d = sys.argv[1]
searches = float(sys.argv[2]) if len(sys.argv) > 2 else 6.78e6
rows = collections.defaultdict(dict)
for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_search" in r["Kernel_Name"]:
            key = int(r["Dispatch_Id"])
            rows[key][r["Counter_Name"]] = rows[key].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(rows)
labels = ["full", "no dfs", "no subtrees", "staging only"]
per = len(ids) // 4
for m, label in enumerate(labels):
    sel = ids[m * per:(m + 1) * per]
    names = sorted(rows[sel[0]])
    print(label, {n: round(sum(rows[i][n] for i in sel) / len(sel) / searches, 2) for n in names})
