"""Per-phase instruction counters: summarise a rocprofv3 --pmc --output-format csv run of tests/gpu_ablate.py. Not a test.

usage: gpu_phase_pmc.py <dir with *counter_collection.csv> [searches per launch]
gpu_ablate.py launches k_search 4x per mode in the order full / no dfs / no level-3 look / pair stage only / set-up only."""
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
            did = int(r["Dispatch_Id"])
            rows[did][r["Counter_Name"]] = rows[did].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
ids = sorted(rows)
labels = ["full", "no dfs", "no level-3 look", "pair stage only", "set-up only"]
per = len(ids) // len(labels)
prev = None
for m, label in reversed(list(enumerate(labels))):
    sel = ids[m * per:(m + 1) * per]
    cur = {n: sum(rows[i][n] for i in sel) / len(sel) / searches for n in sorted(rows[sel[0]])}
    delta = {n: round(cur[n] - (prev[n] if prev else 0.0), 1) for n in cur}
    print("%-13s total %s" % (label, {n: round(v, 1) for n, v in cur.items()}))
    print("%-13s added %s" % ("", delta))
    prev = cur
