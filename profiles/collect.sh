#!/bin/bash
# Profile session of the headline bench on the GPU box: run from the repo root as
#     bash profiles/collect.sh <tag>            (e.g. r02_v14)
# Writes gpurun_out/<tag>/: the bench line, rocprofv3 --kernel-trace --stats of the same command, and separate --pmc
# passes (SQ counters in groups of <= 8, FETCH_SIZE and WRITE_SIZE each alone -- MI355X_MICROARCH.md, PMC slots)
# condensed by summarize_pmc.py.  Copy <tag>_* into profiles/ and point profiles/LATEST at them.
# The box has no .git: profiles/run_collect.sh (run in the build container) refuses a dirty tree, writes the revision to
# .build_head (git-ignored, travels with the snapshot) and calls this script through gpurun; every JSON written here
# carries that revision, and this script refuses to run without it.
set -u
tag=${1:?tag}
head=$(cat .build_head 2>/dev/null || true)
if [ -z "$head" ]; then echo "no .build_head: start this through profiles/run_collect.sh" >&2; exit 2; fi
out=gpurun_out/$tag
mkdir -p "$out"
echo "$head" > "$out/${tag}_git_head.txt"
python3 tests/gpu_valu_ceiling.py "$tag" > "$out/valu.log" 2>&1 && mv "gpurun_out/${tag}_valu_ceiling.json" "$out/"
export TMPDIR=/tmp ENZYMM_AMD_SYNTH_CACHE=/tmp/synth
timeout 600 python3 bench.py --steps 10 --warmup 2 > "$out/${tag}_bench1000.json" 2> "$out/bench.err"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --streamed-passes 0 --sustained-seconds 0 > "$out/trace.log" 2>&1
stats=$(find "$out/trace" -name "*kernel_stats.csv" | head -1)
[ -n "$stats" ] && cp "$stats" "$out/${tag}_bench1000_kernel_stats.csv"
# the same with ONE search in flight: kernel durations with nothing else on the GPU (roofline.exclusive, and the
# duration the per-search instruction figures of summarize_pmc.py are set against)
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace1" -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras --streamed-passes 0 --sustained-seconds 0 --pipeline 1 > "$out/trace1.log" 2>&1
stats=$(find "$out/trace1" -name "*kernel_stats.csv" | head -1)
[ -n "$stats" ] && cp "$stats" "$out/${tag}_bench1000_pipeline1_kernel_stats.csv"
pass() { timeout 200 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_$1" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --streamed-passes 0 --sustained-seconds 0 --pipeline 1 > "$out/pmc_$1.log" 2>&1; }
pass SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT
pass SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU
pass FETCH_SIZE
pass WRITE_SIZE
pass TCC_HIT_sum TCC_MISS_sum
# one summary per kernel: the two launches of a search (block search = k_search<..., 0>, depth-first part = k_deep =
# k_search<..., 2>) and the table build
for spec in "k_search:k_search<true, true, 0>" "k_deep:k_search<true, true, 2>" "k_tab_build:k_tab_build"; do
    short=${spec%%:*}
    kernel=${spec#*:}
    kms=0
    if [ -f "$out/${tag}_bench1000_pipeline1_kernel_stats.csv" ]; then
        kms=$(python3 - "$out/${tag}_bench1000_pipeline1_kernel_stats.csv" "$kernel" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Name"]]
print(float(rows[0]["AverageNs"]) / 1e6 if rows else 0.0)
PY
)
    fi
    python3 profiles/summarize_pmc.py "$kernel" "$out/${tag}_${short}_pmc.json" --searches 6780000 --kernel-ms "$kms" \
        --valu-ceiling "$out/${tag}_valu_ceiling.json" --head "$head" \
        "$out"/pmc_SQ_INSTS_VALU "$out"/pmc_SQ_ACTIVE_INST_ANY "$out"/pmc_SQ_INSTS_BRANCH "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE "$out"/pmc_TCC_HIT_sum >> "$out/summary.log" 2>&1
done
# The kept bench JSON is printed AFTER profiles/LATEST (on this box) points at this very PMC set, so that the traffic and
# the secondary figures inside it are its own (the set above was collected with an earlier LATEST)
for short in k_search k_deep k_tab_build; do cp "$out/${tag}_${short}_pmc.json" profiles/ 2>/dev/null; done
cp "$out/${tag}_valu_ceiling.json" profiles/ 2>/dev/null
python3 - "$tag" "$head" > "$out/LATEST" <<'PY'
import json, sys
tag, head = sys.argv[1], sys.argv[2]
print(json.dumps({"git_head": head, "k_search_pmc": f"{tag}_k_search_pmc.json", "k_deep_pmc": f"{tag}_k_deep_pmc.json",
                  "k_tab_build_pmc": f"{tag}_k_tab_build_pmc.json", "kernel_stats": f"{tag}_bench1000_kernel_stats.csv",
                  "kernel_stats_one_in_flight": f"{tag}_bench1000_pipeline1_kernel_stats.csv",
                  "valu_ceiling": f"{tag}_valu_ceiling.json", "bench": f"{tag}_bench1000.json",
                  "parity_census": [f"{tag}_parity_census_1000.json", f"{tag}_parity_census_real256.json", f"{tag}_parity_census_2000_from_1000.json", f"{tag}_parity_census_config5.json"]}, indent=1))
PY
cp "$out/LATEST" profiles/LATEST
mv "$out/${tag}_bench1000.json" "$out/${tag}_bench1000_before_pmc.json"
timeout 900 python3 bench.py --steps 10 --warmup 2 > "$out/${tag}_bench1000.json" 2> "$out/bench2.err"
# The whole-batch parity census on the SAME build, in the SAME call: 1000 x 6780 of the headline batch,
# a 64-structure sample of the real-structure batch, configs[4] at the default cutoffs -- each stamped with .build_head
timeout 900 python3 tests/gpu_parity_census.py 1000 0 synthetic > "$out/census1000.log" 2>&1 && mv gpurun_out/parity_census.json "$out/${tag}_parity_census_1000.json"
timeout 600 python3 tests/gpu_parity_census.py 256 0 real > "$out/census_real.log" 2>&1 && mv gpurun_out/parity_census.json "$out/${tag}_parity_census_real256.json"
# (structures the bench never sees: seeds 1000 .. 2999 of the same generator)
timeout 900 python3 tests/gpu_parity_census.py 2000 1000 synthetic > "$out/census2000.log" 2>&1 && mv gpurun_out/parity_census.json "$out/${tag}_parity_census_2000_from_1000.json"
timeout 900 python3 tests/gpu_parity_census.py 10 0 config5 > "$out/census_c5.log" 2>&1 && mv gpurun_out/parity_census.json "$out/${tag}_parity_census_config5.json"
# (the raw traces stay on the box: gpurun copies back 64 MiB at most)
rm -rf "$out"/trace "$out"/trace1
find "$out" -name "*.csv" -path "*pmc_*" -size +4M -delete
cat "$out/${tag}_bench1000.json" "$out/summary.log" "$out"/${tag}_parity_census_*.json
[ -f "$out/${tag}_bench1000_kernel_stats.csv" ] && head -8 "$out/${tag}_bench1000_kernel_stats.csv"
