#!/bin/bash
# Quick SQ-counter passes over tests/gpu_time_search.py (the bench batch, one search in flight), summarised per kernel:
#     bash profiles/pmc_quick.sh <tag>          (run on the GPU box, from the repo root)
set -u
tag=${1:?tag}
script=${3:-tests/gpu_time_search.py 1000}
kernels=${4:-"k_search<true, true, 0>|k_search<true, true, 2>|k_tab_build"}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp ENZYMM_AMD_SYNTH_CACHE=/tmp/synth
pass() { timeout 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$out/pmc_$1" -o run -- python3 $script > "$out/pmc_$1.log" 2>&1; }
pass SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT
pass SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_FLAT SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM
[ "${2:-}" = "mem" ] && { pass FETCH_SIZE; pass WRITE_SIZE; pass TCC_HIT_sum TCC_MISS_sum; }
IFS='|' read -ra klist <<< "$kernels"
for kernel in "${klist[@]}"; do
    name=$(echo "$kernel" | tr -dc 'a-z0-9_')
    python3 profiles/summarize_pmc.py "$kernel" "$out/${tag}_${name}_pmc.json" --searches 6780000 \
        "$out"/pmc_SQ_INSTS_VALU "$out"/pmc_SQ_ACTIVE_INST_ANY "$out"/pmc_SQ_INSTS_BRANCH "$out"/pmc_FETCH_SIZE "$out"/pmc_WRITE_SIZE "$out"/pmc_TCC_HIT_sum 2>/dev/null | head -c 1500
    echo
    python3 - "$out/${tag}_${name}_pmc.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))["counters"]
g = lambda k: d.get(k, {}).get("mean_per_launch", float("nan"))
wc = g("SQ_WAVE_CYCLES")
print({k: round(g(k) / wc, 4) for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_WAIT_INST_LDS")},
      "waves", g("SQ_WAVES"), "wave_cycles(x4)", wc, "bank_conflict/lds_active", round(g("SQ_LDS_BANK_CONFLICT") / max(g("SQ_ACTIVE_INST_LDS"), 1), 3))
PY
done
