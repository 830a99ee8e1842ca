#!/bin/bash
# A/B of builds of the library on the headline batch: bash tests/gpu_ab_libs.sh <rounds> <spec> ...
# spec = <lib suffix>[@ENV=VALUE[,ENV=VALUE...]]; suffix '' = enzymm_amd/libenzymm_amd.so, x = enzymm_amd/libenzymm_amd_x.so
# (make -C enzymm_amd/csrc variant NAME=x EXTRA=...).  Prints value, sustained rate, exclusive k_search time, index time,
# hit count and digest per run.  Not a test.
rounds=$1; shift
run() {
    spec=$1
    name=${spec%%@*}
    envs=""
    case "$spec" in *@*) envs=$(echo "${spec#*@}" | tr ',' ' ');; esac
    lib=enzymm_amd/libenzymm_amd${name:+_$name}.so
    env $envs ENZYMM_AMD_LIB=$PWD/$lib python3 bench.py --no-extras --no-cpu-baseline --streamed-passes 0 --sustained-seconds 3 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']['exclusive']
print('%-28s' % '${spec:-default}', 'value %.0f sustained %.0f k_search %.3f ms index %.3f hits %d digest %s' % (d['value'], d['config']['sustained_structures_per_s'], r['k_search_ms'], r['index_ms'], d['config']['hits_total'], d['config']['hit_table_digest']))"
}
for i in $(seq $rounds); do for v in "$@"; do run "$v"; done; done
