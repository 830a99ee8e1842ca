#!/bin/bash
# A/B of the size of the block search's workgroup queue (TM_SHARED_TRIPLES: 320 entries of 56 B as shipped): fewer entries
# leave more LDS for a structure's kind lists (a 4 079-atom structure's lists fit or do not).  Not a test.
run() {
    lib=$1; gen=$2
    ENZYMM_AMD_LIB=$lib python3 bench.py --generator $gen --no-extras --no-cpu-baseline --streamed-passes 0 --sustained-seconds 3 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']['exclusive']
print('$lib $gen', 'value %.0f sustained %.0f k_search %.3f k_deep %.3f ms' % (d['value'], d['config']['sustained_structures_per_s'], r['k_search_ms'], r['k_deep_ms']))"
}
for gen in v2 real; do
    run enzymm_amd/libenzymm_amd.so $gen
    run enzymm_amd/libenzymm_amd_st256.so $gen
    run enzymm_amd/libenzymm_amd_st192.so $gen
done
