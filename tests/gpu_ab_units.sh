#!/bin/bash
# A/B of the queued block search's unit granularity on the headline batch: chunk classes (ENZYMM_AMD_CX) and work items per
# class (ENZYMM_AMD_UNITS).  Prints sustained structures/s and the exclusive k_search time per setting.  Not a test.
run() {
    env "$@" python3 bench.py --no-extras --no-cpu-baseline --streamed-passes 0 --sustained-seconds 3 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']['exclusive']
print('$*', 'value %.0f sustained %.0f k_search %.3f ms index %.3f' % (d['value'], d['config']['sustained_structures_per_s'], r['k_search_ms'], r['index_ms']))"
}
run ENZYMM_AMD_CX=1
run ENZYMM_AMD_CX=2
run ENZYMM_AMD_CX=0
run ENZYMM_AMD_CX=1 ENZYMM_AMD_UNITS=1536
run ENZYMM_AMD_CX=1 ENZYMM_AMD_UNITS=2048
run ENZYMM_AMD_CX=1 ENZYMM_AMD_UNITS=3072
run ENZYMM_AMD_CX=2 ENZYMM_AMD_UNITS=1024
run ENZYMM_AMD_CX=2 ENZYMM_AMD_UNITS=1536
run ENZYMM_AMD_CX=1
