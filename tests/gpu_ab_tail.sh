#!/bin/bash
# A/B of a finer tail of the queued block search (ENZYMM_AMD_TAIL: the structures holding the last fraction of the estimated
# cost get twice the rounds).  Not a test.
run() {
    env "$@" python3 bench.py --no-extras --no-cpu-baseline --streamed-passes 0 --sustained-seconds 3 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']['exclusive']
print('$*', 'value %.0f sustained %.0f k_search %.3f ms' % (d['value'], d['config']['sustained_structures_per_s'], r['k_search_ms']))"
}
run ENZYMM_AMD_TAIL=0
run ENZYMM_AMD_TAIL=0.05
run ENZYMM_AMD_TAIL=0.1
run ENZYMM_AMD_TAIL=0.2
run ENZYMM_AMD_TAIL=0.3
run ENZYMM_AMD_TAIL=0
