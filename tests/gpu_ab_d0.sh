run() {
    ENZYMM_AMD_LIB=$1 python3 bench.py --no-extras --no-cpu-baseline --streamed-passes 0 --sustained-seconds 3 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d['roofline']['exclusive']
print('$1', 'value %.0f sustained %.0f k_search %.3f ms hits %d digest %s' % (d['value'], d['config']['sustained_structures_per_s'], r['k_search_ms'], d['config']['hits_total'], d['config']['hit_table_digest']))"
}
run enzymm_amd/libenzymm_amd.so
run enzymm_amd/libenzymm_amd_d0u2.so
run enzymm_amd/libenzymm_amd.so
run enzymm_amd/libenzymm_amd_d0u2.so
