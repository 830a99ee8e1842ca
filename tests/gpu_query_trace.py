"""Matcher.run over 12 000 molecules with ENZYMM_AMD_TRACE=1 in the environment: the host-side milliseconds of every tm_query and
tm_batch_create_from_structures call, phase by phase, on stderr.  Not a test.

    ENZYMM_AMD_TRACE=1 python tests/gpu_query_trace.py 2>&1 | grep -E "^tm_query|^tm_batch_create"
"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from enzymm_amd import jess_run, synthetic, template
matcher = jess_run.Matcher(template.load_packed_library())
base = synthetic.structures(1000)
mols = [base[i % len(base)].copy() for i in range(12000)]
matcher.run(mols[:2000])
t0 = time.perf_counter(); matcher.run(mols); print("rate", 12000 / (time.perf_counter() - t0), file=sys.stderr)
