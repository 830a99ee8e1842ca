"""cProfile of the main thread in Matcher.run on a long list of host-parsed molecules (the consumer side of the pipelined
batches): where the Python time per batch goes once the GPU step is 2.5 ms.  Not a test.

    python tests/gpu_run_profile.py [molecules]"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import jess_run, synthetic, template  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
matcher = jess_run.Matcher(template.load_packed_library())
base = synthetic.structures(1000)
mols = [base[i % len(base)].copy() for i in range(n)]
matcher.run(mols[:2000])
for rep in range(2):
    t0 = time.perf_counter()
    matcher.run(mols)
    print("Matcher.run, %d molecules: %.0f structures/s" % (n, n / (time.perf_counter() - t0)), flush=True)
prof = cProfile.Profile()
t0 = time.perf_counter()
prof.enable()
matcher.run(mols)
prof.disable()
print("under the profiler: %.0f structures/s" % (n / (time.perf_counter() - t0)))
stats = pstats.Stats(prof)
stats.sort_stats("cumulative").print_stats(28)
stats.sort_stats("tottime").print_stats(18)
