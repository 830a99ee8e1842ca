"""Latency of Matcher.run on ONE structure (BASELINE.json configs[1]): wall clock, phase trace, cProfile. Not a test."""
import cProfile
import os
import pstats
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import helpers  # noqa: E402
from enzymm_amd import jess_run, pyjess, template  # noqa: E402

matcher = jess_run.Matcher(template.load_packed_library())
mol = pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb")
for _ in range(5):
    matcher.run([mol])
walls = []
for _ in range(20):
    t0 = time.perf_counter()
    out = matcher.run([mol])
    walls.append(time.perf_counter() - t0)
print(f"Matcher.run([AF-P0DUB6]): min {1e3 * min(walls):.2f} ms, median {1e3 * sorted(walls)[10]:.2f} ms, "
      f"{sum(len(v) for v in out.values())} matches")
prof = cProfile.Profile()
prof.enable()
for _ in range(20):
    matcher.run([mol])
prof.disable()
pstats.Stats(prof).sort_stats("tottime").print_stats(18)
