"""Matcher.run on a long list of host-parsed molecules (pipelined batches) vs one batch at a time. Not a test."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import jess_run, synthetic, template  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
matcher = jess_run.Matcher(template.load_packed_library())
base = synthetic.structures(min(n, 1000))
mols = [base[i % len(base)].copy() for i in range(n)]
matcher.run(mols[:2000])
for size in (1000, 2000, 4000):
    for rep in range(3):
        t0 = time.perf_counter()
        out = matcher.run(mols, batch_size=size)
        dt = time.perf_counter() - t0
        print(f"Matcher.run, {n} molecules in batches of {size}, rep {rep}: {n / dt:.0f} structures/s "
              f"({sum(len(v) for v in out.values())} matches)", flush=True)
