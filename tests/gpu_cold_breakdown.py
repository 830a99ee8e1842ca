"""Where the first sweep of a process spends its constant: per size bucket the time to pack the template list for the device
(Python), to create the device library (tm_library_create), the first query of one molecule (selection staged, table plan)
and a query of 255 more.  Not a test."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, template, _native  # noqa: E402

lib = template.load_packed_library()
templates = [lib[i] for i in range(len(lib))]
by_size = {}
for t in templates:
    if t.effective_size >= 3:
        by_size.setdefault(int(t.effective_size), []).append(t)
mols = synthetic.structures(256)
pyjess.Jess(by_size[3][:50]).query_batch(mols[:2], 2.0, 1.0, 1.0, max_candidates=10000, ignore_chain=True, return_raw=True)  # context, pools
tot = [0.0, 0.0, 0.0, 0.0]
for size in sorted(by_size, reverse=True):
    tpls, d = by_size[size], helpers.DEFAULT_DISTANCE[min(size, 8)]
    t0 = time.perf_counter()
    jess = pyjess.Jess(tpls)
    t1 = time.perf_counter()
    packed = pyjess._packed(jess._templates, jess._device)
    t2 = time.perf_counter()
    jess.query_batch(mols[:1], 2.0, d, d, max_candidates=10000, ignore_chain=True, return_raw=True)
    t3 = time.perf_counter()
    jess.query_batch(mols[1:], 2.0, d, d, max_candidates=10000, ignore_chain=True, return_raw=True)
    t4 = time.perf_counter()
    print("size %d: %4d templates  Jess() %.1f ms  pack + tm_library_create %.1f ms  first query (1 molecule) %.1f ms  255 molecules %.1f ms" %
          (size, len(tpls), 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)
    for i, v in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
        tot[i] += v
print("total ms:", [round(1e3 * v, 1) for v in tot])
