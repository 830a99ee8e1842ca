"""How long a structure beyond the 16-bit atom range takes: 20 x AF-P0DUB6 (81 580 atoms) alone against the 6780 default
templates, and inside a batch of ordinary structures.  Not a test.    python tests/gpu_wide_timing.py"""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from tests.test_gpu_round4 import _copies  # noqa: E402
from enzymm_amd import _native, pyjess, synthetic  # noqa: E402

idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
packed = pyjess._packed(jess._templates, jess._device)
af = pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb")
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
for copies in (17, 20, 24):
    big = _copies(af, copies, f"AFx{copies}")
    for label, mols in (("alone", [big]), ("with 200 ordinary structures", [big] + synthetic.structures(200))):
        batch = pyjess.make_batch(mols, jess._device)
        times = []
        for rep in range(4):
            t0 = time.perf_counter()
            res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags | (_native.TM_FLAG_REBUILD_INDEX if rep else 0))
            n = res.hit_count()
            times.append((time.perf_counter() - t0, res.timing()))
            res.close()
        batch.close()
        wall, (a, b, c) = min(times[1:], key=lambda t: t[0])
        print(f"{copies} copies ({len(big)} atoms) {label}: {n} hits, {1e3 * wall:.1f} ms wall (index+tables {a:.1f} ms, search {b:.1f} ms, finish {c:.1f} ms); first call {1e3 * times[0][0]:.0f} ms", flush=True)
