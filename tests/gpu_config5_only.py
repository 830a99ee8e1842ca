"""BASELINE.json configs[4] at -j 2 2.5 2.5 only (ten complexes, four queries): the profiling target. Not a test."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
packed = pyjess._packed(jess._templates, jess._device)
base = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | _native.TM_FLAG_REBUILD_INDEX
mols = [synthetic.complex_structure(i, 6000).conserved(70) for i in range(n)]
batch = pyjess.make_batch(mols, jess._device)
c = cut if len(sys.argv) > 2 and sys.argv[2] == "default" else np.full(len(idx), 2.5)
for rep in range(4):
    res = _native.query(packed.native, batch, idx, 2.0, c, c, 10000, base)
    print(rep, res.hit_count(), [round(x, 2) for x in res.timing()], [round(x, 2) for x in res.search_split()], flush=True)
    res.close()
