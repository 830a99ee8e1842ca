"""k_search duration (HIP events of the query) for the bench batch under a few query settings. Not a test."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
mols = synthetic.structures(n)
packed = pyjess._packed(jess._templates, jess._device)
batch = pyjess.make_batch(mols, jess._device)
base = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
for label, extra in (("cached index", 0), ("rebuild index", _native.TM_FLAG_REBUILD_INDEX), ("cached again", 0)):
    for i in range(4):
        res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, base | extra)
        t = res.timing()
        nh = res.hit_count()
        res.close()
        print("%s #%d: index %.3f search %.3f finish %.3f ms, %d hits" % (label, i, t[0], t[1], t[2], nh), flush=True)
