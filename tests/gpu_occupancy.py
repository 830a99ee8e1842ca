"""Search-kernel durations on a batch of SMALL synthetic structures (so that two workgroups' LDS fit a CU) under the
library build and LDS budget named in the environment (ENZYMM_AMD_LIB, ENZYMM_AMD_LDS_BUDGET).  Not a test."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
mean = float(sys.argv[2]) if len(sys.argv) > 2 else 150.0
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
mols = synthetic.structures(n, mean_residues=mean, sd_residues=mean / 10.0, min_residues=int(mean / 2), max_residues=int(mean * 1.5))
print("atoms per structure: mean %.0f max %d" % (sum(len(m) for m in mols) / n, max(len(m) for m in mols)))
packed = pyjess._packed(jess._templates, jess._device)
batch = pyjess.make_batch(mols, jess._device)
base = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
best = None
for i in range(8):
    res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, base | _native.TM_FLAG_REBUILD_INDEX)
    t = res.timing()
    a, b = res.search_split()
    nh = res.hit_count()
    res.close()
    print("#%d: index %.3f search %.3f (k_search %.3f + k_deep %.3f) finish %.3f ms, %d hits" % (i, t[0], t[1], a, b, t[2], nh), flush=True)
    if i >= 2 and (best is None or t[1] < best[0]): best = (t[1], a, b)
print("BEST search %.3f = %.3f + %.3f" % best)
