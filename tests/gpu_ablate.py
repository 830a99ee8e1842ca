"""Phase ablation of k_search with the timing-only build (make -C enzymm_amd/csrc ablate). Not a test."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("ENZYMM_AMD_LIB", str(ROOT / "enzymm_amd" / "libenzymm_amd_ablate.so"))
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
for label, extra in (("full: k_search blocks + k_deep", 0), ("k_deep sets templates up but does not search", 0x1000),
                     ("k_deep only claims and reads its runs", 0x2000), ("stages A-D1, nothing paged (no k_deep work)", 0x800),
                     ("stages A-C (no look at level 3)", 0x400), ("stages A-B (pair stage only)", 0x100), ("stage A only", 0x200)):
    times = []
    for _ in range(4):
        res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, base | extra)
        times.append(res.timing()[1])
        res.close()
    print("k_search ms (%s): %.3f" % (label, min(times)), flush=True)
