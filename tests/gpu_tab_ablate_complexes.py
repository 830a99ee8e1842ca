"""Phase timing of the index and table build for configs[4]'s ten 25 k-atom complexes with the timing-only build
(ENZYMM_AMD_TAB_DEBUG=0|1|2: all / no fill sweep / no sweeps). Not a test."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("ENZYMM_AMD_LIB", str(ROOT / "enzymm_amd" / "libenzymm_amd_ablate.so"))
sys.path.insert(0, str(ROOT))
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
mols = [synthetic.complex_structure(i, 6000).conserved(70) for i in range(n)]
packed = pyjess._packed(jess._templates, jess._device)
batch = pyjess.make_batch(mols, jess._device)
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | _native.TM_FLAG_REBUILD_INDEX | 0x200
times = []
for _ in range(4):
    res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags)
    times.append(res.timing()[0])
    res.close()
print("%d complexes: index + tables ms (ENZYMM_AMD_TAB_DEBUG=%s): %.3f" % (n, os.environ.get("ENZYMM_AMD_TAB_DEBUG", "0"), min(times)), flush=True)
