"""configs[4] at -j 2 2.5 2.5 with the timing-only build: what k_deep spends on claims, template set-up and the search
proper (query flag bits 0x2000 / 0x1000), for one complex and for ten.  Not a test."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("ENZYMM_AMD_LIB", str(ROOT / "enzymm_amd" / "libenzymm_amd_ablate.so"))
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
packed = pyjess._packed(jess._templates, jess._device)
base = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
c = np.full(len(idx), 2.5)
for n in (1, 10):
    mols = [synthetic.complex_structure(i, 6000).conserved(70) for i in range(n)]
    batch = pyjess.make_batch(mols, jess._device)
    for label, extra in (("full", 0), ("k_deep: set-up, no search", 0x1000), ("k_deep: claims only", 0x2000), ("nothing paged", 0x800)):
        best = None
        for _ in range(3):
            res = _native.query(packed.native, batch, idx, 2.0, c, c, 10000, base | extra)
            t, sp = res.timing(), res.search_split()
            res.close()
            if best is None or t[1] < best[0]:
                best = (t[1], sp[0], sp[1], t[2])
        print("%2d complex(es), %-28s search %.2f ms = block %.2f + deep %.2f; finish %.2f" % ((n, label) + best), flush=True)
    batch.close()
