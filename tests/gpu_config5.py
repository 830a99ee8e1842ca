"""BASELINE.json configs[4]: large complexes (>5000 residues), pLDDT-style conservation cutoff applied
explicitly, loosened -j (rmsd 2, distance 2.5).  Timing + truncation census; not a pytest test."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dist = float(sys.argv[2]) if len(sys.argv) > 2 else 2.5
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
t0 = time.time()
mols = [synthetic.structure(900000 + i, mean_residues=5500, sd_residues=300, min_residues=5000, max_residues=6500)
        .conserved(70) for i in range(n)]
print("generated", [len(m) for m in mols], "atoms in", round(time.time() - t0, 1), "s", flush=True)
packed = pyjess._packed(jess._templates, jess._device)
batch = pyjess.make_batch(mols, jess._device)
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | _native.TM_FLAG_COUNT_CANDIDATES
for d in (None, dist):
    c = cut if d is None else np.full(len(idx), d)
    t0 = time.time()
    res = _native.query(packed.native, batch, idx, 2.0, c, c, 10000, flags)
    hits, cands = res.hits(), res.candidates()
    print("distance", "defaults" if d is None else d, "wall %.2f s" % (time.time() - t0), "timing ms", res.timing(),
          "hits", len(hits), "candidates", int((cands & 0x7fffffff).sum()), "truncated pairs", int((cands >> 31).sum()),
          flush=True)
    res.close()
