"""One structure alone against the full library (BASELINE.json configs[1]): block search / depth-first split of the
search time over many repeats (min, median). Not a test."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, _native  # noqa: E402

idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
packed = pyjess._packed(jess._templates, jess._device)
base = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | _native.TM_FLAG_REBUILD_INDEX
for name in ("AF-P0DUB6-F1-model_v4.pdb", "1AMY.pdb"):
    mol = pyjess.Molecule.load(helpers.GOLDEN / name)
    batch = pyjess.make_batch([mol], jess._device)
    rows = []
    for _ in range(40):
        res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, base)
        res.hit_count()
        rows.append(tuple(res.timing()) + tuple(res.search_split()))
        res.close()
    batch.close()
    a = np.array(rows[5:])
    print(f"{name}: index {np.median(a[:, 0]):.3f}, search {np.median(a[:, 1]):.3f} (min {a[:, 1].min():.3f}), "
          f"block {np.median(a[:, 3]):.3f}, deep {np.median(a[:, 4]):.3f} (min {a[:, 4].min():.3f}) ms", flush=True)
