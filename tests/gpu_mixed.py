"""Timing of a mixed batch (many ordinary structures + a few large ones). Not a test."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
mols = synthetic.structures(n)
af = pyjess.Molecule.load(str(ROOT / "tests/golden/AF-P0DUB6-F1-model_v4.pdb"))


def multichain(mol, copies, shift):
    cols = {k: np.concatenate([v] * copies) for k, v in mol._cols.items()}
    xyz = np.concatenate([mol._xyz + np.array([shift * c, 0.0, 0.0]) for c in range(copies)])
    cols["chain_id"] = np.concatenate([np.full(len(mol), (" " + "ABCDEFGH"[c]).encode(), dtype="S2") for c in range(copies)])
    out = pyjess.Molecule.__new__(pyjess.Molecule)
    out._init_columns(cols, xyz, f"{mol.id}x{copies}")
    return out


packed = pyjess._packed(jess._templates, jess._device)
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
for label, batch_mols in (("%d synthetic" % n, mols), ("%d synthetic + AF (4079 atoms)" % n, mols + [af]),
                          ("%d synthetic + 2 x AF trimer (12237 atoms)" % n, mols + [multichain(af, 3, 70.0)] * 2),
                          ("8 x AF trimer only", [multichain(af, 3, 70.0)] * 8)):
    batch = pyjess.make_batch(batch_mols, jess._device)
    times = []
    for _ in range(3):
        res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags)
        times.append(res.timing()[1])
        hits = res.hit_count()
        res.close()
    print("%-45s k_search %.2f ms, %d hits" % (label, min(times), hits), flush=True)
