"""PCIe-inclusive rate: host arrays -> tm_batch_create (H2D) -> tm_query -> hits on the host. Not a test."""
import sys
import time
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
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
for m in mols:
    m._encoded()  # vocabulary encoding is cached on the molecule: not part of the transfer
for rep in range(4):
    t0 = time.perf_counter()
    batch = pyjess.make_batch(mols, jess._device)
    t1 = time.perf_counter()
    res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags)
    hits = res.hits()
    t2 = time.perf_counter()
    res.close()
    batch.close()
    print(f"rep {rep}: batch upload {1e3 * (t1 - t0):.1f} ms, query + hits to host {1e3 * (t2 - t1):.1f} ms "
          f"-> {n / (t2 - t0):.0f} structures/s PCIe-inclusive ({len(hits)} hits)", flush=True)
