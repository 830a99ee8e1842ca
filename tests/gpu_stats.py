"""Search-tree counters of the diagnostic build (make -C enzymm_amd/csrc stats). Not a test."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ["ENZYMM_AMD_LIB"] = str(ROOT / "enzymm_amd" / "libenzymm_amd_stats.so")
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
mols = synthetic.structures(n)
lib = _native.load()
buf = (C.c_ulonglong * 64)()
lib.tm_debug_stats(buf)
t0 = time.time()
table = jess.query_batch(mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True, template_index=idx,
                         return_raw=True)
print("query s", time.time() - t0, "hits", len(table))
lib.tm_debug_stats(buf)
s = np.array(list(buf), dtype=np.float64)
searches = s[49]
print("searches", int(searches), "chunk evals/search", s[:24].sum() / searches, "pair tests/search", s[50] / searches,
      "survivors/search", s[24:48].sum() / searches, "completes", int(s[48]), "fp64 rejects", int(s[51]))
print("evals per level  ", np.round(s[:24] / searches, 2))
print("surv per level   ", np.round(s[24:48] / searches, 2))
