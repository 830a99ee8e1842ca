"""Search-tree counters of the diagnostic build (make -C enzymm_amd/csrc stats). Not a test."""
import ctypes as C
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("ENZYMM_AMD_LIB", str(ROOT / "enzymm_amd" / "libenzymm_amd_stats.so"))
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
mols = synthetic.structures(n)
lib = _native.load()
print("loaded", flush=True)
buf = (C.c_ulonglong * 64)()
print("stats0", lib.tm_debug_stats(buf), flush=True)
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
print("joint tiles/search", s[52] / searches, "joint survivors/search", s[53] / searches,
      "searches with survivor", s[54] / searches)
print("pair rounds with a survivor / search", s[54] / searches, "one-tile segments / search", s[58] / searches,
      "one-tile segments without survivor / search", s[59] / searches)
print("workgroups", int(s[55]), "with kind lists in LDS", int(s[56]), "mean kind-list pool entries", s[57] / max(s[55], 1))

# ablation: k_search time with (a) everything, (b) no subtrees below the joint scan, (c) staging only
packed = pyjess._packed(jess._templates, jess._device)
batch = pyjess.make_batch(mols, jess._device)
base = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
for label, extra in (("full", 0), ("no subtrees", 0x100), ("staging only", 0x200)):
    times = []
    for _ in range(3):
        res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, base | extra)
        times.append(res.timing()[1])
        res.close()
    print("k_search ms (%s): %.3f" % (label, min(times)))
