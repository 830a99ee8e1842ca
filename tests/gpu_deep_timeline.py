"""When the workgroups of k_deep ran and ended for large complexes at the loose cutoffs (timing build: make -C enzymm_amd/csrc
ablate): the span of the launch, how long each workgroup and its first / average wave were at work.  Not a test."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("ENZYMM_AMD_LIB", str(ROOT / "enzymm_amd" / "libenzymm_amd_ablate.so"))
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
mols = [synthetic.complex_structure(i, 6000).conserved(70) for i in range(n)]
packed = pyjess._packed(jess._templates, jess._device)
batch = pyjess.make_batch(mols, jess._device)
lib = _native.load()
c = np.full(len(idx), 2.5)
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
N = 16384
buf = (C.c_ulonglong * (5 * N))()
for which, name in ((N | 0x40000000, "k_deep"), (N, "k_search blocks"), (N | 0x80000000, "k_tab_build")):
    lib.tm_debug_wg_log(buf, which)
    for rep in range(3):
        res = _native.query(packed.native, batch, idx, 2.0, c, c, 10000, flags | (_native.TM_FLAG_REBUILD_INDEX if name == "k_tab_build" else 0))
        t = res.timing(); split = res.search_split()
        res.close()
        lib.tm_debug_wg_log(buf, which)
    log = np.array(list(buf), dtype=np.uint64).reshape(N, 5)
    log = log[log[:, 1] != 0]
    if not len(log):
        print(name + ": nothing logged")
        continue
    start, end, first_end = log[:, 0].astype(np.int64), log[:, 1].astype(np.int64), log[:, 3].astype(np.int64)
    waves = 16
    mean_end = log[:, 4].astype(np.float64) / waves
    t0, t1 = start.min(), end.max()
    tick = 1e-5  # 10 ns ticks -> ms
    print("%s: %d workgroups logged, launch span %.3f ms (timing of the query: %s, split %s)" %
          (name, len(log), (t1 - t0) * tick, [round(x, 2) for x in t], [round(x, 2) for x in split]))
    dur = (end - start) * tick
    print("  workgroup duration ms: mean %.3f  median %.3f  p90 %.3f  max %.3f; starts within %.3f ms" %
          (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(), (start.max() - t0) * tick))
    print("  first wave done after %.2f of its workgroup's duration on average, the average wave after %.2f" %
          (np.mean((first_end - start) / np.maximum(end - start, 1)), np.mean((mean_end - start) / np.maximum(end - start, 1))))
    ends = np.sort((end - t0) * tick)
    print("  workgroups still running at 50/75/90/95/99 %% of the span: %s" %
          [int((ends > f * (t1 - t0) * tick).sum()) for f in (0.5, 0.75, 0.9, 0.95, 0.99)])
    print("  CU-time held by workgroups: %.1f %% of 256 CUs x span" % (100.0 * dur.sum() / (256 * (t1 - t0) * tick)))
