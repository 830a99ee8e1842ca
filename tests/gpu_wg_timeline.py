"""When and where the workgroups of the block search ran (timing build: make -C enzymm_amd/csrc ablate): how much of
the CU-time between the first start and the last end was held by a workgroup, per XCD and over time.  Not a test."""
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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
mols = synthetic.structures(n)
packed = pyjess._packed(jess._templates, jess._device)
batch = pyjess.make_batch(mols, jess._device)
lib = _native.load()
TAB = len(sys.argv) > 3 and sys.argv[3] == "tables"  # third argument "tables": k_tab_build's workgroups (the index is rebuilt by every query)
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | (int(sys.argv[2], 0) if len(sys.argv) > 2 else 0) | (_native.TM_FLAG_REBUILD_INDEX if TAB else 0)  # (ablation bits: 0x800 no depth-first search, 0x400 no triples)
N = 16384
buf = (C.c_ulonglong * (5 * N))()
WHICH = N | (0x80000000 if TAB else 0)
lib.tm_debug_wg_log(buf, WHICH)  # (resets the log: first_end starts from all ones)
for rep in range(4):
    res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags)
    t = res.timing()
    res.close()
    lib.tm_debug_wg_log(buf, WHICH)
log = np.array(list(buf), dtype=np.uint64).reshape(N, 5)
log = log[log[:, 1] != 0]
start, end = log[:, 0].astype(np.int64), log[:, 1].astype(np.int64)
hw, xcc = (log[:, 2] & np.uint64(0xffffffff)).astype(np.int64), (log[:, 2] >> np.uint64(32)).astype(np.int64) & 0xf
cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 0x7
where = ((xcc * 8 + se) * 2 + sh) * 16 + cu
t0, t1 = start.min(), end.max()
span = (t1 - t0) / 1e5
dur = (end - start) / 1e5
print("k_tab_build's workgroups (two per CU: 'CU-time' below counts slots of half a CU, divide by 2)" if TAB else "block search")
print("search %.3f ms by its events; %d workgroups over %d distinct CUs, first start to last end %.3f ms" %
      (t[1], len(log), len(np.unique(where)), span))
print("workgroup duration ms: mean %.3f, median %.3f, p90 %.3f, max %.3f; sum / (256 x span) = %.1f %% of the CU-time held" %
      (dur.mean(), np.median(dur), np.percentile(dur, 90), dur.max(), 100.0 * dur.sum() / (256.0 * span)))
for x in range(8):
    m = xcc == x
    if m.any():
        print("  XCD %d: %4d workgroups on %2d CUs, held %.3f ms per CU, last end at %.3f ms" %
              (x, m.sum(), len(np.unique(where[m])), dur[m].sum() / max(1, len(np.unique(where[m]))), (end[m].max() - t0) / 1e5))
# CUs holding a workgroup over time, in twenty slices
edges = np.linspace(t0, t1, 21)
busy = []
for a, b in zip(edges[:-1], edges[1:]):
    overlap = np.clip(np.minimum(end, b) - np.maximum(start, a), 0, None).sum()
    busy.append(overlap / (b - a))
print("CUs holding a workgroup, twenty slices of the span:", " ".join("%.0f" % v for v in busy))
gaps = []
for w in np.unique(where):
    m = where == w
    order = np.argsort(start[m])
    s_, e_ = start[m][order], end[m][order]
    gaps.extend(((s_[1:] - e_[:-1]) / 1e2).tolist())  # microseconds
gaps = np.array(gaps)
print("gap between a workgroup's end and the next start on the same CU, us: median %.1f, p90 %.1f, max %.1f (n = %d)" %
      (np.median(gaps), np.percentile(gaps, 90), gaps.max(), len(gaps)))
# the longest workgroups: where in the launch order they were, when they started
bid = np.nonzero(np.array(list(buf), dtype=np.uint64).reshape(N, 5)[:, 1] != 0)[0]
top = np.argsort(-dur)[:12]
print("longest workgroups (launch index, XCD, start ms, duration ms):",
      ", ".join("%d/%d/%.2f/%.2f" % (bid[i], xcc[i], (start[i] - t0) / 1e5, dur[i]) for i in top))
late = np.argsort(-end)[:12]
print("last to end (launch index, XCD, start ms, duration ms):",
      ", ".join("%d/%d/%.2f/%.2f" % (bid[i], xcc[i], (start[i] - t0) / 1e5, dur[i]) for i in late))
q = np.argsort(bid)
print("mean duration by launch-order decile:", " ".join("%.3f" % dur[q][k * len(q) // 10:(k + 1) * len(q) // 10].mean() for k in range(10)))
# inside a unit: when its first wave ran out of work, when its waves did on average, against its last wave's end
first = (log[:, 3].astype(np.int64) - start) / 1e5
mean_end = (log[:, 4].astype(np.float64) / 16.0 - start) / 1e5
print("waves of a unit end, as fractions of the unit's duration: first %.2f, mean %.2f (all units); slowest tenth of the units: first %.2f, mean %.2f" %
      ((first / dur).mean(), (mean_end / dur).mean(), (first / dur)[np.argsort(-dur)[:len(dur) // 10]].mean(),
       (mean_end / dur)[np.argsort(-dur)[:len(dur) // 10]].mean()))
print("last to end, (first wave's end, mean end, last end) ms after the unit's start:",
      ", ".join("%.2f/%.2f/%.2f" % (first[i], mean_end[i], dur[i]) for i in late))
