"""Where the waves of the block search spend their clock cycles, stage by stage (waiting included): the diagnostic build
`make -C enzymm_amd/csrc variant NAME=times EXTRA=-DTM_TIMES`.  Not a test."""
import ctypes as C
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
os.environ.setdefault("ENZYMM_AMD_LIB", str(ROOT / "enzymm_amd" / "libenzymm_amd_times.so"))
sys.path.insert(0, str(ROOT))
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
mols = synthetic.structures(n)
packed = pyjess._packed(jess._templates, jess._device)
batch = pyjess.make_batch(mols, jess._device)
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
lib = _native.load()
buf = (C.c_ulonglong * 16)()
for _ in range(2):
    res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags)
    ms = res.timing()[1]
    res.close()
lib.tm_debug_times(buf)
res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags)
ms = res.timing()[1]
res.close()
lib.tm_debug_times(buf)
t = [float(v) for v in buf]
names = ["A", "B", "C (with the D0/D1 flushes nested in it)", "D0", "D1 + queueing", "search below queued triples",
         "waiting at the end of a unit", "(unused)", "the waves' whole time in their units' template loops"]
searches = n * len(idx)
print("k_search + k_deep %.3f ms (instrumented build), %d searches" % (ms, searches))
unit = max(t[8], 1.0)
for i, name in enumerate(names):
    print("  %-52s %14.0f cycles  %5.1f %% of the waves' time in units  %8.1f cycles per search" % (name, t[i], 100.0 * t[i] / unit, t[i] / searches))
acc = sum(t[0:7])
print("  stages + waits accounted: %.1f %% of unit time net of staging" % (100.0 * acc / max(unit - t[7], 1.0)))
print("  blocks %d (%.1f templates each), C calls %d (%.2f per block), look flushes %d (%.2f per block), queue entries searched %d (%.3f per search)" %
      (t[9], searches / max(t[9], 1), t[10], t[10] / max(t[9], 1), t[11], t[11] / max(t[9], 1), t[12], t[12] / searches))
for i, name in ((1, "B"), (2, "C call"), (3, "D0 flush"), (4, "D1 flush"), (5, "queue entry")):
    cnt = {1: t[9], 2: t[10], 3: t[11], 4: t[11], 5: t[12]}[i]
    print("  cycles per %-12s %10.0f" % (name + ("" if i != 1 else " (whole block)"), t[i] / max(cnt, 1)))
