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
if len(sys.argv) > 2 and sys.argv[2].startswith("complexes"):  # BASELINE.json configs[4]: large complexes, loosened thresholds ("complexes_default": the default ones)
    mols = [synthetic.complex_structure(i, 6000).conserved(70) for i in range(n)]
    if sys.argv[2] == "complexes":
        cut = np.full(len(idx), 2.5)
elif len(sys.argv) > 2 and sys.argv[2] == "real":  # rigid copies of the two structures the reference's tests ship (bench.py's real_* figures)
    real = [pyjess.Molecule.load(ROOT / "tests" / "golden" / f) for f in ("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb")]
    mols = synthetic.rigid_copies(real[0], (n + 1) // 2, seed=7) + synthetic.rigid_copies(real[1], n // 2, seed=8)
else:
    mols = synthetic.structures(n)
lib = _native.load()
print("loaded", flush=True)
buf = (C.c_ulonglong * 320)()
print("stats0", lib.tm_debug_stats(buf), flush=True)
t0 = time.time()
table = jess.query_batch(mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True, template_index=idx,
                         return_raw=True)
print("query s", time.time() - t0, "hits", len(table), "atoms per structure", sum(len(m) for m in mols) / len(mols))
lib.tm_debug_stats(buf)
s = np.array(list(buf), dtype=np.float64)
searches = s[49]
print("raw evaluations per level", [int(v) for v in s[0:24]])
print("raw survivors per level  ", [int(v) for v in s[24:48]])
print("searches", int(searches), "completes", int(s[48]), "fp64 rejects", int(s[51]))
print("per search: pair passes %.3f  pair survivors %.3f  triple passes %.3f  triple survivors %.3f" %
      (s[52] / searches, s[53] / searches, s[67] / searches, s[68] / searches))
print("per search: look flushes %.4f  triples looked at %.3f  ended by the residue check (D0) %.3f  asked for a fourth lead (D1) %.3f (chunks each %.2f)  paged out %.3f" %
      (s[64] / searches, s[66] / searches, s[69] / searches, s[60] / searches, s[61] / max(s[60], 1), s[65] / searches))
if s[303]:  # (counters of profiles/r06_logs/experiments_after_v41.patch: not in the tree's stats build)
    print("stage C: %.2f items per search, %.2f of them inside the window of the stored distance (%.1f %%)" %
          (s[303] / searches, s[302] / searches, 100.0 * s[302] / max(s[303], 1)))
print("residue-check items per triple looked at %.2f, passes of 64 items per search %.4f" % (s[62] / max(s[66], 1), s[63] / searches))
paged = max(s[65], 1)
print("k_deep chunk evaluations per level (per paged triple)", np.round(s[3:24] / paged, 2))
print("k_deep survivors per level (per paged triple)        ", np.round(s[27:48] / paged, 2))
print("pair tests (j-loop trips) per paged triple", s[50] / paged)
print("chunk evaluations per paged triple: lead levels %.2f, other levels %.2f" % (s[82] / paged, s[83] / paged))
print("paged triples by residue count of their template (3..8):", np.round(s[87:93] / paged, 3))
for mode, name in ((0, "k_search blocks"), (2, "k_deep")):
    busy, held, staging = s[70 + 4 * mode], s[71 + 4 * mode], s[72 + 4 * mode]
    print("%s: waves working %.1f %% of the time their workgroup holds its CU after staging; staging %.1f %% of that time" %
          (name, 100.0 * busy / max(held, 1), 100.0 * staging / max(held, 1)))
print("queue entries of the block search by how long the search below them took (10 ns ticks, bit length): count, share of the time")
tot = max(s[216:240].sum(), 1)
for b in range(24):
    if s[192 + b]:
        print("  2^%-2d  %9.0f  %5.1f %%" % (b, s[192 + b], 100.0 * s[216 + b] / tot))
print("  entries by depth (3 = triples of the table stages, more = handed over by a wave in a large subtree):", [int(v) for v in s[240:256]])
print("  handed-over entries whose atoms do not add up to their writer's sum: %d of %d" % (int(s[300]), int(s[301])))
print("  handed over by depth:", [int(v) for v in s[256:280]])
print("  taken by depth:      ", [int(v) for v in s[240:256]])
print("k_tab_build (compact layout): waves working %.1f %% of the time their workgroup holds its CU after staging; staging %.1f %% of that time" %
      (100.0 * s[93] / max(s[94], 1), 100.0 * s[95] / max(s[94], 1)))
print("table build, by bit length of the cross product: tests of swept leaders / non-leaders, entries of both (per structure)")
for b in range(24):
    if s[96 + b] or s[120 + b]:
        print("  2^%-2d  %10.0f %10.0f   %9.0f %9.0f" % (b, s[96 + b] / len(mols), s[120 + b] / len(mols), s[144 + b] / len(mols), s[168 + b] / len(mols)))
print("tables swept by a wave (leaders and tables of small groups), by bit length of the cross product (0..15+), per structure:",
      [round(v / len(mols), 1) for v in s[284:300]])
print("big tables (all waves of a workgroup build them together): %.1f per structure, %.0f tests and %.0f entries per structure, %d with list B beyond the staged capacity" %
      (s[280] / len(mols), s[281] / len(mols), s[282] / len(mols), int(s[283])))
print("workgroups", int(s[55]), "with kind lists in LDS", int(s[56]), "mean kind-list pool entries", s[57] / max(s[55], 1))
# ablation: k_search time with (a) everything, (b) no subtrees below the joint scan, (c) staging only
packed = pyjess._packed(jess._templates, jess._device)
batch = pyjess.make_batch(mols, jess._device)
base = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
for label, extra in (("full", 0), ("no k_deep work", 0x800), ("stage A only", 0x200)):
    times = []
    for _ in range(3):
        res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, base | extra)
        times.append(res.timing()[1])
        res.close()
    print("k_search ms (%s): %.3f" % (label, min(times)))
    if label == "full":
        # the same launch again with the counters: how long the 256 CUs were held in sum against how long the kernel took
        lib.tm_debug_stats(buf)
        res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, base)
        t = res.timing()[1]
        res.close()
        lib.tm_debug_stats(buf)
        s2 = np.array(list(buf), dtype=np.float64)
        held = (s2[71] + s2[72]) / 16.0 / 256.0 / 1e5  # ticks of 10 ns, 16 waves per workgroup, 256 CUs
        print("k_search blocks: CUs held %.3f ms each on average, search %.3f ms (both launches): %.1f %% of the CU-time in use"
              % (held, t, 100.0 * held / t))
