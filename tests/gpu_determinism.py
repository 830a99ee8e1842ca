"""The same query over and over on fresh batches: every hit table must equal the first, byte for byte (candidates of a
pair are found by many waves in no fixed order; nothing that is reported may depend on it).  Not a pytest test.

    python tests/gpu_determinism.py [repeats] [structures]"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, template, _native  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
lib = template.load_packed_library()
jess = pyjess.Jess(lib)
packed = pyjess._packed(jess._templates, jess._device)
amy = pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb")
af = pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb")
mols = [amy] + synthetic.structures(n - 10) + [af, pyjess.Molecule()] + synthetic.structures(8, start=500)
sel = np.arange(len(lib), dtype=np.uint32)
eff = np.array([t.effective_size for t in lib])
sel = sel[eff >= 3]
cut = np.array([helpers.DEFAULT_DISTANCE[min(int(e), 8)] for e in eff[sel]])
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | _native.TM_FLAG_COUNT_CANDIDATES
first = None
bad = 0
# between the repeats other queries run (other structures, other thresholds, other table and result sizes): whatever a
# kernel reads without having written it first would then hold leftovers of THOSE queries in the recycled buffers
other = [synthetic.structures(24, start=900 + 30 * i) + ([af] if i % 2 else [amy, amy]) for i in range(4)]
loose = np.full(len(sel), 2.2)
for r in range(reps):
    if r:
        o = other[r % len(other)]
        ob = pyjess.make_batch(o, jess._device)
        ores = _native.query(packed.native, ob, sel[:: 1 + r % 3], 2.0, loose[:: 1 + r % 3] if r % 2 else cut[:: 1 + r % 3],
                             loose[:: 1 + r % 3] if r % 2 else cut[:: 1 + r % 3], 10000 if r % 5 else 50, flags)
        ores.hit_count()
        ores.close(); ob.close()
    batch = pyjess.make_batch(mols, jess._device)
    res = _native.query(packed.native, batch, sel, 2.0, cut, cut, 10000, flags)
    table, cands = res.hits(), res.candidates()
    res.close(); batch.close()
    if first is None:
        first = (table, cands)
        print("reference run:", len(table), "hits,", int((cands & 0x7fffffff).sum()), "candidates", flush=True)
        continue
    if table.tobytes() != first[0].tobytes() or not np.array_equal(cands, first[1]):
        bad += 1
        a, b = first[0], table
        print(f"run {r}: DIFFERENT: {len(a)} vs {len(b)} hits; candidate tables equal: {np.array_equal(cands, first[1])}")
        ka = {(int(h['structure']), int(h['template_sel'])): h for h in a}
        kb = {(int(h['structure']), int(h['template_sel'])): h for h in b}
        for k in sorted(set(ka) ^ set(kb))[:5]:
            print("   only in", "first" if k in ka else "this", k, "candidates", int(first[1][k]) , int(cands[k]))
        for k in sorted(set(ka) & set(kb)):
            if ka[k].tobytes() != kb[k].tobytes():
                print("   differs", k, "rmsd", ka[k]['rmsd'], kb[k]['rmsd'], "atoms equal", np.array_equal(ka[k]['atoms'], kb[k]['atoms']),
                      "n_cand", ka[k]['n_candidates'], kb[k]['n_candidates'], "flags", ka[k]['flags'], kb[k]['flags'])
                break
        d = np.nonzero(cands != first[1])
        for s_, t_ in list(zip(*d))[:5]:
            print("   candidate count differs at", (int(s_), int(t_)), int(first[1][s_, t_]), int(cands[s_, t_]))
print("runs that differed from the first:", bad, "of", reps - 1)
