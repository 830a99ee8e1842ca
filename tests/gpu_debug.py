"""Ad-hoc GPU check used while bringing the kernels up (not collected by pytest)."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, _native  # noqa: E402
from oracle import parse, oracle  # noqa: E402

G = json.loads((helpers.GOLDEN / "golden.json").read_text())
print("devices", _native.device_count(), flush=True)

mol = pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb")
for key in ("G1", "G2"):
    g = G[key]
    tpl = pyjess.Template.load(helpers.GOLDEN / "templates" / g["template"])
    jess = pyjess.Jess([tpl])
    hits = list(jess.query(mol, g["rmsd_threshold"], g["distance_cutoff"], g["max_dynamic_distance"],
                           max_candidates=10000, best_match=True))
    print(key, "hits", len(hits))
    for h in hits:
        print("  rmsd", h.rmsd, "expected", g["rmsd"], "ncand", h.n_candidates)
        print("  atoms", [(a.serial, a.name) for a in h.atoms()])

# parity against the oracle, full default selection
idx, cut = helpers.default_selection()
ptpl = helpers.product_library_templates()
otpl = helpers.oracle_library_templates()
jess = pyjess.Jess(ptpl)
olib = oracle.OracleLibrary([otpl[i] for i in idx])
mols = [pyjess.Molecule.load(helpers.GOLDEN / n) for n in ("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb")]
omols = [parse.parse_molecule((helpers.GOLDEN / n).read_text()) for n in ("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb")]
t0 = time.time()
table, cands = jess.query_batch(mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True, template_index=idx,
                                count_candidates=True, return_raw=True)
print("gpu query", time.time() - t0, "s; hits", len(table), flush=True)
bad = 0
for s, om in enumerate(omols):
    t0 = time.time()
    r = oracle.query_library(om, olib, 2.0, cut, cut, max_candidates=10000, ignore_chain=True)
    print("oracle", s, time.time() - t0, "s; hits", int(r["hit"].sum()))
    gc = cands[s] & 0x7FFFFFFF
    mism = np.nonzero(gc != r["candidates"])[0]
    print("  candidate-count mismatches:", len(mism), mism[:10], gc[mism[:10]], r["candidates"][mism[:10]])
    bad += len(mism)
    mine = {int(h["template_sel"]): h for h in table if h["structure"] == s}
    theirs = set(np.nonzero(r["hit"])[0].tolist())
    print("  hit templates equal:", set(mine) == theirs, len(mine), len(theirs))
    for t in sorted(theirs & set(mine)):
        k = olib.templates[t].k
        a_ok = np.array_equal(mine[t]["atoms"][:k], r["atoms"][t][:k])
        d = abs(mine[t]["rmsd"] - r["rmsd"][t])
        dr = np.abs(mine[t]["rot"].reshape(3, 3) - r["rot"][t]).max()
        if not a_ok or d > 1e-9 or dr > 1e-7:
            bad += 1
            print("   MISMATCH t", t, a_ok, d, dr)
    bad += len(theirs ^ set(mine))
print("TOTAL BAD", bad)
