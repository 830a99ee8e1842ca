"""A longer randomised comparison of the HIP path with the oracle than the suite runs (tests/test_gpu_parity.py::
test_randomised_parameters_vs_oracle draws six trials): template subsets from the library and implanted (dense) ones,
per-template thresholds with the dynamic distance above and below the cutoff, ignore_chain, best_match on and off, small
max_candidates (truncation sets), and batches mixing real, synthetic, multi-chain, off-grid (fp64 hand-over), scattered
(host-sorted) and empty structures -- now and then a structure beyond the 16-bit atom range.  Not a test; writes
gpurun_out/fuzz.json.

    python tests/gpu_fuzz.py [trials] [seed]"""
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from tests.test_gpu_parity import _multichain  # noqa: E402
from tests.test_gpu_round4 import _copies  # noqa: E402
from tests.test_gpu_round5 import _scattered  # noqa: E402
from enzymm_amd import pyjess, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 7
rng = np.random.default_rng(seed)
ptpl, otpl = helpers.product_library_templates(), helpers.oracle_library_templates()
amy = pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb")
af = pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb")
synth = synthetic.structures(24, start=5000)
off_grid = [helpers.subset_molecule(m, np.arange(len(m)), xyz=m._xyz + 3.3e-6 * (i + 1), id=f"off{i}") for i, m in enumerate(synth[:4])]
pool = [amy, af, _multichain(amy, 2, shift=38.0), _multichain(af, 3, shift=45.0), pyjess.Molecule(), _scattered(synth[5]), _scattered(off_grid[0])] + synth + off_grid
wide = _copies(af, 17, "AFx17")
wide._xyz[:] = np.round(wide._xyz + rng.normal(0.0, 0.01, size=wide._xyz.shape), 3)
# a LONG narrow structure (three copies of the AlphaFold model, 12 237 atoms: beyond every LDS layout, 16-bit indices): its
# batch is searched out of global memory -- segmented index build, gathered and culled table build, block search and k_deep
# on records in L2
long_narrow = _copies(af, 3, "AFx3")
long_narrow._xyz[:] = np.round(long_narrow._xyz + rng.normal(0.0, 0.01, size=long_narrow._xyz.shape), 3)
stats = dict(trials=0, searches=0, hits=0, candidates=0, truncated_pairs=0, all_hits_rows=0, wide_trials=0, long_trials=0, mismatches=0, seconds=0.0)
t_start = time.time()
for trial in range(trials):
    dense = trial % 3 == 1
    if dense:
        host = [amy, af, synth[trial % len(synth)]][trial % 3]
        product = helpers.implant_templates(host, ptpl, rng, max_templates=60)
        oracle_tpls = [helpers.to_oracle_template(t) for t in product]
        mols = [host] + [pool[i] for i in rng.choice(len(pool), size=2, replace=False)]
    else:
        idx = np.sort(rng.choice(len(ptpl), size=int(rng.choice([40, 150, 400])), replace=False))
        product, oracle_tpls = [ptpl[i] for i in idx], [otpl[i] for i in idx]
        mols = [pool[i] for i in rng.choice(len(pool), size=int(rng.choice([1, 3, 6])), replace=False)]
    use_wide = trial % 20 == 7
    if use_wide:
        mols = mols[:2] + [wide]
        stats["wide_trials"] += 1
    if trial % 20 == 13:
        mols = mols[:2] + [long_narrow]
        stats["long_trials"] += 1
    n = len(product)
    jess, olib = pyjess.Jess(product), oracle.OracleLibrary(oracle_tpls)
    rmsd = rng.uniform(0.8, 3.0, size=n)
    cutoff = rng.uniform(0.5, 2.2 if not use_wide else 1.4, size=n)
    maxdyn = cutoff + rng.uniform(-0.4, 0.6, size=n)
    ignore = bool(rng.integers(2))
    best = bool(rng.integers(4))  # all-hits mode one trial in four
    cap = int(rng.choice([3, 40, 10000, 10000]))
    out = jess.query_batch(mols, rmsd, cutoff, maxdyn, max_candidates=cap, ignore_chain=ignore, best_match=best,
                           count_candidates=True, return_raw=True)
    table, cands = out
    for s, mol in enumerate(mols):
        r = oracle.query_library(helpers.to_oracle_molecule(mol), olib, rmsd, cutoff, maxdyn, max_candidates=cap, ignore_chain=ignore)
        cut_short = r["truncated"].astype(bool)
        bad = 0
        bad += int(not np.array_equal((cands[s] >> 31).astype(bool), cut_short))
        bad += int(not np.array_equal((cands[s] & 0x7FFFFFFF)[~cut_short], r["candidates"][~cut_short]))
        rows = table[table["structure"] == s]
        if best:
            mine = {int(h["template_sel"]): h for h in rows}
            theirs = set(np.nonzero(r["hit"] & ~cut_short)[0].tolist())
            bad += int({t for t in mine if not cut_short[t]} != theirs)
            for t in theirs & set(mine):
                k = olib.templates[t].k
                bad += int(not np.array_equal(mine[t]["atoms"][:k], r["atoms"][t][:k]))
                bad += int(abs(mine[t]["rmsd"] - r["rmsd"][t]) > 1e-9)
            stats["hits"] += len(theirs)
        else:
            # every candidate within the RMSD threshold is a row: per template as many rows as ... at most its candidates, the
            # best of them the oracle's best hit
            for t in np.nonzero(~cut_short)[0]:
                mine = rows[rows["template_sel"] == t]
                bad += int(len(mine) > int(r["candidates"][t]))
                if r["hit"][t]:
                    bad += int(len(mine) == 0 or abs(float(mine["rmsd"].min()) - r["rmsd"][t]) > 1e-9)
                else:
                    bad += int(len(mine) != 0)
            stats["all_hits_rows"] += len(rows)
        stats["mismatches"] += bad
        stats["searches"] += n
        stats["candidates"] += int(r["candidates"].sum())
        stats["truncated_pairs"] += int(cut_short.sum())
        if bad:
            print(f"trial {trial} structure {s} ({mol.id}, {len(mol)} atoms): {bad} mismatches; dense {dense} ignore_chain {ignore} best {best} cap {cap}", flush=True)
    stats["trials"] += 1
stats["seconds"] = round(time.time() - t_start, 1)
head = ROOT / ".build_head"
stats["git_head"] = head.read_text().strip() if head.exists() else None
stats["seed"] = seed
(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "fuzz.json").write_text(json.dumps(stats, indent=1) + "\n")
print(json.dumps(stats))
