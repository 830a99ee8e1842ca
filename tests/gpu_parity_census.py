"""One-off parity census: N structures of the bench batch x the 6780 default templates, the HIP path against the CPU oracle
(hit sets, matched atoms, RMSD within 1e-5, candidate counts, truncation flags).  Writes gpurun_out/parity_census.json.
Not a test (tests/test_gpu_round4.py::test_synthetic_batch_of_128_vs_oracle is the in-suite sample).

    python tests/gpu_parity_census.py [structures] [first structure]"""
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic  # noqa: E402
from oracle import oracle  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
start = int(sys.argv[2]) if len(sys.argv) > 2 else 0
workload = sys.argv[3] if len(sys.argv) > 3 else "synthetic"
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
otpl = helpers.oracle_library_templates()
olib = oracle.OracleLibrary([otpl[i] for i in idx])
if workload == "real":  # the first n structures of bench.py's real-structure batch (rigid copies of 1AMY and AF-P0DUB6, shuffled)
    real = [pyjess.Molecule.load(helpers.GOLDEN / f) for f in ("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb")]
    copies = synthetic.rigid_copies(real[0], 500, seed=7) + synthetic.rigid_copies(real[1], 500, seed=8)
    order = np.random.default_rng(5).permutation(len(copies))
    mols = [copies[i] for i in order][start:start + n]
elif workload == "config5":  # BASELINE.json configs[4]: multi-chain complexes of ~6300 residues, conserved(70), default cutoffs
    mols = [synthetic.complex_structure(start + i, 6000).conserved(70) for i in range(n)]
else:
    mols = synthetic.structures(n, start=start)
t0 = time.time()
table, cands = jess.query_batch(mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True, template_index=idx,
                                count_candidates=True, return_raw=True)
t_gpu = time.time() - t0
cores = max(1, min(16, len(os.sched_getaffinity(0))))
t0 = time.time()
with ThreadPoolExecutor(cores) as pool:
    results = list(pool.map(lambda m: oracle.query_library(helpers.to_oracle_molecule(m), olib, 2.0, cut, cut,
                                                           max_candidates=10000, ignore_chain=True, threads=1), mols))
t_cpu = time.time() - t0
by_s = {}
for h in table:
    by_s.setdefault(int(h["structure"]), {})[int(h["template_sel"])] = h
out = dict(workload=workload, structures=n, first_structure=start, atoms_per_structure=float(np.mean([len(m) for m in mols])), templates=int(len(idx)), hits_gpu=int(len(table)), hits_oracle=0, hit_set_mismatches=0,
           atom_mismatches=0, rmsd_beyond_1e5=0, max_rmsd_difference=0.0, candidate_count_mismatches=0, truncation_mismatches=0,
           truncated_pairs=0, candidates_total=0, gpu_seconds_first_call=round(t_gpu, 3), oracle_seconds=round(t_cpu, 1), oracle_threads=cores)
for s, r in enumerate(results):
    mine = by_s.get(s, {})
    ok = ~r["truncated"].astype(bool)
    theirs = set(np.nonzero(r["hit"] & ok)[0].tolist())
    out["hits_oracle"] += int(r["hit"].sum())
    out["hit_set_mismatches"] += len(theirs ^ {t for t in mine if ok[t]})
    out["candidate_count_mismatches"] += int(np.count_nonzero((cands[s] & 0x7FFFFFFF) != r["candidates"]))
    out["truncation_mismatches"] += int(np.count_nonzero((cands[s] >> 31).astype(bool) != r["truncated"].astype(bool)))
    out["truncated_pairs"] += int(r["truncated"].sum())
    out["candidates_total"] += int(r["candidates"].sum())
    for t in theirs & set(mine):
        k = olib.templates[t].k
        if not np.array_equal(mine[t]["atoms"][:k], r["atoms"][t][:k]):
            out["atom_mismatches"] += 1
        d = abs(float(mine[t]["rmsd"]) - float(r["rmsd"][t]))
        out["max_rmsd_difference"] = max(out["max_rmsd_difference"], d)
        if d > 1e-5:
            out["rmsd_beyond_1e5"] += 1
head = ROOT / ".build_head"
out["git_head"] = head.read_text().strip() if head.exists() else None
Path(ROOT / "gpurun_out").mkdir(exist_ok=True)
(ROOT / "gpurun_out" / "parity_census.json").write_text(json.dumps(out, indent=1) + "\n")
print(json.dumps(out))
