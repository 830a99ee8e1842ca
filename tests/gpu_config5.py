"""BASELINE.json configs[1] and configs[4] as timings (parity of both is in tests/test_gpu_round2.py /
tests/test_gpu_parity.py).  Not a pytest test.

configs[1]: one AlphaFold monomer (AF-P0DUB6, 511 residues) and 1AMY, each alone, vs the full shipped library:
            latency of the search (HIP events) and of Matcher.run (wall).
configs[4]: multi-chain complexes of > 5000 residues, pLDDT filter conserved(70) applied explicitly, loosened
            `-j 2 2.5 2.5`: structures/s, candidates, truncated (template, structure) pairs."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import jess_run, pyjess, synthetic, template, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
packed = pyjess._packed(jess._templates, jess._device)
base = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | _native.TM_FLAG_REBUILD_INDEX

# ---- configs[1]
matcher = jess_run.Matcher(template.load_packed_library())
for name in ("AF-P0DUB6-F1-model_v4.pdb", "1AMY.pdb"):
    mol = pyjess.Molecule.load(helpers.GOLDEN / name)
    batch = pyjess.make_batch([mol], jess._device)
    times = []
    for _ in range(6):
        res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, base)
        n_hits = res.hit_count()
        times.append(res.timing())
        res.close()
    batch.close()
    walls = []
    for _ in range(4):
        t0 = time.perf_counter()
        out = matcher.run([mol])
        walls.append(time.perf_counter() - t0)
    t = np.array(times[1:]).mean(0)
    print(f"configs[1] {name}: {len(mol)} atoms, {n_hits} hits; index+tables {t[0]:.2f} ms, search {t[1]:.2f} ms, "
          f"finish {t[2]:.2f} ms on the GPU; Matcher.run (upload, search, {sum(len(v) for v in out.values())} filtered "
          f"matches) {1e3 * min(walls):.1f} ms wall", flush=True)

# ---- configs[4]
t0 = time.time()
mols = [synthetic.complex_structure(i, 6000).conserved(70) for i in range(n)]
residues = [len(set(zip(m._cols["chain_id"].tolist(), m._cols["residue_number"].tolist()))) for m in mols]
print(f"configs[4] {n} complexes generated in {time.time() - t0:.1f} s: {int(np.mean([len(m) for m in mols]))} atoms / "
      f"{int(np.mean(residues))} residues each after conserved(70)", flush=True)
batch = pyjess.make_batch(mols, jess._device)
for label, c in (("default cutoffs", cut), ("-j 2 2.5 2.5", np.full(len(idx), 2.5))):
    best = None
    for rep in range(3):
        t0 = time.perf_counter()
        res = _native.query(packed.native, batch, idx, 2.0, c, c, 10000, base | _native.TM_FLAG_COUNT_CANDIDATES)
        hits, cands = res.hits(), res.candidates()
        wall = time.perf_counter() - t0
        if best is None or wall < best[0]:
            best = (wall, res.timing(), len(hits), int((cands & 0x7fffffff).sum()), int((cands >> 31).sum()))
        res.close()
    wall, timing, n_hits, n_cand, n_trunc = best
    print(f"configs[4] {label}: {n / wall:.1f} structures/s ({1e3 * wall / n:.1f} ms per complex; index+tables "
          f"{timing[0]:.1f} ms, search {timing[1]:.1f} ms, finish {timing[2]:.1f} ms per batch of {n}), {n_hits} hits, "
          f"{n_cand} candidates, {n_trunc} truncated pairs", flush=True)
batch.close()
# one complex alone (latency of the largest single query the path sees)
single = pyjess.make_batch(mols[:1], jess._device)
loose = np.full(len(idx), 2.5)
best = None
for rep in range(4):
    t0 = time.perf_counter()
    res = _native.query(packed.native, single, idx, 2.0, loose, loose, 10000, base)
    n_hits = res.hit_count()
    wall = time.perf_counter() - t0
    if best is None or wall < best[0]:
        best = (wall, res.timing(), res.search_split(), n_hits)
    res.close()
single.close()
print(f"configs[4] one complex alone, -j 2 2.5 2.5: {1e3 * best[0]:.1f} ms wall (index+tables {best[1][0]:.1f} ms, search "
      f"{best[1][1]:.1f} ms = block {best[2][0]:.1f} + depth-first {best[2][1]:.1f}, finish {best[1][2]:.1f} ms), {best[3]} hits",
      flush=True)
