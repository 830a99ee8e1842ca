"""Throughput of the device-side PDB ingestion (not a pytest test)."""
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import helpers  # noqa: E402
from enzymm_amd import ingest, pyjess, synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
jess = pyjess.Jess(helpers.product_library_templates())
pyjess._packed(jess._templates, jess._device)  # registers the library's vocabulary
texts = [synthetic.to_pdb(m).encode() for m in synthetic.structures(n)]
total = sum(len(t) for t in texts)
ids = [f"s{i}" for i in range(n)]
for rep in range(3):
    t0 = time.perf_counter()
    pb = ingest.PdbBatch(texts, ids)
    dt = time.perf_counter() - t0
    print(f"PdbBatch: {n} files, {total / 1e6:.1f} MB in {dt * 1e3:.1f} ms = {total / dt / 1e9:.2f} GB/s "
          f"({n / dt:.0f} files/s) incl. the H2D copies", flush=True)
    pb.close()
t0 = time.perf_counter()
mols = [pyjess.Molecule.loads(t.decode()) for t in texts[:50]]
dt = time.perf_counter() - t0
print(f"host Molecule.loads: {50 / dt:.0f} files/s on one core")
