"""Pipelined ingestion + search over PDB files (Matcher.run_files) vs one batch after the other. Not a test."""
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import ingest, jess_run, synthetic, template  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
matcher = jess_run.Matcher(template.load_packed_library())
with tempfile.TemporaryDirectory() as d:
    paths = []
    base = [synthetic.to_pdb(m).encode() for m in synthetic.structures(min(n, 1000))]
    for i in range(n):
        p = Path(d) / f"s{i:05d}.pdb"
        p.write_bytes(base[i % len(base)])
        paths.append(p)
    def serial():
        hits = 0
        for i in range(0, n, bs):
            pb = ingest.PdbBatch.from_paths(paths[i:i + bs], conservation_cutoff=0)
            hits += sum(len(v) for v in matcher.run(pb).values())
            pb.close()
        return hits

    def pipelined():
        return sum(len(v) for v in matcher.run_files(paths, batch_size=bs).values())

    for label, fn in (("one batch after the other", serial), ("next batch ingested on a worker thread", pipelined)):
        for rep in range(3):
            t0 = time.perf_counter()
            hits = fn()
            t1 = time.perf_counter()
            print(f"{label}, rep {rep}: {n / (t1 - t0):.0f} structures/s ({hits} matches), files -> matches", flush=True)
