"""End-to-end pipeline timing: PDB text -> device ingestion -> search + fused filter -> TSV (not a pytest test)."""
import io
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import ingest, jess_run, report, synthetic, template  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
t0 = time.perf_counter()
lib = template.load_packed_library()
matcher = jess_run.Matcher(lib)
t1 = time.perf_counter()
print(f"library + matcher: {t1 - t0:.2f} s", flush=True)
texts = [synthetic.to_pdb(m).encode() for m in synthetic.structures(n)]
ids = [f"s{i:05d}" for i in range(n)]
for rep in range(2):
    t2 = time.perf_counter()
    pb = ingest.PdbBatch(texts, ids)
    t3 = time.perf_counter()
    out = matcher.run(pb)
    t4 = time.perf_counter()
    buf = io.StringIO()
    rows = report.write_tsv(out, buf)
    t5 = time.perf_counter()
    print(f"rep {rep}: ingest {1e3 * (t3 - t2):.0f} ms, Matcher.run {1e3 * (t4 - t3):.0f} ms, "
          f"TSV ({rows} rows) {1e3 * (t5 - t4):.0f} ms -> {n / (t5 - t2):.0f} structures/s end to end", flush=True)
    pb.close()
