"""run_files over 16 000 files with 2-4 ingest threads and 3-5 batches ahead (all within 165-192 k structures/s on one box:
the defaults, 3 and 3, stay).  Not a test."""
import sys, tempfile, time
from pathlib import Path
sys.path.insert(0, '.')
from enzymm_amd import jess_run, synthetic, template
matcher = jess_run.Matcher(template.load_packed_library())
base = synthetic.structures(1000)
with tempfile.TemporaryDirectory() as d:
    texts = [synthetic.to_pdb(m).encode() for m in base]
    paths = []
    for i in range(16000):
        p = Path(d) / f"s{i:05d}.pdb"; p.write_bytes(texts[i % 1000]); paths.append(p)
    matcher.run_files(paths[:2000], batch_size=1000)
    for w, pf in ((3, 3), (4, 4), (2, 3), (3, 4), (4, 5), (3, 3)):
        best = []
        for rep in range(3):
            t0 = time.perf_counter()
            out = matcher.run_files(paths, batch_size=1000, workers=w, prefetch=pf)
            best.append(16000 / (time.perf_counter() - t0))
        print(f"workers {w} prefetch {pf}: " + " ".join(f"{v/1e3:.0f}k" for v in best), flush=True)
