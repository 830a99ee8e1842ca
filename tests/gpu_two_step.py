"""The reference's two-step flow, load_molecules + Matcher.run, end to end. Not a test."""
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import jess_run, synthetic, template  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
matcher = jess_run.Matcher(template.load_packed_library())
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i, m in enumerate(synthetic.structures(n)):
        p = Path(d) / f"s{i:05d}.pdb"
        p.write_text(synthetic.to_pdb(m))
        paths.append(p)
    for rep in range(3):
        t0 = time.perf_counter()
        mols = jess_run.load_molecules(paths)
        t1 = time.perf_counter()
        out = matcher.run(mols)
        t2 = time.perf_counter()
        print(f"rep {rep}: load_molecules {1e3 * (t1 - t0):.0f} ms, Matcher.run {1e3 * (t2 - t1):.0f} ms "
              f"-> {n / (t2 - t0):.0f} structures/s ({sum(len(v) for v in out.values())} matches)", flush=True)
