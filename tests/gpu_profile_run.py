"""cProfile of Matcher.run on a list of host-parsed molecules (the reference's two-step flow). Not a test."""
import cProfile
import pstats
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
    mols = jess_run.load_molecules(paths)
    matcher.run(mols)
    for rep in range(2):
        t0 = time.perf_counter()
        out = matcher.run(mols)
        print(f"Matcher.run: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    prof = cProfile.Profile()
    prof.enable()
    out = matcher.run(mols)
    prof.disable()
    pstats.Stats(prof).sort_stats("cumulative").print_stats(25)
