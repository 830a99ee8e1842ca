"""cProfile of Matcher.run_files (host side of the files -> matches pipeline). Not a test."""
import cProfile
import pstats
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import jess_run, synthetic, template  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
matcher = jess_run.Matcher(template.load_packed_library())
with tempfile.TemporaryDirectory() as d:
    base = [synthetic.to_pdb(m).encode() for m in synthetic.structures(min(n, 1000))]
    paths = []
    for i in range(n):
        p = Path(d) / f"s{i:05d}.pdb"
        p.write_bytes(base[i % len(base)])
        paths.append(p)
    matcher.run_files(paths[:1000])  # warm-up: library upload, staging
    prof = cProfile.Profile()
    t0 = time.perf_counter()
    prof.enable()
    out = matcher.run_files(paths)
    prof.disable()
    dt = time.perf_counter() - t0
    print(f"{n / dt:.0f} structures/s under the profiler, {sum(len(v) for v in out.values())} matches")
    pstats.Stats(prof).sort_stats("cumulative").print_stats(28)
