"""cProfile of load_molecules on 1000 PDB files (the reference's own loader API, host side only).  Not a test."""
import cProfile
import pstats
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import jess_run, synthetic  # noqa: E402

base = synthetic.structures(1000)
with tempfile.TemporaryDirectory() as d:
    paths = []
    for i, m in enumerate(base):
        p = Path(d) / f"s{i:05d}.pdb"
        p.write_bytes(synthetic.to_pdb(m).encode())
        paths.append(p)
    jess_run.load_molecules(paths[:100])
    for _ in range(3):
        t0 = time.perf_counter()
        jess_run.load_molecules(paths)
        print("load_molecules, 1000 files: %.1f ms" % (1e3 * (time.perf_counter() - t0)), flush=True)
    prof = cProfile.Profile()
    prof.enable()
    jess_run.load_molecules(paths)
    prof.disable()
    pstats.Stats(prof).sort_stats("tottime").print_stats(12)
