"""The three end-to-end rates README.md and DESIGN.md quote, in one run: Matcher.run_files on files, Matcher.run on a list
of host-parsed molecules (pipelined batches), and the reference's two-step flow load_molecules + Matcher.run.  Not a test.

    python tests/gpu_end_to_end.py [files] [molecules]"""
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import jess_run, synthetic, template  # noqa: E402

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
n_mols = int(sys.argv[2]) if len(sys.argv) > 2 else 8000
matcher = jess_run.Matcher(template.load_packed_library())
base = synthetic.structures(1000)
with tempfile.TemporaryDirectory() as d:
    texts = [synthetic.to_pdb(m).encode() for m in base]
    paths = []
    for i in range(n_files):
        p = Path(d) / f"s{i:05d}.pdb"
        p.write_bytes(texts[i % len(texts)])
        paths.append(p)
    matcher.run_files(paths[:2000], batch_size=1000)
    for rep in range(3):
        t0 = time.perf_counter()
        out = matcher.run_files(paths, batch_size=1000)
        dt = time.perf_counter() - t0
        print(f"run_files, {n_files} files: {n_files / dt:.0f} structures/s ({sum(len(v) for v in out.values())} matches)", flush=True)
    for rep in range(3):
        t0 = time.perf_counter()
        mols = jess_run.load_molecules(paths[:1000])
        t1 = time.perf_counter()
        out = matcher.run(mols)
        t2 = time.perf_counter()
        print(f"two-step flow, 1000 files: load_molecules {1e3 * (t1 - t0):.0f} ms + Matcher.run {1e3 * (t2 - t1):.0f} ms -> "
              f"{1000 / (t2 - t0):.0f} structures/s", flush=True)
mols = [base[i % len(base)].copy() for i in range(n_mols)]
matcher.run(mols[:2000])
for rep in range(3):
    t0 = time.perf_counter()
    out = matcher.run(mols)
    dt = time.perf_counter() - t0
    print(f"Matcher.run, {n_mols} host-parsed molecules: {n_mols / dt:.0f} structures/s ({sum(len(v) for v in out.values())} matches)", flush=True)
