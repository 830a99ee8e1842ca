"""Matcher.run_files with the native phase trace (ENZYMM_AMD_TRACE=1) and the main thread's steps on one clock: which
thread did what when.  Not a test.    ENZYMM_AMD_TRACE=1 python tests/gpu_files_timeline.py [files] [workers] 2> trace.txt"""
import sys
import tempfile
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import ingest, jess_run, synthetic, template  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 3
events = []
origin = [0.0]


def timed(label, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            events.append((label, threading.current_thread().name[-8:], 1e3 * (t0 - origin[0]), 1e3 * (time.perf_counter() - origin[0])))
    return wrapper


ingest.PdbBatch.from_paths = classmethod(timed("ingest", ingest.PdbBatch.from_paths.__func__))
ingest.PdbBatch.close = timed("close", ingest.PdbBatch.close)
jess_run.Matcher._submit = timed("submit", jess_run.Matcher._submit)
jess_run.Matcher._collect = timed("collect", jess_run.Matcher._collect)
matcher = jess_run.Matcher(template.load_packed_library())
base = synthetic.structures(1000)
with tempfile.TemporaryDirectory() as d:
    texts = [synthetic.to_pdb(m).encode() for m in base]
    paths = []
    for i in range(n):
        p = Path(d) / f"s{i:05d}.pdb"
        p.write_bytes(texts[i % len(texts)])
        paths.append(p)
    matcher.run_files(paths[:2000], batch_size=1000)
    for rep in range(2):
        del events[:]
        print(f"=== rep {rep}", file=sys.stderr, flush=True)
        origin[0] = t0 = time.perf_counter()
        out = matcher.run_files(paths, batch_size=1000, workers=workers, prefetch=workers)
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {n / dt:.0f} structures/s, wall {1e3 * dt:.1f} ms")
        for label, thread, a, b in sorted(events, key=lambda e: e[2]):
            print(f"  {a:8.2f} - {b:8.2f}  {b - a:6.2f}  {label:8s} {thread}")
