"""Timeline of Matcher.run_files' three stages (ingest thread / submit / collect), one line per batch. Not a test."""
import sys
import tempfile
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import ingest, jess_run, synthetic, template  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
workers = int(sys.argv[4]) if len(sys.argv) > 4 else 2
events = []
t_origin = [0.0]


def timed(label, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            events.append((label, threading.current_thread().name[:6], t0 - t_origin[0], time.perf_counter() - t_origin[0]))
    return wrapper


ingest.PdbBatch.from_paths = staticmethod(timed("ingest", ingest.PdbBatch.from_paths))
jess_run.Matcher._submit = timed("submit", jess_run.Matcher._submit)
jess_run.Matcher._collect = timed("collect", jess_run.Matcher._collect)

matcher = jess_run.Matcher(template.load_packed_library())
with tempfile.TemporaryDirectory() as d:
    base = [synthetic.to_pdb(m).encode() for m in synthetic.structures(min(n, 1000))]
    paths = []
    for i in range(n):
        p = Path(d) / f"s{i:05d}.pdb"
        p.write_bytes(base[i % len(base)])
        paths.append(p)
    matcher.run_files(paths[:2 * bs], batch_size=bs)
    for rep in range(reps):
        events.clear()
        t_origin[0] = t0 = time.perf_counter()
        out = matcher.run_files(paths, batch_size=bs, workers=workers, prefetch=max(2, workers))
        dt = time.perf_counter() - t0
        print(f"rep {rep}: {n / dt:.0f} structures/s, {sum(len(v) for v in out.values())} matches, wall {dt * 1e3:.1f} ms")
        by = {}
        for label, thread, a, b in events:
            by.setdefault(label, []).append((a * 1e3, b * 1e3))
        for label, spans in by.items():
            print(f"  {label:8s} " + " ".join(f"{a:6.1f}-{b:6.1f}" for a, b in spans))
        for label, spans in by.items():
            durs = [b - a for a, b in spans]
            print(f"  {label:8s} mean {sum(durs) / len(durs):6.2f} ms  max {max(durs):6.2f}  sum {sum(durs):7.1f}")
