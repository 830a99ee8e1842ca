"""Where a batch's time goes in Matcher.run_files and Matcher.run(list): wall-clock spans of the Python-level steps per
batch (ingest thread: native constructor / the rest of PdbBatch; main thread: submit, wait for the result, annotate_hits,
the column replay).  Not a test.    ENZYMM_AMD_TRACE=1 python tests/gpu_files_profile.py [files] [workers]"""
import sys
import tempfile
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from enzymm_amd import _native, ingest, jess_run, pyjess, synthetic, template  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
workers = int(sys.argv[2]) if len(sys.argv) > 2 else 2
spans = {}
lock = threading.Lock()


def timed(label, fn):
    def wrapper(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            with lock:
                spans.setdefault(label, []).append(time.perf_counter() - t0)
    return wrapper


_native.Batch.from_files = classmethod(timed("native from_files", _native.Batch.from_files.__func__))
_native.Batch.from_structures = classmethod(timed("native from_structures", _native.Batch.from_structures.__func__))
ingest.PdbBatch.from_paths = classmethod(timed("PdbBatch.from_paths", ingest.PdbBatch.from_paths.__func__))
ingest.PdbBatch.annotate_hits = timed("annotate_hits", ingest.PdbBatch.annotate_hits)
pyjess.make_batch = timed("make_batch", pyjess.make_batch)
jess_run.Matcher._submit = timed("submit", jess_run.Matcher._submit)
jess_run.Matcher._collect = timed("collect", jess_run.Matcher._collect)
jess_run.Matcher._collect_table = timed("collect_table", jess_run.Matcher._collect_table)
pyjess._PendingQuery.result = timed("pending.result", pyjess._PendingQuery.result)

matcher = jess_run.Matcher(template.load_packed_library())
base = synthetic.structures(1000)


def report(title, count, dt):
    print(f"{title}: {count / dt:.0f} structures/s")
    for label, v in sorted(spans.items()):
        print(f"   {label:24s} n {len(v):3d}  mean {1e3 * sum(v) / len(v):7.2f} ms  max {1e3 * max(v):7.2f}")
    spans.clear()


with tempfile.TemporaryDirectory() as d:
    texts = [synthetic.to_pdb(m).encode() for m in base]
    paths = []
    for i in range(n):
        p = Path(d) / f"s{i:05d}.pdb"
        p.write_bytes(texts[i % len(texts)])
        paths.append(p)
    matcher.run_files(paths[:2000], batch_size=1000)
    spans.clear()
    for rep in range(3):
        t0 = time.perf_counter()
        out = matcher.run_files(paths, batch_size=1000, workers=workers, prefetch=max(2, workers))
        report(f"run_files, {n} files, {workers} workers", n, time.perf_counter() - t0)
mols = [base[i % len(base)].copy() for i in range(n)]
matcher.run(mols[:2000])
spans.clear()
for rep in range(3):
    t0 = time.perf_counter()
    out = matcher.run(mols)
    report(f"Matcher.run, {n} host-parsed molecules", n, time.perf_counter() - t0)
