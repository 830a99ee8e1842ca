"""The reference's call pattern (a thread pool maps Jess(bucket).query(molecule) over the molecules, bucket after bucket:
/root/reference/enzymm/jess_run.py:896-988) through enzymm_amd.pyjess, with the launches the sweep planner made.  Not a test.

    python tests/gpu_reference_pattern.py [molecules] [threads]"""
import sys
import time
from multiprocessing.pool import ThreadPool
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, template  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
threads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
lib = template.load_packed_library()
templates = [lib[i] for i in range(len(lib))]
eff = np.array([t.effective_size for t in templates])
by_size = {}
for t, e in zip(templates, eff):
    if e >= 3:
        by_size.setdefault(int(e), []).append(t)
sizes = sorted(by_size, reverse=True)
mols = synthetic.structures(n + threads)


def sweep(ms):
    per_bucket = []
    with ThreadPool(threads) as pool:
        for size in sizes:
            tpls, d = by_size[size], helpers.DEFAULT_DISTANCE[min(size, 8)]

            def one(mol, tpls=tpls, d=d):
                return list(pyjess.Jess(tpls).query(molecule=mol, rmsd_threshold=2.0, distance_cutoff=d, max_dynamic_distance=d,
                                                    max_candidates=10000, best_match=True, ignore_chain=True))

            t0 = time.perf_counter()
            hits = sum(len(h) for h in pool.imap(one, ms))
            per_bucket.append((size, time.perf_counter() - t0, hits))
    return per_bucket


sweep(mols[n:])  # warm: libraries packed, selections staged, the six keys known
import gc  # noqa: E402
import os  # noqa: E402
if os.environ.get("GC_MODE") == "freeze":    # the library's quarter of a million objects out of the collector's sight
    gc.collect(); gc.freeze()
elif os.environ.get("GC_MODE") == "off":
    gc.disable()
gc_runs = []
gc.callbacks.append(lambda phase, info: gc_runs.append((info["generation"], time.perf_counter())) if phase == "start" else None)
for rep in range(3):
    ms = mols[:n] if rep == 0 else synthetic.structures(n, start=1000 * rep)
    del pyjess._sweeps.launches[:]
    del gc_runs[:]
    t0 = time.perf_counter()
    per_bucket = sweep(ms)
    dt = time.perf_counter() - t0
    launches = list(pyjess._sweeps.launches)
    print("%d molecules x %d buckets from %d threads: %.1f ms = %.0f structures/s; %d launches (%.1f ms inside them, %.1f molecules each)" %
          (n, len(sizes), threads, dt * 1e3, n / dt, len(launches), 1e3 * sum(l[3] for l in launches),
           sum(l[0] for l in launches) / max(1, len(launches))))
    print("   garbage collections during the sweep, by generation:", [sum(1 for g, _ in gc_runs if g == k) for k in range(3)])
    print("   per bucket (size, ms, hits):", [(s, round(t * 1e3, 1), h) for s, t, h in per_bucket])
    print("   launch ms:", [round(l[3] * 1e3, 2) for l in launches][:24])
    print("   of a launch (device copy, enqueue, wait + table) ms:", np.round(np.mean([l[4] for l in launches[2:]], axis=0), 3))
