"""bench.py's upload-inclusive pass (host atom tables -> device batch on a worker thread while the previous batch is
searched) with the native phase trace of the batch creation.  Not a test.    ENZYMM_AMD_TRACE=1 python tests/gpu_streamed_trace.py"""
import queue
import sys
import threading
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import helpers  # noqa: E402
from enzymm_amd import pyjess, synthetic, _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
producers = int(sys.argv[2]) if len(sys.argv) > 2 else 1
idx, cut = helpers.default_selection()
jess = pyjess.Jess(helpers.product_library_templates())
packed = pyjess._packed(jess._templates, jess._device)
mols = synthetic.structures(1000)
chunks = [mols for _ in range(n // 1000)]
flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | _native.TM_FLAG_REBUILD_INDEX


def one_pass():
    q = queue.Queue(maxsize=2)
    todo = queue.Queue()
    for c in chunks:
        todo.put(c)

    def producer():
        while True:
            try:
                c = todo.get_nowait()
            except queue.Empty:
                break
            t0 = time.perf_counter()
            b = pyjess.make_batch(c, jess._device)
            q.put((b, time.perf_counter() - t0))
    threads = [threading.Thread(target=producer, daemon=True) for _ in range(producers)]
    for t in threads:
        t.start()
    made, fresh = [], []
    for _ in chunks:
        b, dt = q.get()
        made.append(dt)
        fresh.append(b)
        res = _native.query(packed.native, b, idx, 2.0, cut, cut, 10000, flags, retry=False)
        res.hits()
        res.close()
        if len(fresh) > 2:
            fresh.pop(0).close()
    for b in fresh:
        b.close()
    return made


one_pass()
for rep in range(3):
    t0 = time.perf_counter()
    made = one_pass()
    dt = time.perf_counter() - t0
    print("%d structures, %d producer(s): %.0f structures/s; make_batch %.1f ms each" % (n, producers, n / dt, 1e3 * sum(made) / len(made)), flush=True)
