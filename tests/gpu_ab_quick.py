"""Quick A/B of library builds / environment settings on the headline batch: the exclusive durations of a query's phases
(HIP events, one query in flight) per variant, each variant in a fresh process.  Not a test.

    python tests/gpu_ab_quick.py <rounds> <spec> ...      spec = <lib suffix>[@ENV=VALUE[,ENV=VALUE...]]
    ('' or 'default' = enzymm_amd/libenzymm_amd.so, x = enzymm_amd/libenzymm_amd_x.so)
"""
import hashlib
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, str(ROOT))
    import numpy as np
    from tests import helpers
    from enzymm_amd import pyjess, synthetic, _native
    import time
    n = int(sys.argv[2])
    t0 = time.time()
    idx, cut = helpers.default_selection()
    jess = pyjess.Jess(helpers.product_library_templates())
    t1 = time.time()
    mols = synthetic.structures(n)
    t2 = time.time()
    packed = pyjess._packed(jess._templates, jess._device)
    batch = pyjess.make_batch(mols, jess._device)
    t3 = time.time()
    if os.environ.get("AB_VERBOSE"):
        print("setup: library %.1f s, structures %.1f s, device %.1f s" % (t1 - t0, t2 - t1, t3 - t2), file=sys.stderr)
    flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | _native.TM_FLAG_REBUILD_INDEX
    rows = []
    digest = None
    for i in range(14):
        res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags)
        t = res.timing()
        f, d = res.search_split()
        if i == 0:
            h = res.hits()
            digest = hashlib.sha1(np.ascontiguousarray(h[["structure", "template_sel", "n_atoms"]]).tobytes() + h["atoms"].tobytes()).hexdigest()[:12] + " %d hits" % len(h)
        res.close()
        if i >= 4:
            rows.append((t[0], f, d, t[2]))
    a = np.array(rows)
    med, lo = np.median(a, axis=0), a.min(axis=0)
    print("index %.3f  k_search %.3f (min %.3f)  k_deep %.3f  finish %.3f  sum %.3f ms  %s" %
          (med[0], med[1], lo[1], med[2], med[3], med.sum(), digest), flush=True)
    sys.exit(0)

rounds = int(sys.argv[1])
specs = sys.argv[2:]
for _ in range(rounds):
    for spec in specs:
        name, _, envs = spec.partition("@")
        name = "" if name == "default" else name
        env = dict(os.environ)
        env["ENZYMM_AMD_LIB"] = str(ROOT / "enzymm_amd" / ("libenzymm_amd%s.so" % ("_" + name if name else "")))
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        out = subprocess.run([sys.executable, __file__, "--child", os.environ.get("AB_STRUCTURES", "1000")], env=env, capture_output=True, text=True)
        line = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED: " + out.stderr.strip()[-300:]
        print("%-30s %s" % (spec or "default", line), flush=True)
