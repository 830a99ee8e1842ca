#!/usr/bin/env python3
"""Headline benchmark: structures/s against the full shipped M-CSA template library.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2]): per GPU a batch of 1000 seeded synthetic
~300-residue structures (enzymm_amd/synthetic.py), each matched against the 6780 default-eligible
templates of jess_templates_20230210 (effective size >= 3) in their six size buckets with enzymm's
default parameters (jess_run.py:564-571), best_match=True, ignore_chain=True,
max_candidates=10000 (jess_run.py:803-811).  One step = one full pass over the resident batch:
candidate-index build + template staging + search + superposition + best-hit selection (+ the RCCL gather of the hit tables to rank 0
when N > 1).  Inputs are uploaded to HBM before the timed region.

For N > 1 the driver launches this file under torch.distributed.run, one rank per GPU; ranks
shard the structures (weak scaling: 1000 structures per GPU), there is no data-path collective
other than ONE gather of the hit table per pass: every search appends its rows to a table in the
rank's HBM (tm_result_append_hits, no host wait between searches) and the table goes to rank 0
when the pass is done (distributed.DeviceHitTable) -- while the next pass's searches already run.
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

DEFAULT_DISTANCE = {3: 0.9, 4: 1.7, 5: 2.0, 6: 2.0, 7: 2.0, 8: 2.0}  # jess_run.py:564-571
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8 TB/s


def load_library():
    """Shipped templates as enzymm_amd Templates + the default selection (index, cutoff)."""
    from enzymm_amd import template as tmod

    templates = tmod.load_packed_library()
    eff = np.array([t.effective_size for t in templates])
    idx = np.nonzero(eff >= 3)[0].astype(np.uint32)
    cut = np.array([DEFAULT_DISTANCE[min(int(e), 8)] for e in eff[idx]], dtype=np.float64)
    return templates, idx, cut


def reference_baseline(molecules, idx, cut, target_seconds):
    """PyJess itself, if this host happens to have it (it is not in the image this repo is built in): the
    reference's own call, jess_run.py:800-811, one Jess per distinct threshold, threads over structures.
    Returns None when PyJess is absent or its API is not what enzymm 's pyproject pins."""
    try:
        import importlib
        from concurrent.futures import ThreadPoolExecutor
        real = importlib.import_module("pyjess")
        if "enzymm_amd" in (getattr(real, "__file__", "") or ""):
            return None
        from enzymm_amd import synthetic
        from oracle import parse
        texts = parse.read_packed_library(ROOT / "enzymm_amd" / "data" / "jess_templates_20230210.txt.gz")
        by_cut = {}
        for i, c in zip(idx, cut):
            by_cut.setdefault(float(c), []).append(real.Template.loads(texts[i][1], id=str(i)))
        engines = [(c, real.Jess(tpls)) for c, tpls in by_cut.items()]
        cores = usable_cores()

        def one(mol):
            m = real.Molecule.loads(synthetic.to_pdb(mol))
            return sum(len(list(j.query(m, 2.0, c, c, max_candidates=10000, best_match=True, ignore_chain=True)))
                       for c, j in engines)

        done, t0 = 0, time.perf_counter()
        with ThreadPoolExecutor(cores) as pool:
            while done < len(molecules) and time.perf_counter() - t0 < target_seconds:
                chunk = molecules[done: done + cores]
                list(pool.map(one, chunk))
                done += len(chunk)
        dt = time.perf_counter() - t0
        try:
            dump_reference_vectors(real, texts)
        except Exception:
            pass
        return dict(value=done / dt, unit="structures/s", cores=cores, kind="reference",
                    sample=f"first {done} structures of the bench batch vs the same 6780 templates, "
                           f"pyjess {getattr(real, '__version__', '?')} Jess.query on {cores} threads ({dt:.1f} s)")
    except Exception:  # absent, or not the API of pyjess ~= 0.5
        return None


def dump_reference_vectors(real, texts):
    """Should a box ever have the real PyJess: answers to what the reference's own tests leave unpinned, written
    to gpurun_out/pyjess_golden_vectors.json so that they can be committed as fixtures -- match mode 1 (treated like 3 here),
    one-letter X / U / O residue lists, the dynamic distance with weights, ignore_chain=False, the all-hits mode, and
    log_evalue, on the two structures the reference ships.  Inputs are named by template id and file; outputs are hits as
    (rmsd, log_evalue, atom serials).  Never runs in this image (no pyjess); never fails the bench."""
    golden = ROOT / "tests" / "golden"
    molecules = {name: real.Molecule.load(str(golden / name)) for name in ("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb")}
    picks, seen = [], set()
    for name, text in texts:  # a few templates of every kind of interest, by what their ATOM lines hold
        lines = [l for l in text.splitlines() if l.startswith("ATOM")]
        modes = {int(l[6:11]) for l in lines}
        letters = set("".join(l[54:60].strip() for l in lines))
        weighted = any(l[60:66].strip() not in ("", "0.00", "0.0", "0") for l in lines)
        for tag, hit in (("mode1", 1 in modes), ("mode100", any(m >= 100 for m in modes)), ("X", "X" in letters),
                         ("U", "U" in letters), ("O", "O" in letters), ("weighted", weighted)):
            if hit and sum(1 for t, _ in picks if t == tag) < 4 and (tag, name) not in seen:
                picks.append((tag, name)); seen.add((tag, name))
    by_name = dict(texts)
    out = {"pyjess_version": getattr(real, "__version__", None), "cases": []}
    settings = (dict(rmsd_threshold=2.0, distance_cutoff=1.5, max_dynamic_distance=1.5, max_candidates=10000, best_match=True, ignore_chain=True),
                dict(rmsd_threshold=2.0, distance_cutoff=1.0, max_dynamic_distance=2.0, max_candidates=10000, best_match=True, ignore_chain=True),
                dict(rmsd_threshold=2.0, distance_cutoff=1.5, max_dynamic_distance=1.5, max_candidates=10000, best_match=True, ignore_chain=False),
                dict(rmsd_threshold=2.0, distance_cutoff=1.5, max_dynamic_distance=1.5, max_candidates=50, best_match=False, ignore_chain=True))
    for tag, name in picks:
        tpl = real.Template.loads(by_name[name], id=name)
        for mname, mol in molecules.items():
            for kw in settings:
                hits = list(real.Jess([tpl]).query(mol, **kw))
                out["cases"].append({"kind": tag, "template": name, "molecule": mname, "arguments": kw,
                                     "hits": [{"rmsd": h.rmsd, "log_evalue": getattr(h, "log_evalue", None),
                                               "serials": [a.serial for a in h.atoms(transform=False)]} for h in hits]})
    target = ROOT / "gpurun_out"
    target.mkdir(exist_ok=True)
    (target / "pyjess_golden_vectors.json").write_text(json.dumps(out, indent=1) + "\n")


def usable_cores() -> int:
    """CPUs this process may actually use: the affinity mask, cut down to the cgroup's CPU quota (a GPU box can
    show 256 CPUs in its mask and grant the container a handful)."""
    cores = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            text = open(path).read().split()
            if path.endswith("cpu.max"):
                quota, period = text[0], float(text[1])
            else:
                quota, period = text[0], float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota not in ("max", "-1"):
                cores = max(1, min(cores, int(float(quota) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return cores


def cpu_baseline(molecules, idx, cut, target_seconds, repeats=3):
    """Oracle (CPU restatement, NOT PyJess) timed on this host's cores as SURVEY.md 8(d) prescribes: one structure
    per task, one single-threaded oracle call per task, as many tasks in flight as the host has cores
    (ctypes releases the GIL), wall clock over `repeats` passes of the same bounded sample."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle, parse

    texts = parse.read_packed_library(ROOT / "enzymm_amd" / "data" / "jess_templates_20230210.txt.gz")
    otpl = [parse.parse_template(t) for _, t in texts]
    olib = oracle.OracleLibrary([otpl[i] for i in idx])
    cores = usable_cores()

    def as_oracle(mol):
        c = mol._cols
        return parse.OracleMolecule(mol._xyz, c["name"], c["residue_name"], c["chain_id"], c["residue_number"],
                                    c["insertion_code"], c["temperature_factor"], c["serial"])

    def one(om):
        return oracle.query_library(om, olib, 2.0, cut, cut, max_candidates=10000, ignore_chain=True, threads=1)

    # calibrate: one structure alone, then one structure per core in flight; the sample is sized from the loaded rate
    # so that all repeats together take about target_seconds
    t0 = time.perf_counter()
    first = one(as_oracle(molecules[0]))
    per_structure = time.perf_counter() - t0
    rates, results = [], None
    with ThreadPoolExecutor(cores) as pool:
        probe = [as_oracle(m) for m in molecules[:min(cores, len(molecules))]]
        t0 = time.perf_counter()
        list(pool.map(one, probe))
        loaded_rate = len(probe) / (time.perf_counter() - t0)
        sample = int(min(len(molecules), max(len(probe), loaded_rate * target_seconds / repeats)))
        oms = [as_oracle(m) for m in molecules[:sample]]
        for _ in range(repeats):
            t0 = time.perf_counter()
            results = list(pool.map(one, oms))
            rates.append(sample / (time.perf_counter() - t0))
    del first
    value = float(np.mean(rates))
    return dict(value=value, unit="structures/s", cores=cores, kind="port", per_core=value / cores,
                repeats=[round(r, 2) for r in rates],
                sample=f"first {sample} structures of the bench batch vs the same {len(idx)} templates; "
                       f"oracle/jess_oracle.c, one structure per task on {cores} threads, {repeats} passes "
                       f"({sample / value:.1f} s each; one structure alone: {per_structure:.2f} s on one core)"), results


def untimed_extras(args, device, templates, idx, cut, molecules, search, flags):
    """Two rates that are never `value` (config.real_structures_per_s, config.reference_call_pattern_structures_per_s),
    each with its own oracle parity sample."""
    import torch

    from enzymm_amd import _native, pyjess, synthetic
    from oracle import oracle, parse

    out = {}
    texts = parse.read_packed_library(ROOT / "enzymm_amd" / "data" / "jess_templates_20230210.txt.gz")
    otpl = [parse.parse_template(t) for _, t in texts]
    olib = oracle.OracleLibrary([otpl[i] for i in idx])
    cores = usable_cores()

    def as_oracle(mol):
        c = mol._cols
        return parse.OracleMolecule(mol._xyz, c["name"], c["residue_name"], c["chain_id"], c["residue_number"],
                                    c["insertion_code"], c["temperature_factor"], c["serial"])

    def oracle_hits(mols):
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(cores) as pool:
            return list(pool.map(lambda m: oracle.query_library(as_oracle(m), olib, 2.0, cut, cut, max_candidates=10000,
                                                                ignore_chain=True, threads=1), mols))

    def mismatches(table_by_structure, oracle_results):
        mism = 0
        for s, r in enumerate(oracle_results):
            mine = table_by_structure.get(s, {})
            theirs = set(np.nonzero(r["hit"])[0].tolist())
            mism += len(theirs ^ set(mine))
            for ti in theirs & set(mine):
                k = int(mine[ti]["n_atoms"])
                if not np.array_equal(mine[ti]["atoms"][:k], r["atoms"][ti][:k]) or abs(mine[ti]["rmsd"] - r["rmsd"][ti]) > 1e-5:
                    mism += 1
        return mism

    # ---- (1) real proteins: rigid copies of the two structures the reference's tests ship (denser than the synthetic
    # batch: 161-164 atoms within 10 A against 140), the same resident-batch step as the headline
    real = [pyjess.Molecule.load(ROOT / "tests" / "golden" / f) for f in ("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb")]
    n_real = min(1000, args.structures)
    half = (n_real + 1) // 2
    copies = synthetic.rigid_copies(real[0], half, seed=7) + synthetic.rigid_copies(real[1], n_real - half, seed=8)
    order = np.random.default_rng(5).permutation(len(copies))
    copies = [copies[i] for i in order]
    batches = [pyjess.make_batch(copies, device) for _ in range(2)]
    streams = [_native.Stream(device) for _ in range(2)]

    def real_pass(n_steps):
        pending, tables = [], []
        for j in range(n_steps):
            pending.append(search(batches[j % 2], streams[j % 2].handle))
            if len(pending) == 2:
                r = pending.pop(0)
                tables.append(r.hits())
                r.close()
        for r in pending:
            tables.append(r.hits())
            r.close()
        return tables

    real_pass(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tables = real_pass(10)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    n_atoms_real = sum(len(m) for m in copies)
    sample = min(len(copies), max(2, cores))
    by_s = {}
    for h in tables[-1]:
        if int(h["structure"]) < sample:
            by_s.setdefault(int(h["structure"]), {})[int(h["template_sel"])] = h
    out["real_structures_per_s"] = len(copies) / dt
    out["real_atoms_per_s"] = n_atoms_real / dt
    out["real_workload"] = (f"{len(copies)} rigid copies of 1AMY (3339 atoms) and AF-P0DUB6 (4079 atoms), one resident batch, "
                            f"{n_atoms_real / len(copies):.0f} atoms per structure, {len(tables[-1])} hits per pass")
    out["real_parity_sample"] = {"structures": sample, "mismatches": mismatches(by_s, oracle_hits(copies[:sample]))}
    for b in batches:
        b.close()
    for st in streams:
        st.close()

    # ---- (2) the reference's own call pattern through the pyjess mirror: for every size bucket (descending), a
    # thread pool maps Jess(templates of that size).query(one molecule, ...) over the molecules
    # (/root/reference/enzymm/jess_run.py:896-988 with :785-843 inside); enzymm_amd.pyjess is what `import pyjess`
    # resolves to in the drop-in setup
    from multiprocessing.pool import ThreadPool

    eff = np.array([templates[int(i)].effective_size for i in idx])
    by_size = {}
    for i, e in zip(idx, eff):
        by_size.setdefault(int(e), []).append(templates[int(i)])
    sizes = sorted(by_size, reverse=True)
    mols = molecules[: min(len(molecules), 256)]
    threads = max(1, min(cores, 16))

    def reference_pattern(ms):
        found = {}
        with ThreadPool(threads) as pool:
            for size in sizes:
                tpls, d = by_size[size], DEFAULT_DISTANCE[min(size, 8)]

                def one(mol, tpls=tpls, d=d):
                    q = pyjess.Jess(tpls).query(molecule=mol, rmsd_threshold=2.0, distance_cutoff=d, max_dynamic_distance=d,
                                                max_candidates=10000, best_match=True, ignore_chain=True)
                    return list(q)

                for mi, hits in enumerate(pool.imap(one, ms)):
                    found.setdefault(mi, []).extend(hits)
        return found

    # cold: what enzymm's CLI does ONCE per process (/root/reference/enzymm/_cli.py:248 -> jess_run.py:896-988): one sweep
    # with nothing known -- the planner empty, the six buckets' template lists never seen (their device libraries are
    # packed and staged on the way), the molecules new objects.  HIP and the package are initialised (the headline ran).
    pyjess._sweeps.clear()
    cold_mols = [m.copy() for m in mols]
    del pyjess._sweeps.launches[:]
    t0 = time.perf_counter()
    found = reference_pattern(cold_mols)
    dt = time.perf_counter() - t0
    launches = list(pyjess._sweeps.launches)
    out["reference_call_pattern_cold_structures_per_s"] = len(cold_mols) / dt
    out["reference_call_pattern_cold"] = {"seconds": dt, "launches": len(launches),
                                          "molecules_searched": int(sum(l[0] for l in launches)),
                                          "ms_inside_launches": float(1e3 * sum(l[3] for l in launches)),
                                          "note": "first sweep of the process over 256 new molecules: six template lists packed "
                                                  "for the device and staged inside the timed region"}
    sample = min(len(cold_mols), max(2, cores))
    position = {id(templates[int(i)]): n for n, i in enumerate(idx)}
    by_s = {}
    for mi in range(sample):
        for h in found.get(mi, []):
            atoms = np.array(h._atom_index, dtype=np.uint32)
            by_s.setdefault(mi, {})[position[id(h.template)]] = {"n_atoms": len(atoms), "atoms": atoms, "rmsd": h.rmsd}
    cold_oracle = oracle_hits(cold_mols[:sample])
    out["reference_call_pattern_cold_parity_sample"] = {"structures": sample, "mismatches": mismatches(by_s, cold_oracle)}
    del cold_mols, found

    # warm: the libraries of the six buckets packed and their selections staged (first sweep), the library over all six the
    # sweep planner searches once the buckets are known (pyjess._Sweeps) built and staged (second sweep, other molecules)
    spare = molecules[len(mols):]
    for warm in (spare[:threads], spare[threads:2 * threads]) if len(spare) >= 2 * threads else (mols[:threads], mols[:threads]):
        reference_pattern(warm)
    # the timed sweep runs over molecules made just now, the way enzymm loads its molecules right before it matches them
    # (load_molecules -> Matcher.run): nothing about them can have been computed during the warm-up, every answer of the
    # sweep comes from a launch inside the timed region (config.reference_call_pattern_launches)
    mols = [m.copy() for m in mols]
    del pyjess._sweeps.launches[:]
    t0 = time.perf_counter()
    found = reference_pattern(mols)
    dt = time.perf_counter() - t0
    launches = list(pyjess._sweeps.launches)
    out["reference_call_pattern_structures_per_s"] = len(mols) / dt
    out["reference_call_pattern_launches"] = {"launches": len(launches), "molecules_searched": int(sum(l[0] for l in launches)),
                                              "ms_inside_launches": float(1e3 * sum(l[3] for l in launches))}
    out["reference_call_pattern"] = (f"{len(mols)} molecules x {len(sizes)} size buckets, Jess(templates).query(molecule) per "
                                     f"(molecule, size) from a ThreadPool of {threads}: {len(mols) * len(sizes) / dt:.0f} queries/s; "
                                     "the molecules are new objects (copies made right before the sweep), the sweep planner of "
                                     "enzymm_amd.pyjess searches them ahead -- all six buckets in one library -- from the second call on")
    # parity of that path: hits as (template index in idx order -> atoms) against the oracle
    sample = min(len(mols), max(2, cores))
    position = {id(templates[int(i)]): n for n, i in enumerate(idx)}
    by_s = {}
    for mi in range(sample):
        for h in found.get(mi, []):
            atoms = np.array(h._atom_index, dtype=np.uint32)
            by_s.setdefault(mi, {})[position[id(h.template)]] = {"n_atoms": len(atoms), "atoms": atoms, "rmsd": h.rmsd}
    out["reference_call_pattern_parity_sample"] = {"structures": sample,
                                                   "mismatches": mismatches(by_s, oracle_hits(mols[:sample]))}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--structures", type=int, default=1000, help="synthetic structures per GPU")
    ap.add_argument("--batch", type=int, default=1000, help="structures per resident batch (one search launch each)")
    ap.add_argument("--streamed-passes", type=int, default=48,
                    help="extra passes with the batches uploaded from host memory under the timer (reported in "
                         "config.streamed_structures_per_s, never in value); 0 = skip")
    ap.add_argument("--pipeline", type=int, default=2, choices=[1, 2, 3],
                    help="searches in flight per rank (2: launch j + 1 before fetching j's hit table; 1: one after the other)")
    ap.add_argument("--generator", default="v2", choices=["v2", "v1", "real"],
                    help="v2: natively packed synthetic globules (default); v1: the round-1 generator (under-dense, "
                         "kept for side-by-side rates); real: rigid-transform copies of 1AMY / AF-P0DUB6 (density reference)")
    ap.add_argument("--prewarm-seconds", type=float, default=0.5,
                    help="untimed passes before the W warm-up steps, so that the timed region starts at the GPU's working clock (0: none)")
    ap.add_argument("--sustained-seconds", type=float, default=5.0,
                    help="after the timed region, the same loop for at least this long (config.sustained_structures_per_s; 0: skip)")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the untimed extra rates (real-structure batch, the reference's per-molecule call pattern)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    # this rank's shard of the synthetic proteome (weak scaling: fixed work per GPU), generated on worker
    # processes BEFORE anything initialises the GPU
    from enzymm_amd import distributed, synthetic
    lo, hi = distributed.shard_range(args.structures * world, rank, world)
    workers = max(1, min(32, len(os.sched_getaffinity(0)) // max(1, world)))
    if args.generator == "real":
        from enzymm_amd import pyjess as _pj
        real = [_pj.Molecule.load(ROOT / "tests" / "golden" / f) for f in ("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb")]
        half = (hi - lo + 1) // 2
        molecules = (synthetic.rigid_copies(real[0], half, seed=lo) + synthetic.rigid_copies(real[1], hi - lo - half, seed=lo + 1))
    else:
        molecules = synthetic.structures(hi - lo, start=lo, processes=workers, generator=args.generator)

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback")
    # ENZYMM_AMD_BENCH_ONE_GPU=1: every rank on GPU 0 with a gloo group -- only to exercise the N > 1
    # control flow on a 1-GPU box; the driver's real runs use one GPU per rank over RCCL.
    one_gpu = os.environ.get("ENZYMM_AMD_BENCH_ONE_GPU") == "1"
    if one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    os.environ["ENZYMM_AMD_DEVICE"] = str(local_rank)
    # ENZYMM_AMD_BENCH_FORCE_SHARDED=1 (tests): --gpus 1 through the N > 1 branch -- the device hit table, its gather over
    # a world-size-1 "nccl" group (RCCL), the error exchange -- so that the loop the driver runs at N = 2, 4, 8 is
    # exercised over RCCL on the one GPU a test box has
    force_sharded = world == 1 and os.environ.get("ENZYMM_AMD_BENCH_FORCE_SHARDED") == "1"
    if world > 1 or force_sharded:
        if force_sharded:
            import socket
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as sock:
                    sock.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
        if one_gpu:
            # (the test hook's ranks meet on this host: gloo over the loopback interface, whatever the container's hostname
            # resolves to -- or fails to)
            if os.environ.get("MASTER_ADDR", "127.0.0.1") in ("127.0.0.1", "localhost"):
                os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    gather_device = "cpu" if one_gpu else f"cuda:{local_rank}"

    from enzymm_amd import _native, pyjess

    templates, idx, cut = load_library()
    jess = pyjess.Jess(templates, device=local_rank)
    packed = pyjess._packed(jess._templates, local_rank)

    n_atoms = sum(len(m) for m in molecules)
    # resident batches of at most --batch structures each (uploaded outside the timed region); the default workload
    # is one batch of 1000, BASELINE.json configs[3] (100 k structures over 8 GPUs) is 12 500 per rank in 13 batches
    chunks = [molecules[i:i + args.batch] for i in range(0, len(molecules), args.batch)]
    starts = [lo + i for i in range(0, len(molecules), args.batch)]
    batches = [pyjess.make_batch(chunk, local_rank) for chunk in chunks]
    # Steps are software-pipelined two deep, the way Matcher.run_files drives the library: search j + 1 is launched
    # (on the other of two streams) before the hit table of search j is fetched, so the host side of a launch and the
    # copy back run under the kernels of the neighbouring search.  Every search still does all of its work inside the
    # timed region (candidate index + pair tables + search + superposition + selection + hit table to the host).
    # Two searches in flight must not share a batch (TM_FLAG_REBUILD_INDEX rebuilds its index and tables in place):
    # a rank with a single resident batch keeps a second copy of it and alternates.
    resident_sets = [list(zip(starts, batches))]
    if len(batches) < args.pipeline:  # fewer resident batches than searches in flight: more copies, used in turn
        for _ in range(args.pipeline - 1):
            resident_sets.append([(st, pyjess.make_batch(ch, local_rank)) for st, ch in zip(starts, chunks)])
    streams = [_native.Stream(local_rank) for _ in range(max(1, args.pipeline))]
    flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | _native.TM_FLAG_REBUILD_INDEX
    capacity = max(4096, 64 * args.batch)
    cand_capacity = max(65536, 256 * args.batch)

    search_ms, index_ms, finish_ms, hit_total = [], [], [], 0
    front_ms, deep_ms = [], []  # search_ms split: block search over the pair tables, depth-first part
    last_hits = None

    def search(batch, stream=None):
        return _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags, hit_capacity=capacity,
                             stream=streams[0].handle if stream is None else stream,
                             candidate_capacity=cand_capacity, retry=False)

    # N > 1: the rank's rows of one pass collect in HBM and leave in one gather (two tables: pass j + 1 fills one while
    # pass j's is gathered).  Sized for 4x the rows a pass of this workload has been seen to produce.
    sharded = world > 1 or force_sharded
    pass_tables = None
    if sharded:
        rows_per_pass = max(16384, 16 * args.structures)
        pass_tables = [distributed.DeviceHitTable(rows_per_pass, device=f"cuda:{local_rank}") for _ in range(2)]
    gather_ms = []

    def run_steps(n_steps, record):
        """n_steps passes over this rank's structures: every batch searched (candidate index + pair tables + search
        + superposition + selection) with at most --pipeline searches in flight.
        N = 1: each search's hit table is fetched to the host in launch order.
        N > 1: each search appends its rows to the pass's device table; when the pass has been launched, the PREVIOUS
        pass is finished -- its results waited for (overflow checks), its table gathered to rank 0 in one exchange."""
        import collections
        nonlocal hit_total, last_hits
        in_flight = collections.deque()
        state = {}  # step -> [tables, times, hits of this rank, searches still out]

        def collect():
            nonlocal hit_total, last_hits
            step_no, start, res = in_flight.popleft()
            st = state[step_no]
            t = res.hits()  # waits for the kernels on the search's stream
            st[2] += len(t)
            t["structure"] += np.uint32(start - lo)
            st[0].append(t)
            if record:
                st[1].append(res.timing() + res.search_split())
            res.close()
            st[3] -= 1
            if st[3] == 0:  # the step is complete
                tables, times, n_local, _ = state.pop(step_no)
                if record:
                    index_ms.extend(t[0] for t in times)
                    search_ms.extend(t[1] for t in times)
                    finish_ms.extend(t[2] for t in times)
                    front_ms.extend(t[3] for t in times)
                    deep_ms.extend(t[4] for t in times)
                    hit_total = n_local
                    last_hits = np.concatenate(tables) if len(tables) > 1 else tables[0]

        def finish_pass(results, table):
            nonlocal hit_total, last_hits
            # each result is waited for by its OWN events (the counters' copy behind its last kernel, the append behind
            # that), not by its stream, on which the next pass's searches are already queued: they run during the gather.
            # A failure on one rank (an overflowed table) is exchanged before the collective, so that every rank raises
            # instead of one raising and the others waiting in the gather.
            failure = None
            for res in results:
                try:
                    res.wait()
                    res.wait_appended()
                except _native.NativeError as exc:
                    failure = failure or exc
            if dist.is_available() and dist.is_initialized():
                flag = torch.tensor([1 if failure is not None else 0], dtype=torch.int32, device="cpu" if one_gpu else f"cuda:{local_rank}")
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                if int(flag.item()) and failure is None:
                    failure = _native.NativeError(_native.TM_E_OVERFLOW, "a search of this pass failed on another rank")
            if failure is not None:
                raise failure
            g0 = time.perf_counter()
            gathered = table.gather(sort=False, force_collective=force_sharded)  # the pass's ONE exchange; rank 0 sorts after the timed region
            gather_ms.append(1e3 * (time.perf_counter() - g0))
            if record:
                for res in results:
                    a, b_, c = res.timing()
                    index_ms.append(a); search_ms.append(b_); finish_ms.append(c)
                    f_, d_ = res.search_split()
                    front_ms.append(f_); deep_ms.append(d_)
                hit_total = int(table.count.item())
                last_hits = gathered
            for res in results:
                res.close()

        launched = 0
        previous = None
        for step_no in range(n_steps):
            resident = resident_sets[step_no % len(resident_sets)]
            if not sharded:
                state[step_no] = [[], [], 0, len(resident)]
                for start, batch in resident:
                    in_flight.append((step_no, start, search(batch, streams[launched % len(streams)].handle)))
                    launched += 1
                    if len(in_flight) >= len(streams):
                        collect()
                continue
            table = pass_tables[step_no % 2]
            table.reset()
            torch.cuda.current_stream().synchronize()  # the reset ran on torch's stream, the appends come on ours
            results = []
            for start, batch in resident:
                res = search(batch, streams[launched % len(streams)].handle)
                table.append(res, start)  # global structure index = start + index in the batch
                results.append(res)
                launched += 1
            if previous is not None:
                finish_pass(*previous)
            previous = (results, table)
        while in_flight:
            collect()
        if previous is not None:
            finish_pass(*previous)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # clock pre-warm (untimed, before the contract's W warm-up steps): the timed region of the default command lasts a
    # tenth of a second, and a GPU that idled while the batches were generated and uploaded enters it at a low clock
    # (one collection of this round: 248 k structures/s in the timed region, 284 k in each of the three repeats after it)
    if args.prewarm_seconds > 0:
        p0 = time.perf_counter()
        while time.perf_counter() - p0 < args.prewarm_seconds:
            run_steps(2, False)
            if dist.is_available() and dist.is_initialized():
                more = torch.tensor([1 if time.perf_counter() - p0 < args.prewarm_seconds else 0], dtype=torch.int32,
                                    device="cpu" if one_gpu else "cuda")
                dist.all_reduce(more, op=dist.ReduceOp.MIN)  # (every rank runs the same number of passes)
                if not int(more.item()):
                    break
        fence()
    run_steps(args.warmup, False)
    fence()
    t0 = time.perf_counter()
    run_steps(args.steps, True)
    fence()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
    per_rank = None
    if world > 1 or force_sharded:
        # every rank's own clock over the timed region and its mean gather time, for reading the scaling curve: an
        # imbalance (one rank's host slower than the others') shows as one small per-rank value, not only in the MAX
        mine = torch.tensor([elapsed, float(np.mean(gather_ms)) if gather_ms else 0.0], dtype=torch.float64,
                            device="cpu" if one_gpu else "cuda")
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [[float(v[0]), float(v[1])] for v in every]
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    # ---- not part of `value`: the same K steps three more times -- the timed region of the contract is a fraction
    # of a second, these show how far one such measurement moves (config.repeats_structures_per_s)
    repeats = []
    for _ in range(3):  # record=False: the per-kernel timings and the kept hit table stay those of the timed region
        fence()
        r0 = time.perf_counter()
        run_steps(args.steps, False)
        fence()
        rt = torch.tensor([time.perf_counter() - r0], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        if world > 1:
            dist.all_reduce(rt, op=dist.ReduceOp.MAX)
        repeats.append(args.structures * world * args.steps / float(rt.item()))

    # ---- not part of `value`: the same loop for at least --sustained-seconds (the contract's timed region is a tenth
    # of a second: this is the figure that outlives a clock ramp or a cold cache)
    sustained = None
    if args.sustained_seconds > 0:
        fence()
        s0 = time.perf_counter()
        done = 0
        while True:
            run_steps(args.steps, False)
            done += args.steps
            stop = torch.tensor([1 if time.perf_counter() - s0 >= args.sustained_seconds else 0], dtype=torch.int32,
                                device="cpu" if one_gpu else "cuda")
            if dist.is_available() and dist.is_initialized():
                dist.all_reduce(stop, op=dist.ReduceOp.MAX)  # (every rank runs the same number of passes)
            if int(stop.item()):
                break
        fence()
        st = torch.tensor([time.perf_counter() - s0], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        if world > 1:
            dist.all_reduce(st, op=dist.ReduceOp.MAX)
        sustained = {"structures_per_s": args.structures * world * done / float(st.item()), "seconds": float(st.item()), "steps": done}
    free_b, total_b = torch.cuda.mem_get_info()
    hbm_in_use = int(total_b - free_b)  # (this rank's GPU, after the timed loops: resident batches, tables, result pools)

    # ---- not part of `value`: the kernels' durations with ONE search in flight (nothing else on the GPU), for
    # roofline.exclusive -- with two in flight the HIP events of a search also span what the other search's kernels ran
    exclusive_ms = None
    if len(streams) > 1:
        timed = (list(index_ms), list(search_ms), list(finish_ms), list(front_ms), list(deep_ms), hit_total, last_hits)
        saved, streams = streams, streams[:1]
        index_ms, search_ms, finish_ms, front_ms, deep_ms = [], [], [], [], []
        run_steps(3, True)
        fence()
        exclusive_ms = {"index": float(np.mean(index_ms)), "search": float(np.mean(search_ms)), "finish": float(np.mean(finish_ms)),
                        "front": float(np.mean(front_ms)), "deep": float(np.mean(deep_ms))}
        streams = saved
        index_ms, search_ms, finish_ms, front_ms, deep_ms, hit_total, last_hits = timed

    # ---- not part of `value`: the same pass with the structures coming from HOST memory (the molecules' atom tables):
    # batches i + 1, i + 2 are encoded on worker threads and handed over in the compact format (17 bytes per atom, laid
    # out on the device by k_flatten; the constructor returns with its copies queued) while batch i is searched and the
    # hit table of batch i - 1 is fetched -- the PCIe-inclusive rate, pipelined the way Matcher.run drives the library
    streamed, streamed_tables, h2d_bytes = None, None, None
    if args.streamed_passes > 0:
        import collections
        from enzymm_amd import ingest

        def stream_passes(n_passes, keep=None):
            """n_passes x this rank's structures as ONE stream of batches: encode + hand-over of batches i + 1, i + 2 on
            two worker threads, search of batch i, hit table of batch i - 1 -- all at once."""
            todo = [c for _ in range(n_passes) for c in chunks]
            made = ingest._prefetched(todo, lambda c: pyjess.make_batch(c, local_rank), 3, 2)
            in_flight = collections.deque()
            sent = 0

            def collect():
                b, res = in_flight.popleft()
                t = res.hits()
                if keep is not None and len(keep) < len(chunks):  # (the first pass's tables only: the parity sample reads them)
                    keep.append(t)
                res.close()
                b.close()
            for n, b in enumerate(made):
                sent += b.upload_bytes()
                in_flight.append((b, search(b, streams[n % len(streams)].handle)))
                if len(in_flight) >= max(2, len(streams)):
                    collect()
            while in_flight:
                collect()
            return sent / max(1, n_passes)

        stream_passes(2)
        fence()
        t0 = time.perf_counter()
        streamed_tables = []
        h2d_bytes = stream_passes(args.streamed_passes, streamed_tables)
        fence()
        dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device="cpu" if one_gpu else "cuda")
        if world > 1:
            dist.all_reduce(dt, op=dist.ReduceOp.MAX)
        streamed = args.structures * world * args.streamed_passes / float(dt.item())
        streamed_tables = streamed_tables[:len(chunks)]  # (the first pass's tables: the parity sample below reads them)

    # ---- not part of `value` (N = 1 only): the two rates VERDICT r02 asked to see beside the headline
    extras = {}
    if world == 1 and not args.no_extras:
        extras = untimed_extras(args, local_rank, templates, idx, cut, molecules, search, flags)

    merge_ms = None
    if rank == 0 and sharded and last_hits is not None:  # rank 0 orders the gathered rows off the timed path
        m0 = time.perf_counter()
        last_hits = distributed.sort_rows(last_hits)
        merge_ms = 1e3 * (time.perf_counter() - m0)

    if rank == 0:
        total_structs = args.structures * world
        ms_per_step = 1e3 * elapsed / args.steps
        # algorithmic HBM bytes of ONE search launch on this rank, SURVEY.md 8(d)'s per-unit figure x the units of
        # the launch: 16 B per target atom + 128 B per hit + the template library (98 808 atoms x 16 B + 659 760 pair
        # distances x 4 B = 4.2 MB) read once per batch -- about 44 KB per 2 350-atom structure in a batch of 1000
        lib_bytes = 98808 * 16 + 659760 * 4
        algo_bytes = (n_atoms * 16 + hit_total * 128) / len(batches) + lib_bytes  # per launch = per resident batch
        k_ms = float(np.mean(search_ms))  # HIP events on the search's own stream over the timed region (two searches in flight)
        # roofline.achieved / frac are priced on the dominant kernel's duration ALONE on the GPU (one search in flight,
        # measured live after the timed region: what rocprofv3 --kernel-trace --stats of `--pipeline 1` shows); with two
        # searches in flight the events also span the other search's kernels -- that figure is roofline.overlapped_frac
        dom_ms = exclusive_ms["search"] if exclusive_ms is not None else k_ms
        achieved = algo_bytes / (dom_ms * 1e-3) / 1e9
        # HBM traffic of one step's three hot kernels (the block search k_search, the depth-first part k_deep, the pair-table
        # build k_tab_build) from the committed PMC passes of this workload (rocprofv3 cannot run inside the timed
        # process); null when no summary is present.  profiles/LATEST names the summaries of the CURRENT kernels
        # (profiles/collect.sh).
        traffic, traffic_src, secondary, traffic_parts = None, None, None, None
        latest = ROOT / "profiles" / "LATEST"
        if latest.exists() and args.structures == 1000 and args.generator == "v2":
            try:
                pointer = json.loads(latest.read_text())
                traffic, secondary, names, traffic_parts = 0.0, {}, [], {}
                for key in ("k_search_pmc", "k_deep_pmc", "k_tab_build_pmc"):
                    pmc = json.loads((ROOT / "profiles" / pointer[key]).read_text())
                    traffic_parts[key[:-4]] = pmc["hbm_traffic_bytes_per_launch"]["total_corrected"]
                    traffic += traffic_parts[key[:-4]]
                    secondary[key[:-4]] = pmc.get("secondary")
                    names.append(f"profiles/{pointer[key]}")
                traffic_src = " + ".join(names) + " (FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
            except (KeyError, ValueError, OSError):
                traffic, traffic_src, secondary, traffic_parts = None, None, None, None
        valu_issue = None
        if secondary and secondary.get("k_search"):
            valu_issue = secondary["k_search"].get("valu_issue_frac")
        out = {
            "metric": "structures_per_sec_vs_full_mcsa_library",
            "value": total_structs / (elapsed / args.steps),
            "unit": "structures/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64 (f32 prefilter)", "data": "synthetic",
            "config": {
                "workload": (f"{args.structures} synthetic ~300-residue structures per GPU in resident batches of "
                             f"{args.batch} vs the 6780 default-eligible templates of jess_templates_20230210 "
                             + ("(BASELINE.json configs[2])" if args.structures == 1000 else
                                "(BASELINE.json configs[3] when 8 x 12500)")),
                "generator": args.generator,
                "structures_per_gpu": args.structures, "batch": args.batch, "templates": int(len(idx)),
                "repeats_structures_per_s": repeats,  # the same K steps timed three more times (not `value`)
                "sustained_structures_per_s": None if sustained is None else sustained["structures_per_s"],
                "sustained": sustained,  # the same loop for >= --sustained-seconds (not `value`)
                "hbm_bytes_in_use_rank0": hbm_in_use,
                "streamed_structures_per_s": streamed,  # upload-inclusive (host atom tables -> HBM overlapped with the search)
                # host-to-device bytes per structure of those passes (packed word + chain ordinal + int32 thousandths of
                # an Angstrom per atom, 32 B per structure; 46 B per atom before round 5)
                "h2d_bytes_per_structure": None if h2d_bytes is None else h2d_bytes / max(1, len(molecules)),
                # how long that stream was: passes x resident batches, one after the other (a stream pays its fill and its
                # drain once: 16 batches 215-245 k/s, 208 batches 321-344 k/s on the boxes of round 5)
                "streamed_batches": args.streamed_passes * len(chunks),
                "atoms_per_structure": n_atoms / max(1, len(molecules)),
                "parallelism": f"structures sharded over {world} GPU(s), hit-table gather to rank 0" + (" (N > 1 loop forced on one GPU over RCCL)" if force_sharded else ""),
                "hits_rank0": int(hit_total), "hits_total": int(len(last_hits)),
                # fingerprint of (structure, template, atoms) of the whole job's hit table: a sharded run and the
                # unsharded run of the same structures print the same value (tests/test_gpu_round3.py)
                "hit_table_digest": distributed.table_digest(last_hits),
                # N > 1: host milliseconds of the one gather per pass (it runs while the next pass's searches do) and
                # of rank 0's sort of the gathered rows (after the timed region)
                "gather_ms_per_pass": float(np.mean(gather_ms)) if gather_ms else None, "merge_ms": merge_ms,
                # N > 1: every rank's own rate over the timed region (its structures / its clock; `value` divides by the
                # slowest), the gather's host time per pass across the ranks, and which rank was slowest
                "per_rank_structures_per_s": None if per_rank is None else [args.structures * args.steps / e for e, _ in per_rank],
                "gather_ms_per_pass_max": None if per_rank is None else max(g for _, g in per_rank),
                "gather_ms_per_pass_mean": None if per_rank is None else float(np.mean([g for _, g in per_rank])),
                "slowest_rank": None if per_rank is None else int(np.argmax([e for e, _ in per_rank])),
                "host_threads_per_rank": _native.host_threads(),
                "index_ms": float(np.mean(index_ms)), "search_ms": k_ms,
                "finish_ms": float(np.mean(finish_ms)),
                # searches in flight per rank; with 2, index_ms/search_ms/finish_ms (HIP events on each search's own
                # stream) include whatever the neighbouring search's kernels took from the GPU meanwhile
                "pipeline": len(streams),
                **extras,
            },
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         # what `frac` divides by changed in round 5: the dominant kernel's duration
                         # ALONE on the GPU, as rocprofv3 --stats of `--pipeline 1` reports it; BENCH_r01..r04 printed what
                         # is now `overlapped_frac`.  `value` is still the rate with two searches in flight.
                         "frac_definition": "algorithmic bytes / exclusive kernel_ms (one search in flight) / peak; "
                                            "r01-r04 lines used the overlapped duration = overlapped_frac",
                         # the same with the event durations of the timed region (two searches in flight)
                         "overlapped_frac": algo_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         # measured HBM bytes of a step's three hot kernels, and how many times the algorithmic bytes
                         # that is (the path is not HBM-bound: the tables are built, written and read per structure)
                         "traffic": traffic, "traffic_parts": traffic_parts,
                         "traffic_ratio": None if traffic is None else traffic / algo_bytes,
                         "traffic_source": traffic_src,
                         # HBM is not what binds the dominant kernel: its VALU issue rate as a fraction of the MEASURED
                         # ceiling (profiles/valu_probe.hip), from the same PMC set
                         "valu_issue_frac": valu_issue,
                         # one search = two launches back to back on the search's stream: the block search over the pair tables
                         # (k_search<mode 0>) and the depth-first part below the surviving triples (k_deep = k_search<mode 2>)
                         "kernel": "k_search + k_deep",
                         "algorithmic_bytes_per_launch": algo_bytes,
                         # duration the fraction is priced on: the pair alone on the GPU (one search in flight) ...
                         "kernel_ms": dom_ms,
                         # ... and its average over the timed region (two in flight: includes the neighbour's kernels), with its parts
                         "kernel_ms_timed_region": k_ms,
                         "kernel_ms_parts": {"k_search": float(np.mean(front_ms)), "k_deep": float(np.mean(deep_ms))},
                         # the same kernels with one search in flight, measured after the timed region
                         "exclusive": None if exclusive_ms is None else {
                             "kernel_ms": exclusive_ms["search"], "k_search_ms": exclusive_ms["front"], "k_deep_ms": exclusive_ms["deep"],
                             "index_ms": exclusive_ms["index"],
                             "finish_ms": exclusive_ms["finish"],
                             "frac": algo_bytes / (exclusive_ms["search"] * 1e-3) / 1e9 / HBM_PEAK_GBS},
                         # what actually bounds the kernel (instruction issue of a divergent search), from the same
                         # committed PMC passes: instructions per (structure, template) search, VALU issue share
                         "secondary": secondary},
        }
        if world == 1 and not args.no_cpu_baseline:
            base, oracle_results = cpu_baseline(molecules, idx, cut, args.cpu_baseline_seconds)
            ref = reference_baseline(molecules, idx, cut, args.cpu_baseline_seconds)  # None: PyJess not on this host
            out["cpu_baseline"] = ref if ref is not None else base
            if ref is not None:
                out["cpu_baseline_port"] = base
            # the sample doubles as a parity check of the timed GPU output
            mism = 0
            by_structure = {}
            for h in last_hits:
                by_structure.setdefault(int(h["structure"]), {})[int(h["template_sel"])] = h
            for s, r in enumerate(oracle_results):
                mine = by_structure.get(s, {})
                theirs = set(np.nonzero(r["hit"])[0].tolist())
                mism += len(theirs ^ set(mine))
                for ti in theirs & set(mine):
                    k = int(mine[ti]["n_atoms"])
                    if not np.array_equal(mine[ti]["atoms"][:k], r["atoms"][ti][:k]) or \
                            abs(mine[ti]["rmsd"] - r["rmsd"][ti]) > 1e-5:
                        mism += 1
            out["parity_sample"] = {"structures": len(oracle_results), "mismatches": mism}
            # ... and of the STREAMED output (the last upload-inclusive pass: its first batch holds structures 0 ..)
            if streamed_tables:
                smism = 0
                first = streamed_tables[0]
                by_structure = {}
                for h in first:
                    by_structure.setdefault(int(h["structure"]), {})[int(h["template_sel"])] = h
                checked = min(len(oracle_results), len(chunks[0]))
                for s_ in range(checked):
                    r = oracle_results[s_]
                    mine = by_structure.get(s_, {})
                    theirs = set(np.nonzero(r["hit"])[0].tolist())
                    smism += len(theirs ^ set(mine))
                    for ti in theirs & set(mine):
                        k = int(mine[ti]["n_atoms"])
                        if not np.array_equal(mine[ti]["atoms"][:k], r["atoms"][ti][:k]) or \
                                abs(mine[ti]["rmsd"] - r["rmsd"][ti]) > 1e-5:
                            smism += 1
                out["parity_sample_streamed"] = {"structures": checked, "mismatches": smism}
        # (RCCL keeps a version banner in the C library's stdout buffer until the process ends: out with it first, so that
        # the JSON line is the last thing this process prints)
        sys.stdout.flush()
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except OSError:
            pass
        print(json.dumps(out), flush=True)
    if world > 1 or force_sharded:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
