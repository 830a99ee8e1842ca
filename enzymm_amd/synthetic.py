"""Seeded synthetic query structures for benchmarks and parity tests.

The reference ships no benchmark inputs, so the workload "batch of synthetic
~300-residue structures" (BASELINE.json configs[2], [3]) is defined here; structure ``i`` uses seed
``1234 + i`` and L ~ N(300, 30^2) residues clipped to [150, 450].

Two generators:

* **v2 (default)**, ``_synth_core.py``: natively packed globules recombined from wedge-shaped cutouts of the
  two structures the reference's own tests use; atom density and radius of gyration are those of globular
  proteins (about 140 atoms within 10 A of an atom, Rg within a few per cent of 2.2 L^0.38 A; the band is
  asserted in tests/test_host_model.py).  ``complex_structure`` builds multi-chain complexes of such globules
  (BASELINE.json configs[4]).
* **v1** (``generator="v1"``, round 1): backbone-contiguous fragments (8-20 residues) of the same sources placed
  as rigid bodies by a greedy compact growth that only rejects C-alpha clashes.  Real local geometry, but 35 %
  fewer atom pairs inside the templates' distance windows than a real protein (about 105 atoms within 10 A,
  Rg 20 % above 2.2 L^0.38); kept so that rates measured on both can be put side by side.

``rigid_copies`` gives the density reference asked for next to them: the real structures themselves under random
rigid transforms.
"""
from __future__ import annotations

import functools
import os
from pathlib import Path
from typing import List

import numpy as np

from . import _synth_core, pyjess

_POOL = Path(__file__).resolve().parent / "data" / "residue_pool.npz"
BASE_SEED = 1234
TRIALS = 48  # poses tried per fragment


@functools.lru_cache(maxsize=1)
def _pool():
    z = np.load(_POOL)
    xyz, name, res, resid, source = z["xyz"], z["name"], z["res"], z["resid"], z["source"]
    # global residue index over both sources, atom ranges per residue, C-alpha per residue
    gid = resid.astype(np.int64) + source.astype(np.int64) * (int(resid.max()) + 1)
    uniq, first, counts = np.unique(gid, return_index=True, return_counts=True)
    order = np.argsort(first)
    first, counts, uniq = first[order], counts[order], uniq[order]
    ca = np.full(len(first), -1, dtype=np.int64)
    for r, (f, c) in enumerate(zip(first, counts)):
        hit = np.nonzero(name[f:f + c] == b" CA ")[0]
        if len(hit):
            ca[r] = f + hit[0]
    res_source = source[first]
    return dict(xyz=xyz, name=name, res=res, first=first, counts=counts, ca=ca, res_source=res_source)


def _random_rotations(rng, count: int) -> np.ndarray:
    q = rng.normal(size=(count, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((count, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def structure_columns_v1(index: int, mean_residues: float = 300.0, sd_residues: float = 30.0,
                         min_residues: int = 150, max_residues: int = 450):
    """Generator v1: columns (xyz, name, residue_name, residue_number, temperature_factor) of structure ``index``."""
    P = _pool()
    rng = np.random.default_rng(BASE_SEED + int(index))
    length = int(np.clip(round(rng.normal(mean_residues, sd_residues)), min_residues, max_residues))
    n_pool = len(P["first"])
    xyz_parts, name_parts, res_parts, resnum_parts = [], [], [], []
    ca_all = np.zeros((0, 3))
    last_ca = None
    built = 0
    while built < length:
        flen = int(min(rng.integers(8, 21), length - built))
        # a fragment must stay inside one source and have all its C-alphas
        for _ in range(100):
            start = int(rng.integers(0, n_pool - flen))
            rs = np.arange(start, start + flen)
            if P["res_source"][rs[0]] == P["res_source"][rs[-1]] and (P["ca"][rs] >= 0).all():
                break
        a0, a1 = P["first"][rs[0]], P["first"][rs[-1]] + P["counts"][rs[-1]]
        frag = P["xyz"][a0:a1]
        frag_ca = P["xyz"][P["ca"][rs]]
        # all trial poses at once: rotated C-alphas [T, flen, 3]
        Rs = _random_rotations(rng, TRIALS)
        ca_rot = np.einsum("tij,fj->tfi", Rs, frag_ca)
        if last_ca is None:
            shifts = -ca_rot[:, 0, :]
        else:
            d = rng.normal(size=(TRIALS, 3))
            d *= 3.8 / np.linalg.norm(d, axis=1, keepdims=True)
            shifts = last_ca + d - ca_rot[:, 0, :]
        ca_new_all = ca_rot + shifts[:, None, :]
        if len(ca_all):
            d2 = np.zeros(ca_new_all.shape[:2] + (len(ca_all),))
            for axis in range(3):  # elementwise only: bit-reproducible on any host
                diff = ca_new_all[:, :, None, axis] - ca_all[None, None, :, axis]
                diff *= diff
                d2 += diff
            clashes = (d2 < 16.0).sum(axis=(1, 2))
            spread = np.linalg.norm(ca_new_all - ca_all.mean(0), axis=2).mean(axis=1)
            pick = int(np.argmin(1000.0 * clashes + spread))
        else:
            pick = 0
        R, shift, ca_new = Rs[pick], shifts[pick], ca_new_all[pick]
        xyz_parts.append(frag @ R.T + shift)
        name_parts.append(P["name"][a0:a1])
        res_parts.append(P["res"][a0:a1])
        local = np.repeat(np.arange(flen), P["counts"][rs])
        resnum_parts.append(built + 1 + local)
        ca_all = np.vstack([ca_all, ca_new])
        last_ca = ca_new[-1]
        built += flen
    xyz = np.round(np.concatenate(xyz_parts), 3)
    name = np.concatenate(name_parts)
    res = np.concatenate(res_parts)
    resnum = np.concatenate(resnum_parts).astype(np.int32)
    # one value per residue, like the pLDDT column of a predicted model
    bfac = np.round(rng.uniform(30.0, 100.0, size=length), 2)[resnum - 1]
    return dict(xyz=xyz, name=name, residue_name=res, residue_number=resnum, temperature_factor=bfac)


def _molecule(ident: str, c) -> pyjess.Molecule:
    n = len(c["residue_number"])
    element = np.array([nm[1:2] for nm in c["name"]], dtype="S2")
    chain = c.get("chain_id")
    return pyjess.Molecule.from_columns(
        id=ident, xyz=c["xyz"], name=c["name"], residue_name=c["residue_name"],
        chain_id=chain if chain is not None else np.full(n, b" A", dtype="S2"), residue_number=c["residue_number"],
        temperature_factor=c["temperature_factor"], occupancy=np.ones(n), element=element)


def structure_columns(index: int, generator: str = "v2", **kw):
    if generator == "v1":
        return structure_columns_v1(index, **kw)
    if generator != "v2":
        raise ValueError(f"unknown generator {generator!r}")
    return _synth_core.structure_columns(index, **kw)


def structure(index: int, generator: str = "v2", **kw) -> pyjess.Molecule:
    return _molecule(f"synthetic_{index:06d}", structure_columns(index, generator=generator, **kw))


def _cache_file(kind: str, count: int, start: int, kw) -> "Path | None":
    """ENZYMM_AMD_SYNTH_CACHE=<dir>: generated columns are kept there (profiling sessions run the same bench
    several times on one box); unset = nothing is cached."""
    root = os.environ.get("ENZYMM_AMD_SYNTH_CACHE")
    if not root:
        return None
    tag = "_".join(f"{k}{v}" for k, v in sorted(kw.items()))
    return Path(root) / f"synth_{kind}_{start}_{count}_{tag}.npz"


def structures(count: int, start: int = 0, processes: int = 0, generator: str = "v2", **kw) -> List[pyjess.Molecule]:
    """Structures ``start`` .. ``start + count - 1``.  ``processes`` > 1 generates them on forked worker
    processes (numpy only; call it before this process initialises the GPU)."""
    if generator == "v1":
        return [structure(start + i, generator="v1", **kw) for i in range(count)]
    cache = _cache_file("v2", count, start, kw)
    cols = None
    if cache is not None and cache.exists():
        z = np.load(cache)
        off = z["offsets"]
        # (every column read ONCE: an NpzFile decodes the member again at every z[k] -- round 6 found the cached path of
        # 1000 structures at 69 s, slower than generating them)
        arrays = {k: z[k] for k in _CACHE_KEYS}
        cols = [{k: arrays[k][off[i]:off[i + 1]] for k in _CACHE_KEYS} for i in range(count)]
    if cols is None:
        jobs = [("globule", start + i, kw) for i in range(count)]
        if processes > 1 and count >= 4 * processes:
            import multiprocessing
            with multiprocessing.get_context("fork").Pool(processes) as pool:
                cols = pool.map(_synth_core._columns_job, jobs, chunksize=max(1, count // (8 * processes)))
        else:
            cols = [_synth_core._columns_job(j) for j in jobs]
        if cache is not None:
            cache.parent.mkdir(parents=True, exist_ok=True)
            off = np.concatenate([[0], np.cumsum([len(c["residue_number"]) for c in cols])])
            tmp = cache.with_suffix(f".{os.getpid()}.tmp.npz")
            np.savez(tmp, offsets=off, **{k: np.concatenate([c[k] for c in cols]) for k in _CACHE_KEYS})
            os.replace(tmp, cache)
    return [_molecule(f"synthetic_{start + i:06d}", c) for i, c in enumerate(cols)]


_CACHE_KEYS = ("xyz", "name", "residue_name", "residue_number", "temperature_factor", "chain_id")


def complex_structure(index: int, n_residues: int = 6000, **kw) -> pyjess.Molecule:
    """A multi-chain complex of about ``n_residues`` residues (BASELINE.json configs[4])."""
    return _molecule(f"complex_{index:06d}", _synth_core.complex_columns(index, n_residues=n_residues, **kw))


def rigid_copies(mol: pyjess.Molecule, count: int, seed: int = 0) -> List[pyjess.Molecule]:
    """``count`` copies of a real structure under random rigid transforms, coordinates rounded to PDB precision:
    the density reference next to the synthetic batches."""
    rng = np.random.default_rng(BASE_SEED + 9_000_000 + seed)
    Rs = _random_rotations(rng, count)
    shifts = rng.uniform(-50.0, 50.0, size=(count, 3))
    centre = mol._xyz.mean(0)
    out = []
    for i in range(count):
        c = {k: v for k, v in mol._cols.items()}
        xyz = np.round((mol._xyz - centre) @ Rs[i].T + shifts[i], 3)
        m = pyjess.Molecule.from_columns(id=f"{mol.id}_copy{i:04d}", xyz=xyz,
                                         **{k: v for k, v in c.items() if k not in ("xyz",)})
        out.append(m)
    return out


def to_pdb(mol: pyjess.Molecule) -> str:
    """PDB text of a molecule (used to hand identical input to the oracle's own parser)."""
    out = []
    c = mol._cols
    for i in range(len(mol)):
        out.append("ATOM  %5d %4s%1s%3s%2s%4d%1s   %8.3f%8.3f%8.3f%6.2f%6.2f      %-4s%2s  " % (
            c["serial"][i] % 100000, bytes(c["name"][i]).decode().ljust(4), bytes(c["altloc"][i]).decode() or " ",
            bytes(c["residue_name"][i]).decode().rjust(3), bytes(c["chain_id"][i]).decode().rjust(2),
            c["residue_number"][i], bytes(c["insertion_code"][i]).decode() or " ",
            mol._xyz[i, 0], mol._xyz[i, 1], mol._xyz[i, 2], c["occupancy"][i], c["temperature_factor"][i],
            bytes(c["segment"][i]).decode(), bytes(c["element"][i]).decode().rjust(2)))
    out.append("END")
    return "\n".join(out) + "\n"
