"""ctypes binding of oracle/libjess_oracle.so (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from .parse import MAXRESN, OracleMolecule, OracleTemplate

_HERE = Path(__file__).resolve().parent
_LIB = None


def build(force: bool = False) -> Path:
    so = _HERE / "libjess_oracle.so"
    src = _HERE / "jess_oracle.c"
    if force or not so.exists() or so.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(_HERE), "-B", "libjess_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(str(build()))
        _LIB.oracle_query.restype = C.c_int
        _LIB.oracle_query_library.restype = C.c_int
        _LIB.oracle_residue_ids.restype = C.c_int
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def residue_ids(mol: OracleMolecule) -> np.ndarray:
    out = np.zeros(len(mol), dtype=np.int32)
    rc = _lib().oracle_residue_ids(C.c_int(len(mol)), _p(mol.chain), _p(mol.resseq), _p(mol.icode), _p(out))
    if rc < 0:
        raise RuntimeError(f"oracle_residue_ids failed: {rc}")
    return out


class OracleHit:
    __slots__ = ("atoms", "rmsd", "rot", "trans", "candidates", "truncated")

    def __init__(self, atoms, rmsd, rot, trans, candidates, truncated):
        self.atoms, self.rmsd, self.rot, self.trans = atoms, rmsd, rot, trans
        self.candidates, self.truncated = candidates, truncated


def query(mol: OracleMolecule, tpl: OracleTemplate, rmsd_threshold, distance_cutoff, max_dynamic_distance,
          max_candidates=1000, best_match=False, ignore_chain=False, max_hits=100000):
    """Hits of ONE template in enumeration order -> (list[OracleHit], candidates, truncated)."""
    k = tpl.k
    if best_match:
        max_hits = 1
    resid = residue_ids(mol)
    atoms = np.zeros((max_hits, k), dtype=np.int32)
    rmsd = np.zeros(max_hits, dtype=np.float64)
    rot = np.zeros((max_hits, 9), dtype=np.float64)
    trans = np.zeros((max_hits, 3), dtype=np.float64)
    ncand, trunc = C.c_int(0), C.c_int(0)
    rc = _lib().oracle_query(
        C.c_int(len(mol)), _p(mol.xyz), _p(mol.name), _p(mol.res), _p(mol.chain), _p(resid),
        C.c_int(k), _p(tpl.xyz), _p(tpl.mode), _p(tpl.name), _p(tpl.resnames), _p(tpl.nres), C.c_int(MAXRESN),
        _p(tpl.chain), _p(tpl.resseq), _p(tpl.weight),
        C.c_double(rmsd_threshold), C.c_double(distance_cutoff), C.c_double(max_dynamic_distance),
        C.c_int(max_candidates), C.c_int(bool(best_match)), C.c_int(bool(ignore_chain)),
        C.c_int(max_hits), _p(atoms), _p(rmsd), _p(rot), _p(trans), C.byref(ncand), C.byref(trunc))
    if rc < 0:
        raise RuntimeError(f"oracle_query failed: {rc}")
    hits = [OracleHit(atoms[i].copy(), float(rmsd[i]), rot[i].reshape(3, 3).copy(), trans[i].copy(),
                      ncand.value, bool(trunc.value)) for i in range(rc)]
    return hits, ncand.value, bool(trunc.value)


class OracleLibrary:
    """Concatenated template tables for oracle_query_library."""

    def __init__(self, templates):
        self.templates = list(templates)
        ks = [t.k for t in self.templates]
        self.toff = np.zeros(len(ks) + 1, dtype=np.int32)
        self.toff[1:] = np.cumsum(ks)
        self.maxk = max(ks) if ks else 1
        cat = lambda name: np.ascontiguousarray(np.concatenate([getattr(t, name) for t in self.templates]))
        self.xyz, self.mode, self.name = cat("xyz"), cat("mode"), cat("name")
        self.resnames, self.nres, self.chain = cat("resnames"), cat("nres"), cat("chain")
        self.resseq, self.weight = cat("resseq"), cat("weight")


def query_library(mol: OracleMolecule, lib: OracleLibrary, rmsd_threshold, distance_cutoff, max_dynamic_distance,
                  max_candidates=10000, ignore_chain=True, threads=None):
    """best_match query of every template in lib against mol.

    The three thresholds may be scalars or per-template arrays.  Returns dict with
    per-template arrays: hit (0/1), atoms [nt, maxk], rmsd, rot [nt,3,3], trans, candidates, truncated.
    """
    nt = len(lib.templates)
    par = [np.ascontiguousarray(np.broadcast_to(np.asarray(v, dtype=np.float64), (nt,)))
           for v in (rmsd_threshold, distance_cutoff, max_dynamic_distance)]
    resid = residue_ids(mol)
    hit = np.zeros(nt, dtype=np.int32)
    atoms = np.full((nt, lib.maxk), -1, dtype=np.int32)
    rmsd = np.full(nt, np.nan)
    rot = np.zeros((nt, 9))
    trans = np.zeros((nt, 3))
    ncand = np.zeros(nt, dtype=np.int32)
    trunc = np.zeros(nt, dtype=np.int32)
    old = os.environ.get("OMP_NUM_THREADS")
    if threads is not None:
        os.environ["OMP_NUM_THREADS"] = str(threads)
        try:
            C.CDLL("libgomp.so.1").omp_set_num_threads(C.c_int(threads))
        except OSError:
            pass
    rc = _lib().oracle_query_library(
        C.c_int(len(mol)), _p(mol.xyz), _p(mol.name), _p(mol.res), _p(mol.chain), _p(resid),
        C.c_int(nt), _p(lib.toff), _p(lib.xyz), _p(lib.mode), _p(lib.name), _p(lib.resnames), _p(lib.nres),
        C.c_int(MAXRESN), _p(lib.chain), _p(lib.resseq), _p(lib.weight),
        _p(par[0]), _p(par[1]), _p(par[2]), C.c_int(max_candidates), C.c_int(bool(ignore_chain)),
        C.c_int(lib.maxk), _p(hit), _p(atoms), _p(rmsd), _p(rot), _p(trans), _p(ncand), _p(trunc))
    if old is None:
        os.environ.pop("OMP_NUM_THREADS", None)
    else:
        os.environ["OMP_NUM_THREADS"] = old
    if rc < 0:
        raise RuntimeError(f"oracle_query_library failed: {rc}")
    return dict(hit=hit, atoms=atoms, rmsd=rmsd, rot=rot.reshape(nt, 3, 3), trans=trans, candidates=ncand,
                truncated=trunc)
