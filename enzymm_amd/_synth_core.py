"""Numpy-only core of the synthetic-structure generator (see ``synthetic.py`` for the workload definition).

Kept free of anything that touches the GPU or the native library so that worker processes can import it.

Generator v2 ("wedge recombination").  The two structures the reference's own tests use (1AMY, AF-P0DUB6; their
protein residues are packed in ``data/residue_pool.npz``) are the only natively packed material available
offline, so density is inherited from them instead of being approximated:

* each source is centred on the median of its residue centroids; structure ``i`` (seed ``1234 + i``) draws
  K in {2, 3} random directions that cut space into K wedges meeting at the origin, and fills wedge k with the
  residues of a randomly chosen, randomly rotated source whose centroid lies in that wedge -- every wedge is a
  native cutout, only the interfaces between wedges are new;
* whole residues that come within 2.8 A (any heavy atom) of a residue of another wedge are in conflict; the
  conflict graph is thinned to an independent set by the min-degree greedy rule (so as many residues as possible
  survive), with residues up to 6 A beyond their wedge boundary offered as void fillers;
* the innermost L residues under a random ellipsoidal norm (axis ratios exp(U(-0.25, 0.25)), volume preserving)
  are kept, L ~ N(300, 30^2) clipped to [150, 450]: a compact globule with radius of gyration within a few per
  cent of 2.2 L^0.38 A;
* residues are renumbered 1..L on chain A (wedge after wedge, source order inside a wedge), coordinates stay at
  PDB precision, B-factors are pLDDT-like per residue: confident core, less confident surface (``plddt_like``).

What it is not: a single connected chain -- a wedge holds the chain segments of its source that pass through it.
Nothing on the matching path looks at connectivity.

Measured on structures 0..199 (tests/test_host_model.py pins the band): 140 +- 6 atoms within 10 A of an atom
(a native 270-residue cutout of the same sources: 155; the full 1AMY / AF-P0DUB6: 164 / 161; generator v1,
rigid 8-20 residue fragments grown greedily: 105), 36 within 6 A (native 40-42).

Large complexes (BASELINE.json configs[4]): ``complex_columns`` places such globules, one chain each, on a
jittered close-packed lattice and resolves inter-chain conflicts the same way.
"""
from __future__ import annotations

import functools
from pathlib import Path

import numpy as np
from scipy.spatial import cKDTree

_POOL = Path(__file__).resolve().parent / "data" / "residue_pool.npz"
BASE_SEED = 1234
CLASH = 2.8    # A: closest approach of heavy atoms of residues from different wedges / chains
MARGIN = 6.0   # A: how far beyond its wedge boundary a residue may sit and still be offered as a void filler
ANISO = 0.25   # log axis ratio range of the ellipsoidal cut


@functools.lru_cache(maxsize=1)
def sources():
    """Per source structure: centred atoms, residue table, atom -> residue map."""
    z = np.load(_POOL)
    out = []
    for s in np.unique(z["source"]):
        m = z["source"] == s
        x, rid = z["xyz"][m], z["resid"][m]
        _, first, cnt = np.unique(rid, return_index=True, return_counts=True)
        order = np.argsort(first)
        first, cnt = first[order], cnt[order]
        rcen = np.add.reduceat(x, first, axis=0) / cnt[:, None]
        c = np.median(rcen, axis=0)
        out.append(dict(x=x - c, name=z["name"][m], res=z["res"][m], first=first, cnt=cnt, rcen=rcen - c,
                        rof=np.repeat(np.arange(len(first)), cnt)))
    return out


def random_rotations(rng, count: int) -> np.ndarray:
    q = rng.normal(size=(count, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = np.empty((count, 3, 3))
    R[:, 0, 0] = 1 - 2 * (y * y + z * z); R[:, 0, 1] = 2 * (x * y - w * z); R[:, 0, 2] = 2 * (x * z + w * y)
    R[:, 1, 0] = 2 * (x * y + w * z); R[:, 1, 1] = 1 - 2 * (x * x + z * z); R[:, 1, 2] = 2 * (y * z - w * x)
    R[:, 2, 0] = 2 * (x * z - w * y); R[:, 2, 1] = 2 * (y * z + w * x); R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def _independent_set(n_res, res_of_atom, group_of_res, xyz, alive, tie):
    """Residues (of ``alive``) kept so that no two residues of different groups have atoms within CLASH:
    greedy minimum-degree independent set of the conflict graph; ``tie`` breaks equal degrees (smaller first)."""
    am = alive[res_of_atom]
    ra = res_of_atom[am]
    pairs = cKDTree(xyz[am]).query_pairs(CLASH, output_type="ndarray")
    a, b = ra[pairs[:, 0]], ra[pairs[:, 1]]
    cross = group_of_res[a] != group_of_res[b]
    a, b = a[cross], b[cross]
    present = alive.copy()
    if not len(a):
        return present
    nodes, inv = np.unique(np.concatenate([a, b]), return_inverse=True)
    n = len(nodes)
    ia, ib = inv[: len(a)], inv[len(a):]
    A = np.zeros((n, n), dtype=bool)
    A[ia, ib] = True
    A[ib, ia] = True
    present[nodes] = False
    live = np.ones(n, dtype=bool)
    deg = A.sum(1).astype(np.float64)
    bias = tie[nodes]
    while live.any():
        i = int(np.argmin(np.where(live, deg + bias, np.inf)))
        present[nodes[i]] = True
        gone = A[i] & live
        gone[i] = True
        live &= ~gone
        deg -= A[:, gone].sum(1)
    return present


def globule(rng, length: int):
    """One compact globule of ``length`` residues (or as many as the material gives): per-atom arrays
    (xyz, name, res, residue ordinal 0..n-1) centred near the origin."""
    S = sources()
    K = 2 + int(rng.integers(2))
    dirs = rng.normal(size=(K, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    Rs = random_rotations(rng, K)
    ax = np.exp(rng.uniform(-ANISO, ANISO, size=3))
    ax /= ax.prod() ** (1.0 / 3.0)
    xs, names, ress, rids, wedge, rad, outside = [], [], [], [], [], [], []
    base = 0
    for k in range(K):
        src = S[int(rng.integers(len(S)))]
        rc = src["rcen"] @ Rs[k].T
        proj = rc @ dirs.T
        other = np.max(np.delete(proj, k, axis=1), axis=1)
        out_k = (other - proj[:, k]) / np.sqrt(2.0)  # about the distance beyond the wedge boundary
        use = out_k < MARGIN
        atom_use = use[src["rof"]]
        remap = np.cumsum(use) - 1
        xs.append((src["x"] @ Rs[k].T)[atom_use])
        names.append(src["name"][atom_use])
        ress.append(src["res"][atom_use])
        rids.append(remap[src["rof"][atom_use]] + base)
        n_k = int(use.sum())
        base += n_k
        wedge.append(np.full(n_k, k))
        rad.append(np.linalg.norm(rc[use] / ax, axis=1))
        outside.append(out_k[use])
    X = np.vstack(xs)
    name, res = np.concatenate(names), np.concatenate(ress)
    RID = np.concatenate(rids)
    W, rad, outside = np.concatenate(wedge), np.concatenate(rad), np.concatenate(outside)
    inside = outside <= 0
    # a generous radial pre-cut keeps the conflict graph small
    r_in = np.sort(rad[inside])
    alive = rad <= r_in[min(length + 40, len(r_in) - 1)]
    present = _independent_set(base, RID, W, X, alive, tie=outside * 0.01 + rad * 1e-4)
    keep = np.argsort(np.where(present, rad, np.inf), kind="stable")[:length]
    sel = np.zeros(base, dtype=bool)
    sel[keep] = True
    sel &= present
    am = sel[RID]
    new_ord = np.cumsum(sel) - 1  # wedge after wedge, source order inside a wedge
    return X[am], name[am], res[am], new_ord[RID[am]], rad[sel] / max(float(rad[sel].max()), 1e-9)


def plddt_like(rng, rel_radius):
    """Per-residue temperature factors shaped like the pLDDT column of a predicted model: confident core, less
    confident surface (100 - 55 r^2 + N(0, 6), clipped to [30, 100]; about 40 % of the residues reach 70)."""
    return np.round(np.clip(100.0 - 55.0 * rel_radius ** 2 + rng.normal(scale=6.0, size=len(rel_radius)), 30.0, 100.0), 2)


def draw_length(rng, mean_residues, sd_residues, min_residues, max_residues) -> int:
    return int(np.clip(round(rng.normal(mean_residues, sd_residues)), min_residues, max_residues))


def structure_columns(index: int, mean_residues: float = 300.0, sd_residues: float = 30.0,
                      min_residues: int = 150, max_residues: int = 450):
    """Columns of synthetic structure ``index``: one globule on chain A."""
    rng = np.random.default_rng(BASE_SEED + int(index))
    length = draw_length(rng, mean_residues, sd_residues, min_residues, max_residues)
    xyz, name, res, rord, rel = globule(rng, length)
    bfac = plddt_like(rng, rel)[rord]
    return dict(xyz=np.round(xyz, 3), name=name, residue_name=res, residue_number=(rord + 1).astype(np.int32),
                temperature_factor=bfac, chain_id=np.full(len(rord), b" A", dtype="S2"))


_CHAIN_LETTERS = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789"


def complex_columns(index: int, n_residues: int = 6000, chain_mean: float = 300.0, chain_sd: float = 30.0):
    """A multi-chain complex of about ``n_residues`` residues: globules of ~``chain_mean`` residues, one chain
    each, on a jittered close-packed (fcc) lattice, neighbouring globules in contact; residues of different
    chains closer than CLASH are thinned like wedge interfaces.  Residue numbers restart at 1 in every chain."""
    rng = np.random.default_rng(BASE_SEED + 7_000_000 + int(index))
    n_chains = max(1, int(np.ceil(n_residues / (0.89 * chain_mean))))  # about 11 % of the residues go at the contacts
    if n_chains > len(_CHAIN_LETTERS):
        raise ValueError(f"at most {len(_CHAIN_LETTERS)} chains")
    # fcc lattice points nearest the origin; nearest-neighbour distance = globule diameter at native density
    spacing = 2.0 * (3.0 * chain_mean * 135.0 / (4.0 * np.pi)) ** (1.0 / 3.0)
    g = np.arange(-4, 5)
    pts = np.array([(i, j, k) for i in g for j in g for k in g if (i + j + k) % 2 == 0], dtype=np.float64)
    pts *= spacing / np.sqrt(2.0)
    pts = pts[np.argsort((pts ** 2).sum(1), kind="stable")][:n_chains]
    xs, names, ress, rids, chain, bfs = [], [], [], [], [], []
    base = 0
    for c in range(n_chains):
        x, name, res, rord, rel = globule(rng, draw_length(rng, chain_mean, chain_sd, 150, 450))
        xs.append(x + pts[c] + rng.normal(scale=0.5, size=3))
        names.append(name); ress.append(res)
        rids.append(rord + base)
        bfs.append(plddt_like(rng, rel))
        n_c = int(rord.max()) + 1
        chain.append(np.full(n_c, c))
        base += n_c
    X, RID, C = np.vstack(xs), np.concatenate(rids), np.concatenate(chain)
    name, res = np.concatenate(names), np.concatenate(ress)
    present = _independent_set(base, RID, C, X, np.ones(base, dtype=bool), tie=np.zeros(base))
    am = present[RID]
    X, name, res, RID = X[am], name[am], res[am], RID[am]
    chain_of_atom = C[RID]
    # renumber inside every chain
    first_of_chain = np.zeros(n_chains, dtype=np.int64)
    new_ord = np.cumsum(present) - 1
    for c in range(n_chains):
        members = np.nonzero(present & (C == c))[0]
        first_of_chain[c] = new_ord[members[0]] if len(members) else 0
    resnum = (new_ord[RID] - first_of_chain[chain_of_atom] + 1).astype(np.int32)
    bfac = np.concatenate(bfs)[np.nonzero(present)[0]][new_ord[RID]]
    chain_id = np.array([(" " + _CHAIN_LETTERS[c]).encode() for c in range(n_chains)], dtype="S2")[chain_of_atom]
    return dict(xyz=np.round(X, 3), name=name, residue_name=res, residue_number=resnum, temperature_factor=bfac,
                chain_id=chain_id)


def _columns_job(job):
    kind, index, kw = job
    return (complex_columns if kind == "complex" else structure_columns)(index, **kw)
