"""Shared test helpers: the same inputs loaded for the product (enzymm_amd.pyjess) and for the oracle."""
from __future__ import annotations

import functools
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
LIBRARY = ROOT / "enzymm_amd" / "data" / "jess_templates_20230210.txt.gz"

# Matcher._DEFAULT_JESS_PARAMS, /root/reference/enzymm/jess_run.py:564-571
DEFAULT_DISTANCE = {3: 0.9, 4: 1.7, 5: 2.0, 6: 2.0, 7: 2.0, 8: 2.0}


@functools.lru_cache(maxsize=None)
def library_texts():
    from oracle import parse
    return parse.read_packed_library(LIBRARY)


@functools.lru_cache(maxsize=None)
def oracle_library_templates():
    from oracle import parse
    return [parse.parse_template(text) for _, text in library_texts()]


@functools.lru_cache(maxsize=None)
def product_library_templates():
    from enzymm_amd import pyjess
    return [pyjess.Template.loads(text, id=name) for name, text in library_texts()]


def default_selection():
    """Indices + per-template distance cutoffs of the templates enzymm searches by default
    (effective size >= 3, jess_run.py:932-934), in library order."""
    eff = np.array([t.effective_size for t in oracle_library_templates()])
    idx = np.nonzero(eff >= 3)[0]
    cut = np.array([DEFAULT_DISTANCE[min(int(e), 8)] for e in eff[idx]])
    return idx, cut


def golden_text(rel):
    return (GOLDEN / rel).read_text()


# ---------------------------------------------------------------------------------------------
# product objects -> oracle tables, and an oracle-backed stand-in for pyjess.Jess (CPU tests only)


def to_oracle_molecule(mol):
    from oracle import parse
    c = mol._cols
    return parse.OracleMolecule(mol._xyz, c["name"], c["residue_name"], c["chain_id"], c["residue_number"],
                                c["insertion_code"], c["temperature_factor"], c["serial"])


def to_oracle_template(tpl):
    from oracle import parse
    t = parse.OracleTemplate([], [])
    k = len(tpl)
    t.k = k
    t.xyz = np.array([[a.x, a.y, a.z] for a in tpl], dtype=np.float64).reshape(k, 3)
    t.mode = np.array([a.match_mode for a in tpl], dtype=np.int32)
    t.name = np.array([a._raw_names[0] for a in tpl], dtype="S4")
    t.resnames = np.zeros((k, parse.MAXRESN), dtype="S3")
    t.nres = np.zeros(k, dtype=np.int32)
    for i, a in enumerate(tpl):
        t.nres[i] = len(a._raw_residues)
        for r, nm in enumerate(a._raw_residues):
            t.resnames[i, r] = nm
    t.chain = np.array([a.chain_id.rjust(2).encode() for a in tpl], dtype="S2")
    t.resseq = np.array([a.residue_number for a in tpl], dtype=np.int32)
    t.weight = np.array([a.distance_weight for a in tpl], dtype=np.float64)
    return t


class OracleJess:
    """Duck-types enzymm_amd.pyjess.Jess on top of the CPU oracle so host logic (Match, Matcher)
    can be exercised without a GPU.  TESTS ONLY."""

    def __init__(self, templates=(), device=None):
        self._templates = tuple(templates)

    def query_batch(self, molecules, rmsd_threshold, distance_cutoff, max_dynamic_distance, *,
                    max_candidates=None, ignore_chain=False, best_match=True, **_):  # `models` ignored: no epilogue here
        from enzymm_amd import pyjess
        from oracle import oracle
        assert best_match
        nt = len(self._templates)
        if nt == 0:
            return [[] for _ in molecules]
        lib = oracle.OracleLibrary([to_oracle_template(t) for t in self._templates])
        cap = 2**31 - 1 if max_candidates is None else int(max_candidates)
        out = []
        for mol in molecules:
            r = oracle.query_library(to_oracle_molecule(mol), lib, rmsd_threshold, distance_cutoff,
                                     max_dynamic_distance, max_candidates=cap, ignore_chain=ignore_chain)
            hits = []
            for t in np.nonzero(r["hit"])[0]:
                k = lib.templates[t].k
                hits.append(pyjess.Hit(self._templates[t], mol, r["rmsd"][t], r["rot"][t], r["trans"][t],
                                       r["atoms"][t][:k], r["candidates"][t], bool(r["truncated"][t])))
            out.append(hits)
        return out

    def query(self, molecule, rmsd_threshold, distance_cutoff, max_dynamic_distance, *, max_candidates=None,
              ignore_chain=False, best_match=False):
        from enzymm_amd import pyjess
        hits = self.query_batch([molecule], rmsd_threshold, distance_cutoff, max_dynamic_distance,
                                max_candidates=max_candidates, ignore_chain=ignore_chain, best_match=best_match)
        return pyjess.Query(self, molecule, rmsd_threshold, distance_cutoff, max_dynamic_distance, max_candidates,
                            ignore_chain, best_match, hits[0])


def implant_templates(mol, templates, rng, max_templates=200, radius=14.0):
    """Re-seat library templates on residues of ``mol``: every template atom takes the coordinates of the
    same-named atom of a randomly chosen residue of an admissible type, residues picked within ``radius`` of the
    first one.  The result is a list of pyjess.Template that are guaranteed to hit ``mol`` (RMSD 0 plus all
    the symmetric alternatives) -- a dense parity workload, unlike the sparse hits of real templates."""
    from enzymm_amd import pyjess
    c = mol._cols
    names = np.char.strip(c["name"].astype("U4"))
    resn = np.char.strip(c["residue_name"].astype("U3"))
    keys = list(zip(c["chain_id"].tolist(), c["residue_number"].tolist(), c["insertion_code"].tolist()))
    by_res = {}
    for i, k in enumerate(keys):
        by_res.setdefault(k, []).append(i)
    residues = [(k, resn[idx[0]], {names[i]: i for i in idx}, mol._xyz[idx].mean(0)) for k, idx in by_res.items()]
    out = []
    order = rng.permutation(len(templates))
    for ti in order:
        if len(out) >= max_templates:
            break
        tpl = templates[ti]
        groups = {}
        for a in tpl:
            groups.setdefault((a.chain_id, a.residue_number), []).append(a)
        chosen, used, centre, ok = {}, set(), None, True
        for g, atoms in groups.items():
            allowed = set(atoms[0].residue_names)
            cands = [r for r in residues if r[0] not in used and (atoms[0].match_mode >= 100 or r[1] in allowed)
                     and all(a.atom_names[0] in r[2] for a in atoms)
                     and (centre is None or np.linalg.norm(r[3] - centre) < radius)]
            if not cands:
                ok = False
                break
            r = cands[int(rng.integers(len(cands)))]
            chosen[g] = r
            used.add(r[0])
            centre = r[3] if centre is None else centre
        if not ok:
            continue
        new_atoms = []
        for a in tpl:
            r = chosen[(a.chain_id, a.residue_number)]
            x, y, z = mol._xyz[r[2][a.atom_names[0]]]
            b = a.copy()
            b.x, b.y, b.z = float(x), float(y), float(z)
            new_atoms.append(b)
        out.append(pyjess.Template(new_atoms, id=f"implant_{ti}"))
    return out


def check_completeness_vector(Matcher, jess_factory, amy, af):
    """Matcher._check_completeness (/root/reference/enzymm/jess_run.py:738-783) on a real multi-member cluster: the
    3-residue templates of M-CSA 285 form two clusters of two members each.  1AMY is hit by both members of
    cluster 1 (complete), by neither of cluster 2; with member 2 of cluster 1 withheld, member 1's match is
    incomplete.  AF-P0DUB6 needs the looser 1.2 A to be hit at all."""
    from enzymm_amd import template
    tpls = list(template.load_templates(GOLDEN / "templates" / "3_residues" / "results" / "csa3d_0285"))
    assert sorted((t.cluster.id, t.cluster.member, t.cluster.size) for t in tpls) == \
        [(1, 1, 2), (1, 2, 2), (2, 1, 2), (2, 2, 2)]
    summary = lambda ms: sorted((m.hit.template.cluster.id, m.hit.template.cluster.member, m.complete) for m in ms)
    out = Matcher(tpls, filter_matches=False, jess_factory=jess_factory).run([amy, af])
    assert summary(out[amy]) == [(1, 1, True), (1, 2, True)] and af not in out
    partial = [t for t in tpls if (t.cluster.id, t.cluster.member) != (1, 2)]
    out = Matcher(partial, filter_matches=False, jess_factory=jess_factory).run([amy, af])
    assert summary(out[amy]) == [(1, 1, False)]
    loose = {s: {"rmsd": 2.0, "distance": 1.2, "max_dynamic_distance": 1.2} for s in range(3, 9)}
    out = Matcher(tpls, jess_params=loose, filter_matches=False, jess_factory=jess_factory).run([amy, af])
    assert summary(out[amy]) == summary(out[af]) == [(1, 1, True), (1, 2, True)]


# ---------------------------------------------------------------------------------------------
# constructed cases of round 3


def _kabsch_rmsd(q, p):
    """RMSD of q onto p after the optimal proper rotation (numpy SVD; fp64)."""
    qc, pc = q - q.mean(0), p - p.mean(0)
    u, s, vt = np.linalg.svd(qc.T @ pc)
    d = np.sign(np.linalg.det(u @ vt))
    e = (qc * qc).sum() + (pc * pc).sum() - 2.0 * (s[0] + s[1] + d * s[2])
    return float(np.sqrt(max(e, 0.0) / len(q)))


def subset_molecule(mol, index, xyz=None, id=None, residue_offset=0, chain=None):
    """A pyjess.Molecule of the atoms ``index`` of ``mol`` (file order kept), optionally with other coordinates,
    residue numbers shifted and another chain id."""
    from enzymm_amd import pyjess
    c = {k: v[index].copy() for k, v in mol._cols.items()}
    c["residue_number"] = c["residue_number"] + residue_offset
    if chain is not None:
        c["chain_id"][:] = chain.rjust(2).encode()
    return pyjess.Molecule.from_columns(id=id or mol.id, xyz=np.asarray(mol._xyz[index] if xyz is None else xyz, dtype=np.float64),
                                        **{k: v for k, v in c.items() if k != "serial"})


def concat_molecules(mols, id="joined"):
    from enzymm_amd import pyjess
    cols = {k: np.concatenate([m._cols[k] for m in mols]) for k in mols[0]._cols if k != "serial"}
    return pyjess.Molecule.from_columns(id=id, xyz=np.concatenate([m._xyz for m in mols]), **cols)


def tie_case(golden):
    """Three copies of the site that template G1 hits in 1AMY (tests/test_jess_run.py:24-44), 40 A apart, each scaled
    about its centroid by a hair so that the copies' RMSDs are r + 1.2e-9, r + 0.6e-9 and r (file order).  The copies
    are pairwise closer than the 1e-9 tie width but only the second and third lie within it of the MINIMUM: the
    global-minimum rule (oracle/jess_oracle.c header, k_select) picks the second copy -- the smallest atom tuple among
    the candidates tied with the minimum.  A running "better than the best so far by more than eps" comparison picks
    the third or the first, depending on the order in which it meets them.
    Returns (molecule, template, atoms per copy, index of the expected copy = 1)."""
    from enzymm_amd import pyjess, template
    from oracle import oracle, parse
    g = golden["G1"]
    amy = pyjess.Molecule.load(GOLDEN / g["molecule"])
    tpl = template.Template.load(GOLDEN / "templates" / g["template"])
    otpl = parse.parse_template((GOLDEN / "templates" / g["template"]).read_text())
    hits, _, _ = oracle.query(to_oracle_molecule(amy), otpl, g["rmsd_threshold"], g["distance_cutoff"],
                              g["max_dynamic_distance"], max_candidates=g["max_candidates"], best_match=True)
    site = np.sort(hits[0].atoms)  # file order
    pos_of = {int(a): i for i, a in enumerate(site)}
    in_template_order = [pos_of[int(a)] for a in hits[0].atoms]
    txyz = np.array([[a.x, a.y, a.z] for a in tpl])
    base = amy._xyz[site]
    centre = base.mean(0)

    def rmsd_at(scale):
        return _kabsch_rmsd(((base - centre) * (1.0 + scale) + centre)[in_template_order], txyz)

    r0 = rmsd_at(0.0)
    slope = (rmsd_at(1e-6) - r0) / 1e-6
    copies = []
    for c, delta in enumerate((1.2e-9, 0.6e-9, 0.0)):
        scale = delta / slope
        for _ in range(4):  # secant refinement: the RMSD is smooth in the scale
            scale -= (rmsd_at(scale) - (r0 + delta)) / slope
        assert abs(rmsd_at(scale) - (r0 + delta)) < 2e-11
        xyz = (base - centre) * (1.0 + scale) + centre + np.array([40.0 * c, 0.0, 0.0])
        copies.append(subset_molecule(amy, site, xyz=xyz, residue_offset=1000 * c))
    mol = concat_molecules(copies, id="tied_sites")
    n = len(site)
    return mol, tpl, [np.arange(c * n, (c + 1) * n) for c in range(3)], 1


def run_child(cmd, env, timeout=300, cwd=None):
    """A child python process (a bench line, an RCCL world-size-1 group) with a deadline: when it does not finish in time
    its whole process group is killed and it is started ONCE more.  In round 4 one such child in a few dozen hung on a
    fresh GPU box -- python processes importing torch and meeting over gloo / RCCL while the box is still paging the
    image in; none of the three children that use this hung when run alone in a loop (6, 4 and 4 runs in a row)."""
    import os
    import signal
    import subprocess
    for attempt in range(2):
        child = subprocess.Popen(cmd, cwd=cwd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                 start_new_session=True)
        try:
            stdout, stderr = child.communicate(timeout=timeout)
            return subprocess.CompletedProcess(cmd, child.returncode, stdout, stderr)
        except subprocess.TimeoutExpired:
            os.killpg(child.pid, signal.SIGKILL)  # (the launcher and its ranks: exactly the process group started above)
            child.communicate()
            if attempt == 1:
                raise

