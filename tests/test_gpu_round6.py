"""GPU tests added in round 6: the 16-bit limits that are left beside the 24-bit atom indices -- residues of
one structure, list positions of one template-atom kind -- take ONE structure out of a batch, not the batch: it stays
as an empty structure (no hits), it is listed (tm_batch_refused / tm_result_unsearched, Matcher.refused), and the other
structures get exactly the hits they get without it.  Every test calls through the C ABI."""
import warnings

import numpy as np
import pytest

from tests import helpers
from tests.test_gpu_parity import compare
from tests.test_gpu_round5 import _encoded_batch
from enzymm_amd import _native, ingest, jess_run, pyjess, synthetic, template
from oracle import oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def af():
    return pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb")


@pytest.fixture(scope="module")
def amy():
    return pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb")


def _water_box(n_res=66000):
    """A structure of ``n_res`` one-atom residues (waters on a 3 A lattice, 10 chains): more residues than the kernels
    number with 16 bits (TM_MAX_STRUCTURE_RESIDUES = 65533), and more than 65 535 atoms."""
    k = np.arange(n_res)
    side = 41
    xyz = np.stack([(k % side) * 3.0 + 500.0, ((k // side) % side) * 3.0, (k // (side * side)) * 3.0], axis=1)
    chains = np.array([(" " + "ABCDEFGHIJ"[i // 9000]).encode() for i in range(n_res)], dtype="S2")
    return pyjess.Molecule.from_columns(id="water_box", xyz=np.round(xyz, 3), name=np.full(n_res, b" O  ", "S4"),
                                        residue_name=np.full(n_res, b"HOH", "S3"), chain_id=chains,
                                        residue_number=(k % 9000 + 1).astype(np.int32), element=np.full(n_res, b" O", "S2"))


def test_structure_with_too_many_residues_is_refused_not_its_batch(af, amy, tmp_path):
    """66 000 one-atom residues: beyond TM_MAX_STRUCTURE_RESIDUES.  Until round 6 such a structure failed the whole batch
    with TM_E_UNSUPPORTED on all three ways into a batch (and Matcher.run_files raised); now it is emptied and listed, and
    the batch's other structures get byte-identical hit tables -- host atom tables, encoded columns, PDB files, Matcher."""
    box = _water_box()
    assert len(box) == 66000 > _native.TM_MAX_STRUCTURE_RESIDUES
    idx, cut = helpers.default_selection()
    every = np.arange(0, len(idx), 7)
    jess = pyjess.Jess(helpers.product_library_templates())
    q = lambda mols, **kw: jess.query_batch(mols, 2.0, cut[every], cut[every], max_candidates=10000, ignore_chain=True,
                                            template_index=idx[every], return_raw=True, **kw)
    others = [amy, synthetic.structure(7), af]
    ref = q(others)
    assert len(ref) > 3

    def without_box(table, at):
        assert not (table["structure"] == at).any()
        rest = table.copy()
        rest["structure"] -= (rest["structure"] > at).astype(np.uint32)
        return rest.tobytes()

    # host atom tables (tm_batch_create_from_structures): warned about, listed, the others unchanged
    with pytest.warns(UserWarning, match="residues are not searched"):
        batch = pyjess.make_batch([amy, box, synthetic.structure(7), af], jess._device)
    assert batch.refused() == [(1, 66000)] == batch.refused_list
    assert without_box(q([amy, box, synthetic.structure(7), af], batch=batch), 1) == ref.tobytes()
    batch.close()
    with pytest.warns(UserWarning):
        assert without_box(q([amy, box, synthetic.structure(7), af]), 1) == ref.tobytes()
    # together with a structure beyond the ATOM limit (a handcrafted descriptor): both listed, ascending
    rows = np.stack([box._pointer_row(), amy._pointer_row(), af._pointer_row()]).copy()
    rows[1, 0] = _native.TM_MAX_STRUCTURE_ATOMS + 1
    batch = _native.Batch.from_structures(jess._device, rows, pyjess.VOCAB.native_tables())
    assert batch.refused() == [(0, 66000), (1, _native.TM_MAX_STRUCTURE_ATOMS + 1)]
    batch.close()
    # the encoded columns of the C ABI (tm_batch_create)
    batch = _encoded_batch([amy, synthetic.structure(7), box, af], jess._device)
    assert batch.refused() == [(2, 66000)]
    batch.vocab_version = pyjess.VOCAB.version
    assert without_box(q([amy, synthetic.structure(7), box, af], batch=batch), 2) == ref.tobytes()
    batch.close()
    # PDB files (tm_batch_create_from_files: the device path sends the file to the host flattening, which counts its residues)
    paths = []
    for k, m in enumerate([amy, synthetic.structure(7), af, box]):
        path = tmp_path / f"s{k}.pdb"
        path.write_text(synthetic.to_pdb(m))
        paths.append(str(path))
    with pytest.warns(UserWarning):
        pdb = ingest.PdbBatch.from_paths(paths)
    assert pdb.refused == [(3, 66000)] and len(pdb.molecules[3]) == 0
    assert without_box(q(pdb.molecules, batch=pdb), 3) == ref.tobytes()
    pdb.close()
    # Matcher: run and run_files list it and report the others' matches
    tpls = list(template.load_templates(helpers.GOLDEN / "templates"))
    matcher = jess_run.Matcher(list(tpls), warn=False)
    want = matcher.run([amy, af])
    summary = lambda ms: sorted((tpls.index(x.hit.template), round(x.hit.rmsd, 6), tuple(a.serial for a in x.hit.atoms())) for x in ms)
    with pytest.warns(UserWarning):
        got = matcher.run([amy, box, af])
    assert [(m.id, n) for m, n in matcher.refused] == [("water_box", 66000)]
    assert summary(got[amy]) == summary(want[amy]) and summary(got[af]) == summary(want[af]) and not got.get(box)
    with pytest.warns(UserWarning):
        from_files = matcher.run_files(paths)
    by_id = {m.id: v for m, v in from_files.items()}
    assert [n for _, n in matcher.refused] == [66000]
    assert summary(by_id["s0"]) == summary(want[amy]) and summary(by_id["s2"]) == summary(want[af])


def _trp_lattice(amy, n_res=15000):
    """15 000 tryptophans on a 12 A lattice (210 000 atoms, 165 000 carbons) and a function making three-residue templates
    of it (tests/test_gpu_round5.py::test_kind_list_beyond_16_bit_positions_is_reported)."""
    c = amy._cols
    names = np.char.strip(c["residue_name"].astype("U3"))
    first = int(np.nonzero(names == "TRP")[0][0])
    key = (c["chain_id"][first], c["residue_number"][first])
    sel = np.nonzero((c["chain_id"] == key[0]) & (c["residue_number"] == key[1]))[0]
    side = 25
    cols = {k: np.tile(v[sel], n_res) for k, v in c.items() if k != "serial"}
    cols["residue_number"] = np.repeat(np.arange(n_res, dtype=np.int32) % 9000 + 1, len(sel))
    cols["chain_id"] = np.repeat(np.array([(" " + "ABCDEFGHIJ"[k // 9000]).encode() for k in range(n_res)], dtype="S2"), len(sel))
    shift = np.array([[(k % side) * 12.0, ((k // side) % side) * 12.0, (k // (side * side)) * 12.0] for k in range(n_res)])
    xyz = np.round((amy._xyz[sel][None, :, :] + shift[:, None, :]).reshape(-1, 3), 3)
    mol = pyjess.Molecule.from_columns(id="trp_lattice", xyz=xyz, **cols)
    base = amy._xyz[sel]
    atom_names = [bytes(x).decode().strip() for x in c["name"][sel]]

    def make(mode, picks, ident):
        atoms = []
        for r, offset in enumerate(([0.0, 0.0, 0.0], [12.0, 0.0, 0.0], [0.0, 12.0, 0.0])):
            for p in picks:
                x, y, z = base[p] + np.array(offset)
                atoms.append(pyjess.TemplateAtom(chain_id="A", residue_number=r + 1, x=float(x), y=float(y), z=float(z),
                                                 residue_names=["TRP"], atom_names=[atom_names[p]], distance_weight=0.0, match_mode=mode))
        return pyjess.Template(atoms, id=ident)

    carbons = [i for i, nm in enumerate(atom_names) if nm.startswith("C") and nm not in ("C", "CA")][:3]
    return mol, make, carbons


def test_unsearchable_structure_leaves_its_batch_searched(af, amy):
    """A wide structure with more than 65 535 atoms admissible to ONE template-atom kind (165 000 carbons of tryptophan
    against a template that matches by element) cannot be searched with that library: the pair tables hold list positions
    in 16 bits.  Until round 6 the whole query ended with TM_E_UNSUPPORTED; now that structure alone gets empty lists --
    no hits, listed by tm_result_unsearched, a warning, Matcher.refused -- and the other structures of the batch get the
    hits they get without it.  Against a library it CAN be searched with, the same batch object is searched in full."""
    mol, make, carbons = _trp_lattice(amy)
    assert len(mol) == 210000
    by_element = pyjess.Jess([make(3, carbons, "by_element")] + list(helpers.product_library_templates()[:300:3]))
    others = [amy, synthetic.structure(3), af]
    kw = dict(max_candidates=100000, ignore_chain=True, return_raw=True)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        ref = by_element.query_batch(others, 2.0, 1.0, 1.0, **kw)
    assert len(ref) > 0
    with pytest.warns(UserWarning, match="admissible to one template atom kind"):
        pending = by_element.query_batch([amy, synthetic.structure(3), mol, af], 2.0, 1.0, 1.0, _async=True, **kw)
        got = pending.result()
    assert pending.unsearched == [2]
    assert not (got["structure"] == 2).any()
    rest = got.copy()
    rest["structure"] -= (rest["structure"] > 2).astype(np.uint32)
    assert rest.tobytes() == ref.tobytes()
    # alone: an empty table, not an error
    with pytest.warns(UserWarning):
        assert len(by_element.query_batch([mol], 2.0, 0.5, 0.5, **kw)) == 0
    # the same molecules against templates with exact names (lists of 15 000): everything is searched, nothing is listed
    exact = pyjess.Jess([make(0, carbons, "exact_names")])
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        pending = exact.query_batch([amy, mol, af], 2.0, 0.5, 0.5, _async=True, **kw)
        table = pending.result()
    assert pending.unsearched == [] and (table["structure"] == 1).sum() == 1 and table[table["structure"] == 1]["rmsd"][0] < 1e-3
    # Matcher lists it
    tpl = template.load_templates(helpers.GOLDEN / "templates")
    matcher = jess_run.Matcher(list(tpl), warn=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        matcher.run([amy, mol])
    # (the golden templates match TRP carbons by exact name or not at all: whether the lattice is searchable with them is
    # the engine's call -- either it is listed or it was searched)
    assert matcher.refused in ([], [(mol, 210000)])


# ------------------------------------------------------------------------------------------- stage D0's scan (round 6)


def _padded(mol, every=7, extra=37, ident="padded"):
    """``mol`` with ``extra`` dummy atoms (hydrogens named H001 ..., far from everything) put IN FRONT of the atoms of every
    ``every``-th residue, so that the residue's real atoms sit beyond position 32 of a residue of more than 32 atoms."""
    c = mol._cols
    keys = list(zip(c["chain_id"].tolist(), c["residue_number"].tolist(), c["insertion_code"].tolist()))
    starts = [0] + [i for i in range(1, len(keys)) if keys[i] != keys[i - 1]] + [len(keys)]
    order, dummy = [], []
    for r in range(len(starts) - 1):
        lo, hi = starts[r], starts[r + 1]
        if r % every == 0:
            order.extend([-(lo + 1)] * extra)  # (negative: a dummy copied from atom lo)
        order.extend(range(lo, hi))
    src = np.array([(-o - 1) if o < 0 else o for o in order])
    is_dummy = np.array([o < 0 for o in order])
    cols = {k: v[src].copy() for k, v in c.items() if k != "serial"}
    n_d = int(is_dummy.sum())
    cols["name"][is_dummy] = np.array([("H%03d" % (i % 1000)).encode() for i in range(n_d)], dtype="S4")
    cols["element"][is_dummy] = b" H"
    xyz = mol._xyz[src].copy()
    xyz[is_dummy] = np.round(xyz[is_dummy] + 400.0 + np.arange(n_d)[:, None] * 0.001, 3)
    return pyjess.Molecule.from_columns(id=ident, xyz=xyz, **cols)


@pytest.mark.parametrize("which", ["lds", "global"])
def test_residues_of_more_than_32_atoms(amy, af, which):
    """Stage D0 scans a residue's names 32 atoms at a time (round 6).  Residues padded to 40-50 atoms with dummy hydrogens in
    FRONT of their real atoms -- the atoms a level asks for sit beyond position 32 -- against implanted templates that hit
    those residues and every fifth template of the default selection: hit sets, atoms, RMSD and candidate counts equal the
    oracle's, with the structure in LDS (1AMY, every thirteenth residue padded) and in global memory (AF-P0DUB6, all
    of them: 23 000 atoms)."""
    base = amy if which == "lds" else af
    mol = _padded(base, every=13 if which == "lds" else 1, ident="padded_" + which)
    assert (len(mol) < 5200) == (which == "lds")
    implanted = helpers.implant_templates(mol, helpers.product_library_templates(), np.random.default_rng(17), max_templates=150)
    assert len(implanted) > 100
    jess = pyjess.Jess(implanted)
    table, cands = jess.query_batch([mol, synthetic.structure(5)], 2.0, 1.0, 1.0, max_candidates=10000, ignore_chain=True,
                                    count_candidates=True, return_raw=True)
    olib = oracle.OracleLibrary([helpers.to_oracle_template(t) for t in implanted])
    hits = 0
    for s, m in enumerate([mol, synthetic.structure(5)]):
        want = oracle.query_library(helpers.to_oracle_molecule(m), olib, 2.0, 1.0, 1.0, max_candidates=10000, ignore_chain=True)
        hits += compare(table, cands, s, want, olib)
    assert hits > 100
    # ... and the library's own templates (sparse hits; every level of D0 asked of padded residues)
    idx, cut = helpers.default_selection()
    every = np.arange(0, len(idx), 5)
    lib = pyjess.Jess(helpers.product_library_templates())
    table, cands = lib.query_batch([mol], 2.0, cut[every], cut[every], max_candidates=10000, ignore_chain=True,
                                   template_index=idx[every], count_candidates=True, return_raw=True)
    otpl = helpers.oracle_library_templates()
    olib2 = oracle.OracleLibrary([otpl[i] for i in idx[every]])
    want = oracle.query_library(helpers.to_oracle_molecule(mol), olib2, 2.0, cut[every], cut[every], max_candidates=10000, ignore_chain=True)
    compare(table, cands, 0, want, olib2)
