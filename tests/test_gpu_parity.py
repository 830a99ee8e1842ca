"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference's known answers.
Needs a real MI355X: every test here is marked gpu."""
import io
import json

import numpy as np
import pytest

from tests import helpers
from enzymm_amd import _native, jess_run, pyjess, synthetic, template
from oracle import oracle

pytestmark = pytest.mark.gpu

G = json.loads((helpers.GOLDEN / "golden.json").read_text())
TPL = helpers.GOLDEN / "templates"
RMSD_TOL = 1e-5  # north_star: RMSD within 1e-5 (observed agreement is ~1e-12)


@pytest.fixture(scope="module")
def amy():
    return pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb")


@pytest.fixture(scope="module")
def af():
    return pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb", id="AF-P0DUB6-F1-model_v4")


@pytest.fixture(scope="module")
def default_library():
    idx, cut = helpers.default_selection()
    ptpl = helpers.product_library_templates()
    otpl = helpers.oracle_library_templates()
    return pyjess.Jess(ptpl), oracle.OracleLibrary([otpl[i] for i in idx]), idx, cut


def compare(table, cands, s, r, lib, check_candidates=True):
    """GPU hit table rows of structure s vs the oracle's query_library result r."""
    mine = {int(h["template_sel"]): h for h in table if int(h["structure"]) == s}
    if check_candidates:
        assert np.array_equal(cands[s] & 0x7FFFFFFF, r["candidates"])  # same number of complete assignments
        assert np.array_equal((cands[s] >> 31).astype(bool), r["truncated"].astype(bool))
    ok = ~r["truncated"].astype(bool)  # parity is defined for searches that were not cut short
    theirs = set(np.nonzero(r["hit"] & ok)[0].tolist())
    assert {t for t in mine if ok[t]} == theirs
    for t in theirs:
        k = lib.templates[t].k
        assert np.array_equal(mine[t]["atoms"][:k], r["atoms"][t][:k]), t  # bit-identical hit set
        assert abs(mine[t]["rmsd"] - r["rmsd"][t]) < RMSD_TOL
        assert np.abs(mine[t]["rot"].reshape(3, 3) - r["rot"][t]).max() < 1e-6
        assert np.abs(mine[t]["trans"] - r["trans"][t]).max() < 1e-5
        assert int(mine[t]["n_candidates"]) == int(r["candidates"][t])
    return len(theirs)


@pytest.mark.parametrize("key", ["G1", "G2"])
def test_reference_known_answers(amy, key):
    """tests/test_jess_run.py:24-59, 77, 133: the exact reference call, on the GPU."""
    g = G[key]
    tpl = template.Template.load(TPL / g["template"])
    hits = list(pyjess.Jess([tpl]).query(amy, g["rmsd_threshold"], g["distance_cutoff"], g["max_dynamic_distance"],
                                         max_candidates=g["max_candidates"], best_match=True))
    assert len(hits) == 1 and hits[0].template is tpl
    m = jess_run.Match(hit=hits[0], pairwise_distance=g["distance_cutoff"], complete=True, index=0)
    assert m.hit.rmsd == pytest.approx(g["rmsd"], abs=5e-8)
    assert m.orientation == pytest.approx(g["orientation"], abs=5e-8)
    assert [list(r) for r in m.matched_residues] == g["matched_residues"]
    assert m.preserved_resid_order == g["preserved_resid_order"]
    if key == "G1":
        for fname, kw in (("1AMY_matches_no_query.pdb", dict(transform=False, include_query=False)),
                          ("1AMY_matches_template.pdb", dict(transform=True, include_query=False)),
                          ("1AMY_matches_query_included.pdb", dict(transform=False, include_query=True))):
            buf = io.StringIO()
            m.dump2pdb(buf, **kw)
            assert buf.getvalue() == (helpers.GOLDEN / fname).read_text()


def test_full_library_real_structures(amy, af, default_library):
    """configs[1]: one structure vs the full shipped library -- hit sets, candidate counts, RMSD."""
    jess, olib, idx, cut = default_library
    table, cands = jess.query_batch([amy, af], 2.0, cut, cut, max_candidates=10000, ignore_chain=True,
                                    template_index=idx, count_candidates=True, return_raw=True)
    n = 0
    for s, mol in enumerate((amy, af)):
        r = oracle.query_library(helpers.to_oracle_molecule(mol), olib, 2.0, cut, cut, max_candidates=10000,
                                 ignore_chain=True)
        n += compare(table, cands, s, r, olib)
    assert n >= 20


def test_synthetic_batch_vs_oracle(default_library):
    jess, olib, idx, cut = default_library
    mols = synthetic.structures(24)
    table, cands = jess.query_batch(mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True,
                                    template_index=idx, count_candidates=True, return_raw=True)
    for s, mol in enumerate(mols):
        r = oracle.query_library(helpers.to_oracle_molecule(mol), olib, 2.0, cut, cut, max_candidates=10000,
                                 ignore_chain=True)
        compare(table, cands, s, r, olib)


def test_implanted_templates_dense_hits(amy, af):
    """Library templates re-seated on residues of the query: every one of them hits (RMSD 0) and has symmetric
    alternatives (OD1/OD2 swaps, ring flips) -- exercises best-hit selection, ties and deep levels."""
    rng = np.random.default_rng(7)
    lib = helpers.product_library_templates()
    big = [t for t in lib if len(t) >= 15]
    for mol in (amy, af, synthetic.structure(5)):
        tpls = helpers.implant_templates(mol, lib, rng, max_templates=150) + \
               helpers.implant_templates(mol, big, rng, max_templates=50)
        assert len(tpls) > 120
        jess, olib = pyjess.Jess(tpls), oracle.OracleLibrary([helpers.to_oracle_template(t) for t in tpls])
        for cutoff in (0.9, 2.0):
            table, cands = jess.query_batch([mol], 2.0, cutoff, cutoff, max_candidates=10000, ignore_chain=True,
                                            count_candidates=True, return_raw=True)
            r = oracle.query_library(helpers.to_oracle_molecule(mol), olib, 2.0, cutoff, cutoff,
                                     max_candidates=10000, ignore_chain=True)
            n = compare(table, cands, 0, r, olib)
            assert n == len(tpls) - int(r["truncated"].sum())  # every implanted template hits
            assert (r["rmsd"][r["hit"] == 1] < 1e-6).sum() >= 0.9 * n


def test_one_and_two_residue_templates(amy):
    """Templates cut down to their first one / two residues: the tiled scans run with wildcard lists for the
    missing residues (and long lead lists span several 64-entry blocks), with and without chain tests."""
    lib = helpers.product_library_templates()
    rng = np.random.default_rng(11)
    tpls = []
    for ti in rng.permutation(len(lib))[:60]:
        groups = {}
        for a in lib[ti]:
            groups.setdefault((a.chain_id, a.residue_number), []).append(a)
        keys = list(groups)
        for n_res in (1, 2):
            atoms = [a.copy() for g in keys[:n_res] for a in groups[g]]
            tpls.append(pyjess.Template(atoms, id=f"{lib[ti].id}_first{n_res}"))
    jess, olib = pyjess.Jess(tpls), oracle.OracleLibrary([helpers.to_oracle_template(t) for t in tpls])
    dimer = _multichain(amy, 2, shift=38.0)
    for mol, ignore in ((amy, True), (dimer, False), (dimer, True)):
        table, cands = jess.query_batch([mol], 2.0, 1.0, 1.0, max_candidates=100000, ignore_chain=ignore,
                                        count_candidates=True, return_raw=True)
        r = oracle.query_library(helpers.to_oracle_molecule(mol), olib, 2.0, 1.0, 1.0, max_candidates=100000,
                                 ignore_chain=ignore)
        assert compare(table, cands, 0, r, olib) >= 80


def test_randomised_parameters_vs_oracle(amy, af):
    """Seeded random draws of everything a caller can vary: template subset, per-template thresholds
    (max_dynamic_distance above and below the cutoff), ignore_chain, max_candidates small enough to truncate,
    and a batch mixing real, synthetic, multi-chain and empty structures."""
    rng = np.random.default_rng(2024)
    ptpl, otpl = helpers.product_library_templates(), helpers.oracle_library_templates()
    dimer = _multichain(amy, 2, shift=38.0)
    pool = [amy, af, dimer, pyjess.Molecule()] + synthetic.structures(6)
    checked = truncated = 0
    for trial in range(6):
        idx = np.sort(rng.choice(len(ptpl), size=120, replace=False))
        jess, olib = pyjess.Jess([ptpl[i] for i in idx]), oracle.OracleLibrary([otpl[i] for i in idx])
        rmsd = rng.uniform(0.8, 3.0, size=len(idx))
        cutoff = rng.uniform(0.6, 2.4, size=len(idx))
        maxdyn = cutoff + rng.uniform(-0.4, 0.6, size=len(idx))
        ignore = bool(trial % 2)
        cap = int(rng.choice([3, 40, 10000]))
        mols = [pool[i] for i in rng.choice(len(pool), size=4, replace=False)]
        table, cands = jess.query_batch(mols, rmsd, cutoff, maxdyn, max_candidates=cap, ignore_chain=ignore,
                                        count_candidates=True, return_raw=True)
        for s, mol in enumerate(mols):
            r = oracle.query_library(helpers.to_oracle_molecule(mol), olib, rmsd, cutoff, maxdyn, max_candidates=cap,
                                     ignore_chain=ignore)
            cut_short = r["truncated"].astype(bool)
            assert np.array_equal((cands[s] >> 31).astype(bool), cut_short)
            assert np.array_equal((cands[s] & 0x7FFFFFFF)[~cut_short], r["candidates"][~cut_short])
            checked += compare(table, cands, s, r, olib, check_candidates=False)
            truncated += int(cut_short.sum())
    assert checked > 10 and truncated > 0


def test_pdb_corner_cases(default_library):
    """Insertion codes, alternate locations, negative residue numbers: the oracle's parser, the host loader
    and the device-side ingestion must agree atom for atom, and the searches must agree hit for hit."""
    from enzymm_amd import ingest
    from oracle import parse
    jess, olib, idx, cut = default_library
    out = []
    for line in (helpers.GOLDEN / "1AMY.pdb").read_text().splitlines(keepends=True):
        if line.startswith("ATOM"):
            resseq = int(line[22:26])
            if resseq == 204:    # whole residue gets an insertion code
                line = line[:26] + "A" + line[27:]
            if resseq <= 10:     # negative numbering
                line = line[:22] + "%4d" % (resseq - 11) + line[26:]
            if resseq == 179:    # alternate location B, slightly displaced, after the A record
                a = line[:16] + "A" + line[17:]
                b = line[:16] + "B" + line[17:30] + "%8.3f" % (float(line[30:38]) + 0.3) + line[38:]
                out += [a, b]
                continue
        out.append(line)
    text = "".join(out)
    om = parse.parse_molecule(text)
    mol = pyjess.Molecule.loads(text, id="corner")
    assert len(om) == len(mol) and np.array_equal(om.xyz, mol._xyz)
    assert mol[0].residue_number == -10 and any(a.insertion_code == "A" for a in mol) and any(a.altloc == "B" for a in mol)
    r = oracle.query_library(om, olib, 2.0, cut, cut, max_candidates=10000, ignore_chain=True)
    table, cands = jess.query_batch([mol], 2.0, cut, cut, max_candidates=10000, ignore_chain=True,
                                    template_index=idx, count_candidates=True, return_raw=True)
    assert compare(table, cands, 0, r, olib) > 5
    pb = ingest.PdbBatch([text.encode()], ["corner"], device=jess._device)
    dev = jess.query_batch(pb.molecules, 2.0, cut, cut, max_candidates=10000, ignore_chain=True, template_index=idx,
                           batch=pb, return_raw=True)
    assert dev.tobytes() == table.tobytes()


def _subset(n_per_size=40):
    """A few templates of every effective size 3..8 (library order)."""
    otpl = helpers.oracle_library_templates()
    eff = np.array([t.effective_size for t in otpl])
    idx = np.concatenate([np.nonzero(eff == e)[0][:: max(1, (eff == e).sum() // n_per_size)][:n_per_size]
                          for e in range(3, 9)])
    return np.sort(idx)


def test_loose_thresholds_dynamic_distance_and_truncation(amy):
    """configs[4] flavour: loosened -j, max_dynamic_distance > distance_cutoff (distance weights take
    part), and a max_candidates small enough that some searches are cut short."""
    idx = _subset(25)
    ptpl, otpl = helpers.product_library_templates(), helpers.oracle_library_templates()
    jess, olib = pyjess.Jess([ptpl[i] for i in idx]), oracle.OracleLibrary([otpl[i] for i in idx])
    for cutoff, maxdyn, cap in ((2.0, 2.6, 10000), (2.5, 2.5, 50)):
        table, cands = jess.query_batch([amy], 3.0, cutoff, maxdyn, max_candidates=cap, ignore_chain=True,
                                        count_candidates=True, return_raw=True)
        r = oracle.query_library(helpers.to_oracle_molecule(amy), olib, 3.0, cutoff, maxdyn, max_candidates=cap,
                                 ignore_chain=True)
        # truncated pairs: both sides must agree WHICH pairs are truncated (the total is order independent)
        assert np.array_equal((cands[0] >> 31).astype(bool), r["truncated"].astype(bool))
        compare(table, cands, 0, r, olib, check_candidates=False)
        assert np.array_equal((cands[0] & 0x7FFFFFFF)[~r["truncated"].astype(bool)],
                              r["candidates"][~r["truncated"].astype(bool)])
    assert r["truncated"].sum() > 0


def _multichain(mol, copies, shift=45.0):
    """mol repeated as chains A, B, ... shifted along x (a crude oligomer)."""
    cols = {k: np.concatenate([v] * copies) for k, v in mol._cols.items()}
    n = len(mol)
    xyz = np.concatenate([mol._xyz + np.array([shift * c, 0.0, 0.0]) for c in range(copies)])
    cols["chain_id"] = np.concatenate([np.full(n, (" " + "ABCDEFGH"[c]).encode(), dtype="S2") for c in range(copies)])
    out = pyjess.Molecule.__new__(pyjess.Molecule)
    out._init_columns(cols, xyz, f"{mol.id}x{copies}")
    return out


def test_ignore_chain_false_and_multichain(amy):
    """ignore_chain=False (the default of the reference's test calls) on a two-chain target with
    multi-chain templates."""
    ptpl, otpl = helpers.product_library_templates(), helpers.oracle_library_templates()
    multi = [i for i, t in enumerate(ptpl) if len({a.chain_id for a in t}) > 1][:150]
    idx = np.array(sorted(set(multi) | set(_subset(10).tolist())))
    jess, olib = pyjess.Jess([ptpl[i] for i in idx]), oracle.OracleLibrary([otpl[i] for i in idx])
    dimer = _multichain(amy, 2, shift=38.0)
    for ignore in (False, True):
        table, cands = jess.query_batch([dimer], 2.0, 2.0, 2.0, max_candidates=10000, ignore_chain=ignore,
                                        count_candidates=True, return_raw=True)
        r = oracle.query_library(helpers.to_oracle_molecule(dimer), olib, 2.0, 2.0, 2.0, max_candidates=10000,
                                 ignore_chain=ignore)
        assert compare(table, cands, 0, r, olib) > 0


def test_large_structure_global_memory_path(af, default_library):
    """> 10k atoms does not fit the LDS-resident kernel variant: the global-memory variant must agree too."""
    jess, olib, idx, cut = default_library
    big = _multichain(af, 3, shift=70.0)
    assert len(big) == 3 * 4079
    sel = np.arange(0, len(idx), 7)
    sub = oracle.OracleLibrary([olib.templates[i] for i in sel])
    table, cands = jess.query_batch([big], 2.0, cut[sel], cut[sel], max_candidates=10000, ignore_chain=True,
                                    template_index=idx[sel], count_candidates=True, return_raw=True)
    r = oracle.query_library(helpers.to_oracle_molecule(big), sub, 2.0, cut[sel], cut[sel], max_candidates=10000,
                             ignore_chain=True)
    compare(table, cands, 0, r, sub)


def test_structure_near_the_atom_limit(af, default_library):
    """16 copies of a 4079-atom chain = 65 264 atoms, just under TM_NARROW_STRUCTURE_ATOMS: u16 atom indices and
    residue tables at their upper end, kind lists of thousands of entries, work items cut into many rounds."""
    jess, olib, idx, cut = default_library
    cols = {k: np.concatenate([v] * 16) for k, v in af._cols.items()}
    xyz = np.concatenate([af._xyz + np.array([70.0 * (c % 4), 70.0 * (c // 4), 0.0]) for c in range(16)])
    cols["chain_id"] = np.concatenate([np.full(len(af), (" " + "ABCDEFGHIJKLMNOP"[c]).encode(), dtype="S2")
                                       for c in range(16)])
    big = pyjess.Molecule.__new__(pyjess.Molecule)
    big._init_columns(cols, xyz, "AFx16")
    assert len(big) == 16 * 4079 <= _native.TM_NARROW_STRUCTURE_ATOMS
    sel = np.arange(3, len(idx), 41)
    sub = oracle.OracleLibrary([olib.templates[i] for i in sel])
    small = synthetic.structure(3)
    table, cands = jess.query_batch([small, big], 2.0, cut[sel], cut[sel], max_candidates=10000, ignore_chain=True,
                                    template_index=idx[sel], count_candidates=True, return_raw=True)
    for s, mol in enumerate((small, big)):
        r = oracle.query_library(helpers.to_oracle_molecule(mol), sub, 2.0, cut[sel], cut[sel], max_candidates=10000,
                                 ignore_chain=True)
        n = compare(table, cands, s, r, sub)
    assert n >= 1
    # one atom beyond the 16-bit range: 65 536 atoms, the smallest structure the template-at-a-time launch with 24-bit
    # indices takes (round 5; refused before: tests/test_gpu_round4.py::test_structure_beyond_65535_atoms_is_searched).
    # The 272 extra atoms are a shifted copy of the chain's first residues under a seventeenth chain id.
    m = 65536 - len(big)
    extra = {k: np.concatenate([v, v[:m]]) for k, v in cols.items()}
    extra["chain_id"] = np.concatenate([cols["chain_id"], np.full(m, b" Q", dtype="S2")])
    over = pyjess.Molecule.__new__(pyjess.Molecule)
    over._init_columns(extra, np.concatenate([xyz, xyz[:m] + np.array([0.0, 0.0, 90.0])]), "one_beyond")
    assert len(over) == _native.TM_NARROW_STRUCTURE_ATOMS + 1
    table, cands = jess.query_batch([small, over], 2.0, cut[sel], cut[sel], max_candidates=10000, ignore_chain=True,
                                    template_index=idx[sel], count_candidates=True, return_raw=True)
    for s, mol in enumerate((small, over)):
        r = oracle.query_library(helpers.to_oracle_molecule(mol), sub, 2.0, cut[sel], cut[sel], max_candidates=10000,
                                 ignore_chain=True)
        n = compare(table, cands, s, r, sub)
    assert n >= 1


def test_index_pool_overflow_is_reported_and_retried(amy, af, monkeypatch):
    """The candidate index is built without a host round trip; if its pool overflows, the query says so
    (TM_E_OVERFLOW) and the next query of the batch has room.  ENZYMM_AMD_INDEX_POOL starts the pool too small."""
    tpls = [template.Template.load(TPL / p) for p in G["A1"]["templates"]]
    jess = pyjess.Jess(tpls)
    want = jess.query_batch([amy, af], 2.0, 1.7, 1.7, max_candidates=10000, ignore_chain=True, return_raw=True)
    monkeypatch.setenv("ENZYMM_AMD_INDEX_POOL", "256")
    got = jess.query_batch([amy, af], 2.0, 1.7, 1.7, max_candidates=10000, ignore_chain=True, return_raw=True)
    assert got.tobytes() == want.tobytes() and len(want) >= 4
    # without the automatic retry the overflow surfaces as an error, and the same batch then succeeds
    packed = pyjess._packed(jess._templates, jess._device)
    batch = pyjess.make_batch([amy, af], jess._device)
    idx = np.arange(len(tpls), dtype=np.uint32)
    flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
    res = _native.query(packed.native, batch, idx, 2.0, 1.7, 1.7, 10000, flags, retry=False)
    with pytest.raises(_native.NativeError) as err:
        res.wait()
    assert err.value.code == _native.TM_E_OVERFLOW
    res.close()
    res = _native.query(packed.native, batch, idx, 2.0, 1.7, 1.7, 10000, flags, retry=False)
    assert res.hits().tobytes() == want.tobytes()
    res.close()
    batch.close()


def test_edge_cases(amy):
    tpl = template.Template.load(TPL / G["G1"]["template"])
    jess = pyjess.Jess([tpl])
    empty = pyjess.Molecule()
    one_atom = amy[:1]
    waters = amy._subset(amy._cols["residue_name"] == b"HOH")
    out = jess.query_batch([empty, one_atom, waters, amy, empty], 2, 1.5, 1.5, max_candidates=10000)
    assert [len(h) for h in out] == [0, 0, 0, 1, 0]
    assert out[3][0].rmsd == pytest.approx(G["G1"]["rmsd"], abs=5e-8)
    # max_candidates = 0 evaluates nothing and flags the search
    t, c = jess.query_batch([amy], 2, 1.5, 1.5, max_candidates=0, count_candidates=True, return_raw=True)
    assert len(t) == 0 and c[0, 0] == 0x80000000
    # None = unlimited
    assert len(jess.query_batch([amy], 2, 1.5, 1.5, max_candidates=None)[0]) == 1
    # a template with more atoms than the build supports is refused, not mis-searched
    atoms = list(tpl) + [a.copy() for a in tpl]
    for i, a in enumerate(atoms[15:]):
        a.residue_number += 1000
    with pytest.raises(_native.NativeError):
        pyjess.Jess([pyjess.Template(atoms, id="too_big")]).query_batch([amy], 2, 1.5, 1.5)


def test_all_hits_mode_and_thread_pool(amy):
    """best_match=False (pyjess's default) yields every hit within the RMSD threshold in enumeration order;
    and the query entry point is re-entrant: enzymm calls it from ThreadPool workers (jess_run.py:919-921)."""
    from multiprocessing.pool import ThreadPool
    g = G["G1"]
    tpl = template.Template.load(TPL / g["template"])
    otpl = helpers.to_oracle_template(tpl)
    want, _, _ = oracle.query(helpers.to_oracle_molecule(amy), otpl, 2, 1.5, 1.5, max_candidates=10000,
                              best_match=False)
    got = list(pyjess.Jess([tpl]).query(amy, 2, 1.5, 1.5, max_candidates=10000))
    assert len(got) == len(want) == 10
    for h, w in zip(got, want):
        assert list(h._atom_index) == w.atoms.tolist() and abs(h.rmsd - w.rmsd) < RMSD_TOL
    assert min(h.rmsd for h in got) == pytest.approx(g["rmsd"], abs=5e-8)

    def one(_):
        return [round(h.rmsd, 9) for h in pyjess.Jess([tpl]).query(amy, 2, 1.5, 1.5, max_candidates=10000,
                                                                   best_match=True)]
    with ThreadPool(4) as pool:
        results = pool.map(one, range(16))
    assert all(r == results[0] and len(r) == 1 for r in results)


def test_matcher_end_to_end(amy, af):
    """tests/test_jess_run.py:301-357 through the real GPU-backed Jess."""
    a1 = G["A1"]
    tpls = [template.Template.load(TPL / p) for p in a1["templates"]]
    params = {int(k): dict(rmsd=v[0], distance=v[1], max_dynamic_distance=v[2]) for k, v in a1["jess_params"].items()}
    unf = jess_run.Matcher(tpls, jess_params=params, filter_matches=False).run_single(amy)
    fil = jess_run.Matcher(tpls, jess_params=params, filter_matches=True).run_single(amy)
    assert sorted(m.hit.template.pdb_id for m in unf) == a1["unfiltered_pdb_ids"]
    assert sorted(m.hit.template.pdb_id for m in fil) == a1["filtered_pdb_ids"]
    load = lambda d: list(template.load_templates(TPL / d))
    t54 = load("5_residues/results/csa3d_0285") + load("4_residues/results/csa3d_0285")
    r1 = jess_run.Matcher(t54).run([amy, af])
    assert list(r1.keys()) == [amy, af] and len(r1[amy]) == 2 and len(r1[af]) == 2
    af80 = af.conserved(80)
    r2 = jess_run.Matcher(t54, skip_smaller_hits=True).run([amy, af80])
    assert len(r2[amy]) == 1 and len(r2[af80]) == 1
    m3 = jess_run.Matcher(t54 + load("3_residues/results/csa3d_0344"), match_small_templates=True)
    assert len(m3.run([amy])[amy]) == 3
    # the reference's own single-call boundary
    single = jess_run.Matcher._run_jess(amy, t54[:2], 2.0, 2.0, 2.0)
    assert all(isinstance(m, jess_run.Match) for m in single)


def test_run_files_equals_host_loader(tmp_path):
    """Matcher.run_files (GPU ingestion, next batch prepared on a worker thread) gives the matches of
    Matcher.run(load_molecules(paths)), in the same order, with the same TSV rows."""
    import shutil
    paths = []
    for name in ("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb"):
        shutil.copy(helpers.GOLDEN / name, tmp_path / name)
        paths.append(tmp_path / name)
    for i, m in enumerate(synthetic.structures(5)):
        p = tmp_path / f"synthetic_{i}.pdb"
        p.write_text(synthetic.to_pdb(m))
        paths.append(p)
    load = lambda d: list(template.load_templates(TPL / d))
    tpls = load("5_residues/results/csa3d_0285") + load("4_residues/results/csa3d_0285") + \
        load("3_residues/results/csa3d_0344")
    matcher = jess_run.Matcher(tpls, filter_matches=False)
    host = matcher.run(jess_run.load_molecules(paths))
    dev = matcher.run_files(paths, batch_size=3)  # three batches: 3 + 3 + 1 files
    rows = lambda res: [(mol.id, m.hit.template.id, round(m.hit.rmsd, 9), m.hit._atom_index, m.complete)
                        for mol, ms in res.items() for m in ms]
    assert rows(dev) == rows(host) and len(rows(host)) >= 4
    assert [m.dumps() for ms in dev.values() for m in ms] == [m.dumps() for ms in host.values() for m in ms]
    # the rows came from the device's residue counts and the matched atoms' lines alone: no file was parsed whole
    assert all(mol._loaded is None and mol._residue_count is not None for mol in dev)
    buf_dev, buf_host = io.StringIO(), io.StringIO()
    for ms_d, ms_h in zip(dev.values(), host.values()):
        for m_d, m_h in zip(ms_d, ms_h):
            m_d.dump2pdb(buf_dev, transform=True)
            m_h.dump2pdb(buf_host, transform=True)
    assert buf_dev.getvalue() == buf_host.getvalue()


def test_fused_epilogue_matches_python_filter(amy, af):
    """orientation angle and logistic verdict computed in the hit epilogue on the GPU must
    equal what Match.orientation / Match.predicted_correct compute in Python (jess_run.py:298-346, 425-478)."""
    lib = template.load_packed_library()
    matcher = jess_run.Matcher(lib, filter_matches=False)
    mols = [amy, af] + synthetic.structures(60)
    out = matcher.run(mols)
    n = decided = 0
    for mol, matches in out.items():
        for m in matches:
            assert m.hit.orientation == pytest.approx(m.orientation, abs=1e-9)
            if str(m.hit.template.effective_size) in jess_run.Match._logistic_regression_models:
                assert m.hit.predicted is not None and m.hit.predicted == m.predicted_correct
                decided += 1
            else:
                assert m.hit.predicted in (None, True) and m.predicted_correct is True
            n += 1
    assert n > 40 and decided > 10
    # filtered run == Python-filtered unfiltered run
    kept = jess_run.Matcher(lib, filter_matches=True).run(mols)
    for mol in mols:
        want = [id(m.hit.template) for m in out.get(mol, []) if m.predicted_correct]
        assert [id(m.hit.template) for m in kept.get(mol, [])] == want
    # G1's pinned orientation straight from the epilogue
    tpl = template.Template.load(TPL / G["G1"]["template"])
    h = list(pyjess.Jess([tpl]).query(amy, 2, 1.5, 1.5, max_candidates=10000, best_match=True))[0]
    assert h.orientation == pytest.approx(G["G1"]["orientation"], abs=5e-8)


def test_device_pdb_ingestion(amy, af, default_library):
    """PDB text parsed on the GPU must give exactly the batch the host loader gives."""
    from enzymm_amd import ingest
    jess, olib, idx, cut = default_library
    synth = synthetic.structures(12)
    texts = [(helpers.GOLDEN / "1AMY.pdb").read_bytes(), (helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb").read_bytes()]
    texts += [synthetic.to_pdb(m).encode() for m in synth]
    ids = ["1AMY", "AF"] + [m.id for m in synth]
    host_mols = [pyjess.Molecule.loads(t.decode(), id=i) for t, i in zip(texts, ids)]
    assert [len(m) for m in host_mols[:2]] == [3339, 4079]
    for cutoff in (0.0, 80.0):
        pb = ingest.PdbBatch(texts, ids, device=jess._device, conservation_cutoff=cutoff)
        want_mols = [m.conserved(cutoff) if cutoff else m for m in host_mols]
        assert [len(m) for m in pb.molecules] == [len(m) for m in want_mols]
        got = jess.query_batch(pb.molecules, 2.0, cut, cut, max_candidates=10000, ignore_chain=True,
                               template_index=idx, batch=pb, return_raw=True)
        want = jess.query_batch(want_mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True,
                                template_index=idx, return_raw=True)
        assert len(got) == len(want) > 0 and got.tobytes() == want.tobytes()
        # the lazy molecules materialise to the same atoms the host loader produced
        hits = jess.query_batch(pb.molecules, 2.0, cut, cut, max_candidates=10000, ignore_chain=True,
                                template_index=idx, batch=pb)
        h = next(h for per in hits for h in per)
        assert [a.serial for a in h.atoms()] == [h._molecule._atom_list()[i].serial for i in h._atom_index]
        # record offsets point at the kept ATOM/HETATM lines
        offs, counts = pb.native.line_offsets(), pb.native.atom_offsets
        for f in (0, 2):
            first = int(offs[counts[f]])
            assert texts[f][first:first + 6] in (b"ATOM  ", b"HETATM")
        pb.close()
    # the batch driver accepts the device-parsed batch directly
    lib54 = list(template.load_templates(TPL / "5_residues")) + list(template.load_templates(TPL / "4_residues"))
    pb = ingest.PdbBatch(texts[:2], ids[:2], device=jess._device)
    via_device = jess_run.Matcher(lib54).run(pb)
    via_host = jess_run.Matcher(lib54).run(host_mols[:2])
    assert [[m.hit.template.id for m in v] for v in via_device.values()] == \
           [[m.hit.template.id for m in v] for v in via_host.values()]
    assert [m.query_residue_count for m in via_device[pb.molecules[0]]] == \
           [m.query_residue_count for m in via_host[host_mols[0]]]
    # ENDMDL handling: two models, only the first is read unless asked otherwise
    model2 = b"".join(l[:20] + b" B" + l[22:] if l.startswith(b"ATOM") else l for l in texts[2].splitlines(keepends=True))
    two = b"MODEL 1\n" + texts[2] + b"ENDMDL\nMODEL 2\n" + model2 + b"ENDMDL\n"
    assert len(ingest.PdbBatch([two], ["x"]).molecules[0]) == len(synth[0])
    assert len(ingest.PdbBatch([two], ["x"], ignore_endmdl=True).molecules[0]) == 2 * len(synth[0])
    # files the device path cannot lay out (the same residues twice, interleaved residue records) are flattened on
    # the host inside tm_batch_create_from_pdb: same atoms as the host loader (hits: tests/test_gpu_round2.py)
    twice = b"MODEL 1\n" + texts[2] + b"ENDMDL\nMODEL 2\n" + texts[2]
    assert len(ingest.PdbBatch([twice], ["x"], ignore_endmdl=True).molecules[0]) == 2 * len(synth[0])
    lines = texts[2].decode().splitlines(keepends=True)
    shuffled = "".join(lines[:3] + lines[40:43] + lines[3:40] + lines[43:]).encode()
    assert len(ingest.PdbBatch([shuffled], ["shuffled"]).molecules[0]) == len(synth[0])
    with pytest.raises(_native.NativeError):
        ingest.PdbBatch([b"ATOM      1  N   GLY A   1       0.000   x.000   0.000\n"], ["bad"])
    assert len(ingest.PdbBatch([b"", b"END\n"], ["e1", "e2"]).molecules[0]) == 0


def test_packed_library_same_hits(amy, af, tmp_path):
    """querying through the memory-mapped pack == querying through Template objects."""
    from enzymm_amd import library_pack
    lib = template.load_packed_library()
    path = tmp_path / "lib.pack"
    library_pack.write_pack(lib, path)
    P = library_pack.PackedTemplates(path)
    eff = P.effective_size
    idx = np.nonzero(eff >= 3)[0]
    cut = np.array([helpers.DEFAULT_DISTANCE[min(int(e), 8)] for e in eff[idx]])
    mols = [amy, af] + synthetic.structures(6)
    a = pyjess.Jess(P).query_batch(mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True, template_index=idx,
                                   return_raw=True)
    b = pyjess.Jess(lib).query_batch(mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True,
                                     template_index=idx, return_raw=True)
    assert len(a) > 20 and a.tobytes() == b.tobytes()
    hits = pyjess.Jess(P).query_batch([amy], 2.0, cut, cut, max_candidates=10000, ignore_chain=True,
                                      template_index=idx)[0]
    assert all(isinstance(h.template, template.Template) for h in hits)
    assert len(P._cache) == len({id(h.template) for h in hits})  # only templates with hits were materialised


def test_full_batch_properties(default_library):
    """configs[2] size (1000 structures x 6780 templates): properties that need no oracle."""
    jess, olib, idx, cut = default_library
    mols = synthetic.structures(1000)
    batch = pyjess.make_batch(mols, jess._device)
    run = lambda: jess.query_batch(mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True, template_index=idx,
                                   batch=batch, return_raw=True)
    a, b = run(), run()
    assert len(a) > 100 and a.tobytes() == b.tobytes()  # deterministic, independent of scheduling
    # a structure's hits do not depend on what else is in the batch
    pick = [3, 500, 999]
    alone = jess.query_batch([mols[i] for i in pick], 2.0, cut, cut, max_candidates=10000, ignore_chain=True,
                             template_index=idx, return_raw=True)
    for j, i in enumerate(pick):
        x, y = a[a["structure"] == i].copy(), alone[alone["structure"] == j].copy()
        x["structure"] = y["structure"] = 0
        assert x.tobytes() == y.tobytes()
    # every hit is what it claims to be
    tpls = jess._templates
    for h in a[:: max(1, len(a) // 300)]:
        mol, tpl = mols[int(h["structure"])], tpls[int(idx[h["template_sel"]])]
        k = int(h["n_atoms"])
        assert k == len(tpl) and h["rmsd"] <= 2.0
        q = mol._xyz[h["atoms"][:k]]
        p = np.array([[t.x, t.y, t.z] for t in tpl])
        R, T = h["rot"].reshape(3, 3), h["trans"]
        assert abs(np.linalg.det(R) - 1) < 1e-9 and np.abs(R @ R.T - np.eye(3)).max() < 1e-9
        assert np.sqrt((((q @ R.T + T) - p) ** 2).sum(1).mean()) == pytest.approx(h["rmsd"], abs=1e-9)
        res = mol._cols["residue_number"][h["atoms"][:k]]
        groups = [(t.chain_id, t.residue_number) for t in tpl]
        for i in range(k):
            for j in range(i):
                assert (groups[i] == groups[j]) == (res[i] == res[j])
                dq, dt = np.linalg.norm(q[i] - q[j]), np.linalg.norm(p[i] - p[j])
                assert abs(dq - dt) <= cut[h["template_sel"]] + 1e-12
    batch.close()
