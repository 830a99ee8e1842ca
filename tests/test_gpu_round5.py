"""GPU tests added in round 5: the compact hand-over of host atom tables (17 bytes per atom, laid out on
the device by k_flatten) gives the batch the host flattening gave, bit for bit where it can; a template with more
residue-check levels than stage D0 has slots.  Every test calls through the C ABI."""
import numpy as np
import pytest

from tests import helpers
from tests.test_gpu_parity import compare
from enzymm_amd import _native, pyjess, synthetic
from enzymm_amd._vocab import VOCAB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def af():
    return pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb")


@pytest.fixture(scope="module")
def amy():
    return pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb")


def _first_appearance(keys):
    """ordinals of ``keys`` numbered by first appearance (what the library's host encoder assigns)."""
    _, first, inverse = np.unique(keys, return_index=True, return_inverse=True)
    rank = np.empty(len(first), dtype=np.int64)
    rank[np.argsort(first, kind="stable")] = np.arange(len(first))
    return rank[inverse]


def _encoded_batch(molecules, device):
    """The same molecules through tm_batch_create: ids, residue and chain ordinals computed here (numpy), flattened by
    the library's HOST code (batch_from_encoded) and uploaded as 46 bytes per atom -- the path every batch took before
    round 5."""
    res_type, name_id, residue, chain, xyz, offsets = [], [], [], [], [], [0]
    for m in molecules:
        c, n = m._cols, len(m)
        chain16 = c["chain_id"].view(np.uint8).reshape(n, 2).astype(np.int64) if n else np.zeros((0, 2), np.int64)
        code = chain16[:, 0] << 8 | chain16[:, 1]
        icode = c["insertion_code"].view(np.uint8).astype(np.int64) if n else np.zeros(0, np.int64)
        key = (code << 40) | ((c["residue_number"].astype(np.int64) + (1 << 24)) << 8) | icode
        residue.append(_first_appearance(key).astype(np.uint32))
        chain.append((_first_appearance(code) & 0xFF).astype(np.uint8))
        res_type.append(VOCAB.encode_residues(c["residue_name"]))
        name_id.append(VOCAB.encode_names(c["name"]))
        xyz.append(m._xyz.reshape(-1, 3))
        offsets.append(offsets[-1] + n)
    cat = lambda parts, dt: np.concatenate(parts).astype(dt) if parts else np.zeros(0, dt)
    return _native.Batch(device, np.array(offsets, np.uint32), cat(xyz, np.float64), cat(res_type, np.uint8),
                         cat(name_id, np.uint8), cat(residue, np.uint32), cat(chain, np.uint8))


def _scattered(mol, residues=(5, 17)):
    """``mol`` with the side-chain atoms of some residues moved to the end of the file: those residues' records are no
    longer contiguous (the library's encoder has to gather them)."""
    c = mol._cols
    key = c["residue_number"].astype(np.int64)
    numbers = np.unique(key)
    moved = np.zeros(len(mol), dtype=bool)
    for r in residues:
        sel = np.nonzero(key == numbers[r])[0]
        moved[sel[len(sel) // 2:]] = True
    order = np.concatenate([np.nonzero(~moved)[0], np.nonzero(moved)[0]])
    return helpers.subset_molecule(mol, order, id=f"{mol.id}_scattered")


def test_packed_upload_lays_out_the_batch_the_host_flattening_did(amy, af):
    """tm_batch_create_from_structures sends 17 bytes per atom (packed word, chain ordinal, int32 thousandths of an
    Angstrom) and k_flatten writes the batch; tm_batch_create (encoded columns, flattened on the host, 46 bytes per atom)
    is the layout it has to equal: fp64 coordinates bit for bit, packed words, chain ordinals, file-order indices and
    residue tables exactly, centred fp32 coordinates to the rounding of the centroid (exact integer mean against a
    sequential fp64 sum).  Cases: 3-decimal coordinates (exact thousandths), arbitrary doubles (fp64 side array), a
    negative zero (a 3-decimal PDB field, but not (double)0 / 1000: fp64 side array), residues whose records are not
    contiguous (sorted on the host, file-order indices sent along), both at once, an empty molecule, a single atom.
    Then the same search on both batches: byte-identical hit tables."""
    device = pyjess.default_device()
    synth = synthetic.structures(6, start=900)
    # arbitrary doubles (rigid_copies rounds to three decimals, which IS (double)k / 1000: nudged off that grid)
    turned = [helpers.subset_molecule(m, np.arange(len(m)), xyz=m._xyz + 1.234567e-5 * (1 + i), id=f"off_grid_{i}")
              for i, m in enumerate(synthetic.rigid_copies(amy, 2, seed=3))]
    minus_zero = synth[0].copy()
    minus_zero._xyz[7, 1] = -0.0
    empty = helpers.subset_molecule(amy, np.zeros(0, dtype=np.int64), id="empty")
    one = helpers.subset_molecule(amy, np.array([100]), id="one_atom")
    molecules = synth + [amy, af] + turned + [minus_zero, _scattered(synth[1]), _scattered(turned[0]), empty, one]
    for m in molecules:
        m._row = None
    packed = pyjess.make_batch(molecules, device)
    encoded = _encoded_batch(molecules, device)
    got, want = packed.copy_layout(), encoded.copy_layout()
    n = sum(len(m) for m in molecules)
    assert got["xyz"].shape == (n, 3)
    assert got["xyz"].tobytes() == want["xyz"].tobytes()  # (-0.0 included)
    assert np.array_equal(got["chain"], want["chain"])
    assert np.array_equal(got["orig"], want["orig"])
    assert np.array_equal(got["residue_offsets"], want["residue_offsets"])
    assert np.array_equal(got["residue_start"], want["residue_start"])
    assert np.array_equal(got["records"][:, 3].view(np.uint32), want["records"][:, 3].view(np.uint32))
    assert np.abs(got["records"][:, :3] - want["records"][:, :3]).max() < 2e-5
    # the scattered molecules really took the gather: file-order indices are not the identity there
    off = packed.atom_offsets
    k = len(synth) + 2 + len(turned) + 1
    assert not np.array_equal(got["orig"][off[k]:off[k + 1]], np.arange(off[k + 1] - off[k]))
    # 17 bytes per atom + 32 per structure for the exact ones; the four inexact molecules send 24 more per atom
    inexact = sum(len(molecules[i]) for i in (8, 9, 10, 12))
    sent = packed.upload_bytes()
    assert sent == 17 * n + 32 * len(molecules) + 8 * (len(molecules) + 1) + 24 * inexact + 4 * (len(molecules[11]) + len(molecules[12]))
    # the same search on both
    idx, cut = helpers.default_selection()
    lib = pyjess._packed(pyjess.Jess(helpers.product_library_templates())._templates, device)
    flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN | _native.TM_FLAG_COUNT_CANDIDATES
    a = _native.query(lib.native, packed, idx, 2.0, cut, cut, 10000, flags)
    b = _native.query(lib.native, encoded, idx, 2.0, cut, cut, 10000, flags)
    ha, hb = a.hits(), b.hits()
    assert len(ha) > 20 and ha.tobytes() == hb.tobytes()
    assert np.array_equal(a.candidates(), b.candidates())
    for r in (a, b):
        r.close()
    packed.close(); encoded.close()


def test_batch_is_searched_before_its_upload_has_been_waited_for():
    """The constructor returns with its copies queued; a query launched at once on another stream waits for them on the
    device (tm_batch::ready).  Forty batches in a row, each searched the moment it exists, equal the same batches
    created with a host wait (ENZYMM_AMD_SYNC_UPLOAD is read once per process, so the waited-for variant uses
    tm_batch_wait)."""
    device = pyjess.default_device()
    idx, cut = helpers.default_selection()
    every = np.arange(0, len(idx), 3)
    lib = pyjess._packed(pyjess.Jess(helpers.product_library_templates())._templates, device)
    flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
    stream = _native.Stream(device)
    mols = synthetic.structures(240, start=300)
    tables = {}
    for waited in (False, True):
        out = []
        for k in range(40):
            batch = pyjess.make_batch(mols[6 * k: 6 * k + 6], device)
            if waited:
                batch.wait()
            res = _native.query(lib.native, batch, idx[every], 2.0, cut[every], cut[every], 10000, flags, stream=stream.handle,
                                retry=False)
            out.append((batch, res))
        tables[waited] = [r.hits().tobytes() for _, r in out]
        for batch, res in out:
            res.close()
            batch.close()
    assert tables[False] == tables[True]
    assert sum(len(t) for t in tables[True]) > 0
    stream.close()


def test_template_with_more_residue_check_levels_than_d0_has_slots():
    """stage D0 keeps kResChecks = 21 records per staged template; a template of ONE residue with 24 atoms
    has 23 levels that qualify (every atom but its lead).  The extra levels are dropped from the filter (it is a
    necessary condition), not written into the next template's slots.  A 24-atom residue implanted in a synthetic
    structure, templates made of it with 24 / 23 / 22 atoms (one residue each), next to ordinary templates that follow
    them in the staged array: hits equal the oracle's."""
    from oracle import oracle
    rng = np.random.default_rng(5)
    base = synthetic.structures(1, start=77)[0]
    n_lig = 24
    # a compact blob of 24 atoms with distinct names next to the protein
    centre = base._xyz.mean(0) + np.array([3.0, 2.0, 1.0])
    pts = np.round(centre + rng.normal(scale=2.2, size=(n_lig, 3)), 3)
    # (names and residue name from the standard vocabulary: the process-wide registry holds 63 names and is append-only)
    from enzymm_amd._vocab import _STANDARD_ATOMS
    names = [nm.strip() for nm in _STANDARD_ATOMS[:n_lig]]
    lig_cols = dict(name=np.array([nm.encode() for nm in _STANDARD_ATOMS[:n_lig]], dtype="S4"),
                    residue_name=np.full(n_lig, b"TRP", dtype="S3"), chain_id=np.full(n_lig, b" A", dtype="S2"),
                    residue_number=np.full(n_lig, 9001, dtype=np.int32))
    lig = pyjess.Molecule.from_columns(id="lig", xyz=pts, **lig_cols)
    mol = helpers.concat_molecules([base, lig], id="with_ligand")

    def ligand_template(count, ident="lig"):
        atoms = []
        for i in range(count):
            atoms.append(pyjess.TemplateAtom(chain_id="A", residue_number=9001, x=float(pts[i, 0]), y=float(pts[i, 1]),
                                             z=float(pts[i, 2]), residue_names=["TRP"], atom_names=[names[i]],
                                             distance_weight=0.0, match_mode=0))
        return pyjess.Template(atoms, id=ident)

    ordinary = helpers.implant_templates(base, helpers.product_library_templates(), rng, max_templates=12)
    templates = []
    for k, t in enumerate([ligand_template(24, ident="lig24"), ligand_template(23, ident="lig23"),
                           ligand_template(22, ident="lig22")]):
        templates += [t, ordinary[2 * k], ordinary[2 * k + 1]]
    templates += ordinary[6:]
    pyjess.register_vocabulary(templates)
    jess = pyjess.Jess(templates)
    hits = jess.query_batch([mol], 2.0, 1.5, 1.5, max_candidates=10000, ignore_chain=True, best_match=True)[0]
    by_id = {h.template.id: h for h in hits}
    assert {"lig24", "lig23", "lig22"} <= set(by_id)
    lib = oracle.OracleLibrary([helpers.to_oracle_template(t) for t in templates])
    want = oracle.query_library(helpers.to_oracle_molecule(mol), lib, 2.0, 1.5, 1.5, max_candidates=10000, ignore_chain=True)
    assert sorted(by_id) == sorted(templates[t].id for t in np.nonzero(want["hit"])[0])
    for t in np.nonzero(want["hit"])[0]:
        h = by_id[templates[t].id]
        k = len(templates[t])
        assert [a.serial for a in h.atoms()] == [int(mol._cols["serial"][i]) for i in want["atoms"][t][:k]], templates[t].id
        assert abs(h.rmsd - want["rmsd"][t]) < 1e-9


def test_two_bench_processes_share_one_gpu_and_the_host():
    """The closest stand-in a one-GPU box has for the ranks of a node contending for the host: TWO
    `bench.py --gpus 1` processes at the same time on this GPU, each told it is one of two ranks of the node
    (LOCAL_WORLD_SIZE=2: half the host threads each), both through the sharded loop over RCCL, with the streamed passes on
    (the host-side encode of the hand-over runs beside the other process's).  Both print the hit table of the plain run."""
    from concurrent.futures import ThreadPoolExecutor
    from tests.test_gpu_round3 import _bench_line
    common = ["--gpus", "1", "--structures", "192", "--batch", "96", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
              "--no-extras", "--streamed-passes", "2", "--sustained-seconds", "0.5"]
    alone = _bench_line(common, {})
    env = {"ENZYMM_AMD_BENCH_FORCE_SHARDED": "1", "LOCAL_WORLD_SIZE": "2"}
    with ThreadPoolExecutor(2) as pool:
        both = list(pool.map(lambda _: _bench_line(common, env, timeout=420), range(2)))
    for line in both:
        assert line["config"]["hit_table_digest"] == alone["config"]["hit_table_digest"]
        assert line["config"]["hits_total"] == alone["config"]["hits_total"] > 100
        assert line["config"]["host_threads_per_rank"] == max(1, alone["config"]["host_threads_per_rank"] // 2)
        assert line["config"]["per_rank_structures_per_s"] and line["config"]["slowest_rank"] == 0
        assert line["config"]["streamed_structures_per_s"] > 0


# (test_kind_list_beyond_16_bit_positions_is_reported moved to tests/test_gpu_round6.py: since round 6 such a structure is
# left out of the search on its own instead of ending the query)


# ------------------------------------------------------------------------------------------- long structures (round 5)


def test_long_structure_tables_culled_and_plain(af, amy, monkeypatch):
    """Structures too long for the batch's LDS layout (large complexes) are built by a launch of their own: k_tab_gather
    lays the record of every kind-list position out in global memory and gives every 16 entries of a list their bounding
    sphere, the sweeps test the chunks of list B against a row first and lay only the surviving ones over the lanes -- in
    the order of the plain sweep, so the pair tables are the same.  A 12 237-atom complex (three copies of AF-P0DUB6 70 A
    apart: most chunk pairs are culled) alone and inside a batch of ordinary structures (the batch's layout for those, the
    long-structure launch beside it), index built in segments of 4096 atoms: hit tables and candidate counts are equal with
    ENZYMM_AMD_TAB_CULL=0 (plain sweeps, records through the kind lists), the complex equals the oracle on every fifth
    template of the default selection, and the ordinary structures get what they get without it."""
    from oracle import oracle
    from tests.test_gpu_round4 import _copies
    idx, cut = helpers.default_selection()
    every = np.arange(0, len(idx), 5)
    jess = pyjess.Jess(helpers.product_library_templates())
    big = _copies(af, 3, "AFx3")
    big._xyz[:] = np.round(big._xyz + np.random.default_rng(5).normal(0.0, 0.01, size=big._xyz.shape), 3)
    assert 4096 * 2 < len(big) < _native.TM_NARROW_STRUCTURE_ATOMS
    others = [amy] + synthetic.structures(5, start=70)
    q = lambda mols: jess.query_batch(mols, 2.0, cut[every], cut[every], max_candidates=10000, ignore_chain=True,
                                      template_index=idx[every], return_raw=True, count_candidates=True)
    alone, alone_c = q([big])
    mixed, mixed_c = q(others[:3] + [big] + others[3:])
    plain_others, plain_others_c = q(others)
    monkeypatch.setenv("ENZYMM_AMD_TAB_CULL", "0")
    alone0, alone0_c = q([big])
    mixed0, mixed0_c = q(others[:3] + [big] + others[3:])
    monkeypatch.delenv("ENZYMM_AMD_TAB_CULL")
    assert len(alone) >= 1 and alone.tobytes() == alone0.tobytes() and np.array_equal(alone_c, alone0_c)
    assert mixed.tobytes() == mixed0.tobytes() and np.array_equal(mixed_c, mixed0_c)
    # the complex inside the batch = the complex alone; the others = the others without it
    sub = mixed[mixed["structure"] == 3].copy()
    sub["structure"] = 0
    assert sub.tobytes() == alone.tobytes() and np.array_equal(mixed_c[3], alone_c[0])
    rest = mixed[mixed["structure"] != 3].copy()
    rest["structure"] = np.where(rest["structure"] > 3, rest["structure"] - 1, rest["structure"])
    assert rest.tobytes() == plain_others.tobytes() and np.array_equal(np.delete(mixed_c, 3, axis=0), plain_others_c)
    # ... and against the oracle
    otpl = helpers.oracle_library_templates()
    olib = oracle.OracleLibrary([otpl[i] for i in idx[every]])
    want = oracle.query_library(helpers.to_oracle_molecule(big), olib, 2.0, cut[every], cut[every], max_candidates=10000, ignore_chain=True)
    ok = ~want["truncated"].astype(bool)
    assert sorted(int(t) for t in alone["template_sel"] if ok[t]) == np.nonzero(want["hit"] & ok)[0].tolist()
    for h in alone:
        t = int(h["template_sel"])
        if ok[t]:
            k = olib.templates[t].k
            assert np.array_equal(h["atoms"][:k], want["atoms"][t][:k]) and abs(h["rmsd"] - want["rmsd"][t]) < 1e-9, t
    assert np.array_equal(alone_c[0] & 0x7FFFFFFF, want["candidates"])
    # densely: library templates re-seated on residues of the chain hit every copy (dozens of candidates per template)
    implanted = helpers.implant_templates(af, helpers.product_library_templates(), np.random.default_rng(9), max_templates=80)
    dense = pyjess.Jess(implanted)
    dq = lambda: dense.query_batch([amy, big], 2.0, 1.2, 1.2, max_candidates=10000, ignore_chain=True, count_candidates=True, return_raw=True)
    table, dc = dq()
    monkeypatch.setenv("ENZYMM_AMD_TAB_CULL", "0")
    table0, dc0 = dq()
    monkeypatch.delenv("ENZYMM_AMD_TAB_CULL")
    assert table.tobytes() == table0.tobytes() and np.array_equal(dc, dc0)
    dlib = oracle.OracleLibrary([helpers.to_oracle_template(t) for t in implanted])
    dwant = oracle.query_library(helpers.to_oracle_molecule(big), dlib, 2.0, 1.2, 1.2, max_candidates=10000, ignore_chain=True)
    dok = ~dwant["truncated"].astype(bool)
    assert int(dwant["candidates"].sum()) > 500
    assert np.array_equal((dc[1] & 0x7FFFFFFF)[dok], dwant["candidates"][dok])
    rows = table[table["structure"] == 1]
    assert sorted(int(t) for t in rows["template_sel"] if dok[t]) == np.nonzero(dwant["hit"] & dok)[0].tolist()


def test_deep_search_handover_on_a_long_structure(af, monkeypatch):
    """k_deep on a structure in global memory: the sixteen waves of a workgroup hand each other subtrees through the
    workgroup's queue in LDS (survivors of levels below k - 3).  A 12 237-atom complex against implanted templates at a loose
    cutoff (deep subtrees: interchangeable atoms in every residue; thousands of candidates): hit tables and candidate counts
    are equal without hand-over (ENZYMM_AMD_HANDOVER=0), with the shipped rule and with everything handed over that can be
    (=2), and the shipped one equals the oracle (candidate counts of the pairs that were not cut short, hit sets)."""
    from oracle import oracle
    from tests.test_gpu_round4 import _copies
    big = _copies(af, 3, "AFx3")
    big._xyz[:] = np.round(big._xyz + np.random.default_rng(21).normal(0.0, 0.01, size=big._xyz.shape), 3)
    implanted = helpers.implant_templates(af, helpers.product_library_templates(), np.random.default_rng(17), max_templates=60)
    dense = pyjess.Jess(implanted)
    q = lambda: dense.query_batch([big], 2.0, 2.0, 2.0, max_candidates=10000, ignore_chain=True, count_candidates=True, return_raw=True)
    table, cands = q()
    assert int((cands[0] & 0x7FFFFFFF).sum()) > 2000
    for mode in ("0", "2"):
        monkeypatch.setenv("ENZYMM_AMD_HANDOVER", mode)
        t2, c2 = q()
        # (a pair that was cut short keeps whichever 10 000 candidates came first: its best may differ between two runs)
        full = (cands[0] >> 31) == 0
        assert np.array_equal(c2[0], cands[0]), mode
        keep = lambda t: t[np.isin(t["template_sel"], np.nonzero(full)[0])]
        assert keep(t2).tobytes() == keep(table).tobytes(), mode
    monkeypatch.delenv("ENZYMM_AMD_HANDOVER")
    dlib = oracle.OracleLibrary([helpers.to_oracle_template(t) for t in implanted])
    want = oracle.query_library(helpers.to_oracle_molecule(big), dlib, 2.0, 2.0, 2.0, max_candidates=10000, ignore_chain=True)
    ok = ~want["truncated"].astype(bool)
    assert np.array_equal((cands[0] >> 31).astype(bool), want["truncated"].astype(bool))
    assert np.array_equal((cands[0] & 0x7FFFFFFF)[ok], want["candidates"][ok])
    assert sorted(int(t) for t in table["template_sel"] if ok[t]) == np.nonzero(want["hit"] & ok)[0].tolist()
