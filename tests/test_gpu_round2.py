"""GPU parity tests added in round 2: the two BASELINE.json configurations round 1 left untested (configs[3]
sharding, configs[4] large complexes), the reference CLI's remaining vectors (tests/test_cli.py), the TSV through
the HIP path, and the regressions behind the round-1 code review.  Every test calls through the C ABI."""
import io
import os
import warnings

import numpy as np
import pytest

from tests import helpers
from enzymm_amd import _native, distributed, ingest, jess_run, pyjess, report, synthetic, template
from oracle import oracle

pytestmark = pytest.mark.gpu

TPL = helpers.GOLDEN / "templates"
RMSD_TOL = 1e-5


@pytest.fixture(scope="module")
def amy():
    return pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb")


@pytest.fixture(scope="module")
def af():
    return pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb", id="AF-P0DUB6-F1-model_v4")


def compare(table, cands, s, r, lib):
    """Hit rows of structure s vs the oracle's result r: the set of truncated pairs, candidate counts and hits of
    the pairs that were searched to the end (truncation order is the kernel's own), atoms bit-identical."""
    trunc = r["truncated"].astype(bool)
    assert np.array_equal((cands[s] >> 31).astype(bool), trunc)
    assert np.array_equal((cands[s] & 0x7FFFFFFF)[~trunc], r["candidates"][~trunc])
    mine = {int(h["template_sel"]): h for h in table if int(h["structure"]) == s}
    theirs = set(np.nonzero(r["hit"] & ~trunc)[0].tolist())
    assert {t for t in mine if not trunc[t]} == theirs
    for t in theirs:
        k = lib.templates[t].k
        assert np.array_equal(mine[t]["atoms"][:k], r["atoms"][t][:k]), t
        assert abs(mine[t]["rmsd"] - r["rmsd"][t]) < RMSD_TOL
    return len(theirs)


def _every(n):
    """Every n-th template of the default selection (all six size buckets)."""
    idx, cut = helpers.default_selection()
    sel = np.arange(0, len(idx), n)
    ptpl, otpl = helpers.product_library_templates(), helpers.oracle_library_templates()
    return (pyjess.Jess([ptpl[i] for i in idx[sel]]), oracle.OracleLibrary([otpl[i] for i in idx[sel]]))


# ----------------------------------------------------------------------------------------------- configs[4]


@pytest.mark.parametrize("kind", ["complex", "small"])
def test_config5_large_complex(kind):
    """BASELINE.json configs[4]: a >5000-residue multi-chain complex, pLDDT filter conserved(70) applied on the host
    AND by the device reader (min_bfactor), loosened `-j 2 2.5 2.5`, against the oracle.  "complex" (25 k atoms after
    the filter) runs k_search<false, ...> (structure read from L2), "small" (two chains, 450 residues) k_search<true, ...>."""
    mol = synthetic.complex_structure(0, 6000) if kind == "complex" else synthetic.complex_structure(1, 450)
    n_res = len(set(zip(mol._cols["chain_id"].tolist(), mol._cols["residue_number"].tolist())))
    kept = mol.conserved(70)
    if kind == "complex":
        assert n_res > 5000 and 15000 < len(kept) < len(mol) <= _native.TM_MAX_STRUCTURE_ATOMS
    else:
        assert len(kept) < 3500
    jess, olib = _every(17 if kind == "complex" else 9)
    om = helpers.to_oracle_molecule(kept)
    pdb = ingest.PdbBatch([synthetic.to_pdb(mol).encode()], [mol.id], device=jess._device, conservation_cutoff=70.0)
    assert len(pdb.molecules[0]) == len(kept)
    total = 0
    for cap in (10000, 40):  # the reference's cap, and one low enough that searches are cut short
        r = oracle.query_library(om, olib, 2.0, 2.5, 2.5, max_candidates=cap, ignore_chain=True)
        host, hc = jess.query_batch([kept], 2.0, 2.5, 2.5, max_candidates=cap, ignore_chain=True,
                                    count_candidates=True, return_raw=True)
        dev, dc = jess.query_batch(pdb.molecules, 2.0, 2.5, 2.5, max_candidates=cap, ignore_chain=True,
                                   count_candidates=True, return_raw=True, batch=pdb)
        total += compare(host, hc, 0, r, olib)
        assert np.array_equal(hc, dc)
        ok = ~r["truncated"].astype(bool)[host["template_sel"]]
        assert host[ok].tobytes() == dev[ok].tobytes()  # host filter + upload == device reader with min_bfactor
        if cap == 40:
            assert r["truncated"].sum() > 0
        else:
            assert r["candidates"].sum() > (1000 if kind == "complex" else 50)
    assert total > 0
    pdb.close()


# ----------------------------------------------------------------------------------------------- configs[3]


def test_shard_consistency_on_one_gpu(amy, af):
    """BASELINE.json configs[3] shards the batch over ranks: the shards of `shard_range`, queried separately and
    passed through the gather's merge, must be byte-equal to the unsharded hit table."""
    idx, cut = helpers.default_selection()
    jess = pyjess.Jess(helpers.product_library_templates())
    mols = [amy] + synthetic.structures(40) + [af] + synthetic.structures(23, start=100)
    q = lambda ms: jess.query_batch(ms, 2.0, cut, cut, max_candidates=10000, ignore_chain=True, template_index=idx,
                                    return_raw=True)
    whole = q(mols)
    assert len(whole) > 60
    for world in (2, 3, 8):
        parts = []
        for rank in range(world):
            lo, hi = distributed.shard_range(len(mols), rank, world)
            parts.append(distributed.gather_tables(q(mols[lo:hi]), lo))  # no process group: offsets only
        merged = distributed.merge_tables(parts)
        assert merged.tobytes() == whole.tobytes()
    # balanced by atoms for mixed sizes: still contiguous, still the same table
    ranges = distributed.shard_by_atoms([len(m) for m in mols], 4)
    assert ranges[0][0] == 0 and ranges[-1][1] == len(mols) and all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))
    merged = distributed.merge_tables([distributed.gather_tables(q(mols[lo:hi]), lo) for lo, hi in ranges])
    assert merged.tobytes() == whole.tobytes()


# ----------------------------------------------------------------------------------------------- reference CLI vectors


def test_cli_vector_a4_full_library_unfiltered(amy):
    """tests/test_cli.py:50-60,121 (`-j 2 0.5 0.5 --unfiltered`, the whole shipped library): the reference only
    asserts exit code 0, i.e. that every emitted hit is residue-consistent and finite -- Match.atom_triplets raises
    ValueError otherwise (jess_run.py:364-371) -- and that the TSV can be written.  All 7607 templates load; the
    6780 of effective size >= 3 are searched (match_small_templates is off by default)."""
    lib = template.load_packed_library()
    assert len(lib) == 7607
    params = {s: {"rmsd": 2.0, "distance": 0.5, "max_dynamic_distance": 0.5} for s in range(3, 9)}  # _cli.py:170-214
    matcher = jess_run.Matcher(lib, jess_params=params, filter_matches=False)
    out = matcher.run([amy])
    matches = out[amy]
    assert len(matches) >= 1
    for m in matches:
        assert np.isfinite(m.orientation) and np.isfinite(m.hit.rmsd) and m.hit.rmsd <= 2.0
        assert len(m.matched_residues) == m.hit.template.dimension  # computed without ValueError
    buf = io.StringIO()
    report.write_tsv(out, buf, predict_correctness=False)  # --unfiltered: no verdict column (_cli.py:270-280)
    assert buf.getvalue().count("\n") == len(matches) + 2  # version line, header, rows
    # tests/test_cli.py:131-132: the same arguments WITHOUT --unfiltered raise KeyError (no logistic models for a
    # pairwise distance of 0.5) -- which they only do if a template of effective size 3 or 4 hit 1AMY at 0.5 A
    assert any(m.hit.template.effective_size in (3, 4) for m in matches)
    with pytest.raises(KeyError):
        jess_run.Matcher(lib, jess_params=params, filter_matches=True).run([amy])


def test_cli_vector_dynamic_distance_and_conservation(tmp_path):
    """tests/test_cli.py:62-75 (`-l list -j 5 1.1 1.5 -c 2`): rmsd 5, distance 1.1, max_dynamic_distance 1.5 > distance
    (distance weights take part), conservation cutoff 2 (a no-op filter in the reference, jess_run.py:541-545, and
    a no-op here too because every B-factor is >= 2), filtered run.  The reference asserts exit code 0; here the
    hits are also compared with the oracle.  Filtering at distance 1.1 has logistic models (1.1 is one of the 14
    tabulated distances), so no KeyError."""
    import shutil
    paths = []
    for name in ("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb"):
        shutil.copy(helpers.GOLDEN / name, tmp_path / name)
        paths.append(tmp_path / name)
    mols = jess_run.load_molecules(paths, conservation_cutoff=2)
    assert [len(m) for m in mols] == [3339, 4079]
    idx, _ = helpers.default_selection()
    sel = idx[::11]
    ptpl, otpl = helpers.product_library_templates(), helpers.oracle_library_templates()
    jess, olib = pyjess.Jess([ptpl[i] for i in sel]), oracle.OracleLibrary([otpl[i] for i in sel])
    table, cands = jess.query_batch(mols, 5.0, 1.1, 1.5, max_candidates=10000, ignore_chain=True, count_candidates=True,
                                    return_raw=True)
    n = 0
    for s, mol in enumerate(mols):
        r = oracle.query_library(helpers.to_oracle_molecule(mol), olib, 5.0, 1.1, 1.5, max_candidates=10000,
                                 ignore_chain=True)
        n += compare(table, cands, s, r, olib)
    assert n > 5
    # the reference's exact template directory for this argument set (5_residues/results/csa3d_0285), HIP vs oracle
    five = list(template.load_templates(TPL / "5_residues" / "results" / "csa3d_0285"))
    params5 = {s: {"rmsd": 5.0, "distance": 1.1, "max_dynamic_distance": 1.5} for s in range(3, 9)}
    rows = lambda res: [(mol.id, m.hit.template.id, list(m.hit._atom_index), round(m.hit.rmsd, 6))
                        for mol, ms in res.items() for m in ms]
    got = rows(jess_run.Matcher(five, jess_params=params5, conservation_cutoff=2).run(mols))
    assert got == rows(jess_run.Matcher(five, jess_params=params5, conservation_cutoff=2,
                                        jess_factory=helpers.OracleJess).run(mols)) and len(got) >= 2
    lib = [t for t in template.load_packed_library() if t.effective_size in (3, 4, 5)][::7]
    params = {s: {"rmsd": 5.0, "distance": 1.1, "max_dynamic_distance": 1.5} for s in range(3, 9)}
    filtered = jess_run.Matcher(lib, jess_params=params, conservation_cutoff=2).run(mols)
    unfiltered = jess_run.Matcher(lib, jess_params=params, conservation_cutoff=2, filter_matches=False).run(mols)
    for mol in mols:
        want = [id(m.hit.template) for m in unfiltered.get(mol, []) if m.predicted_correct]
        assert [id(m.hit.template) for m in filtered.get(mol, [])] == want


def test_results_tsv_row_through_the_gpu(amy, golden_dir):
    """_cli.py:270-280 / tests/test_jess_run.py:147-160: the reference's results.tsv row for G1, produced by the HIP
    path.  Columns the build cannot reproduce are listed explicitly, not skipped silently: `log_evalue` (Jess's
    E-value model is in absent C source) and the five AnnotatedTemplate columns (their data file is among the
    reference's .MISSING_LARGE_BLOBS)."""
    import json
    g = json.loads((golden_dir / "golden.json").read_text())["G1"]
    tpl = template.Template.load(TPL / g["template"])
    hit = list(pyjess.Jess([tpl]).query(amy, 2, 1.5, 1.5, max_candidates=10000, best_match=True))[0]
    m = jess_run.Match(hit=hit, pairwise_distance=1.5, complete=True, index=0)
    buf = io.StringIO()
    m.dump(buf, header=True)
    got_header, got_row = [l.split("\t") for l in buf.getvalue().splitlines()]
    want_header, want_row = [l.split("\t") for l in (golden_dir / "results.tsv").read_text().splitlines()[:2]]
    assert got_header == want_header
    unreproducible = {"log_evalue", "number_of_mutated_residues", "number_of_side_chain_residues_(template,reference)",
                      "number_of_metal_ligands_(template,reference)", "number_of_ptm_residues_(template, reference)",
                      "total_reference_residues"}
    differing = {name for name, a, b in zip(want_header, got_row, want_row) if a != b}
    assert differing <= unreproducible, differing
    assert len(want_header) - len(unreproducible) == 24  # 24 of the 30 columns are checked byte for byte
    assert "log_evalue" in differing  # the day this fails, the E-value has been implemented: drop the xfail below


@pytest.mark.xfail(strict=True, reason="Hit.log_evalue: Jess's E-value model lives in absent third-party C source "
                                        "(pyjess ~= 0.5); two known answers cannot determine it, it is reported as NaN")
@pytest.mark.parametrize("key", ["G1", "G2"])
def test_log_evalue_known_answers(amy, golden, key):
    """tests/test_jess_run.py:78 (-3.08424478) and :134 (1.8517964201): kept visible as strict xfails."""
    g = golden[key]
    tpl = template.Template.load(TPL / g["template"])
    hit = list(pyjess.Jess([tpl]).query(amy, g["rmsd_threshold"], g["distance_cutoff"], g["max_dynamic_distance"],
                                        max_candidates=g["max_candidates"], best_match=True))[0]
    assert hit.log_evalue == pytest.approx(g["log_evalue"], abs=1e-7)


def test_completeness_with_multi_member_cluster(amy, af):
    """jess_run.py:738-783 on real hits (shared vector with tests/test_host_model.py, here through the HIP path)."""
    helpers.check_completeness_vector(jess_run.Matcher, pyjess.Jess, amy, af)


# ----------------------------------------------------------------------------------------------- regressions


def test_pipelined_queries_inherit_index_overflow(amy, af, monkeypatch, golden):
    """A second tm_query issued before the first result was waited for reuses the candidate index the first one
    built; if that build overflowed its pool, BOTH must report TM_E_OVERFLOW (the second used to return TM_OK with
    hits missing)."""
    tpls = [template.Template.load(TPL / p) for p in golden["A1"]["templates"]]
    jess = pyjess.Jess(tpls)
    want = jess.query_batch([amy, af], 2.0, 1.7, 1.7, max_candidates=10000, ignore_chain=True, return_raw=True)
    monkeypatch.setenv("ENZYMM_AMD_INDEX_POOL", "256")
    packed = pyjess._packed(jess._templates, jess._device)
    batch = pyjess.make_batch([amy, af], jess._device)
    idx = np.arange(len(tpls), dtype=np.uint32)
    flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
    first = _native.query(packed.native, batch, idx, 2.0, 1.7, 1.7, 10000, flags, retry=False)
    second = _native.query(packed.native, batch, idx, 2.0, 1.7, 1.7, 10000, flags, retry=False)  # no rebuild
    for res in (second, first):
        with pytest.raises(_native.NativeError) as err:
            res.wait()
        assert err.value.code == _native.TM_E_OVERFLOW
        res.close()
    third = _native.query(packed.native, batch, idx, 2.0, 1.7, 1.7, 10000, flags, retry=False)
    assert third.hits().tobytes() == want.tobytes() and len(want) >= 4
    third.close()
    batch.close()


def test_pdb_batch_with_files_the_device_path_cannot_lay_out(amy, tmp_path):
    """One file whose residue records are interleaved (a residue's atoms continue after other residues) and one
    with the same residues twice, inside a batch of ordinary files: tm_batch_create_from_pdb flattens those on the
    host and keeps the batch; hits equal the host loader's for every file, and Matcher.run_files works on them."""
    synth = synthetic.structures(3)
    texts = [synthetic.to_pdb(m).encode() for m in synth]
    lines = texts[1].decode().splitlines(keepends=True)
    interleaved = "".join(lines[:3] + lines[40:43] + lines[3:40] + lines[43:]).encode()
    amy_lines = (helpers.GOLDEN / "1AMY.pdb").read_text().splitlines(keepends=True)
    atoms = [i for i, l in enumerate(amy_lines) if l.startswith("ATOM")]
    moved = [amy_lines[i] for i in atoms[1500:1504]]  # four records of one stretch re-appended at the end as a tail
    amy_split = "".join(l for i, l in enumerate(amy_lines) if i not in atoms[1500:1504] and not l.startswith("END"))
    amy_split = (amy_split + "".join(moved) + "END\n").encode()
    twice = b"MODEL 1\n" + texts[2] + b"ENDMDL\nMODEL 2\n" + texts[2]
    batch_texts = [texts[0], interleaved, amy_split, texts[2]]
    ids = ["s0", "interleaved", "amy_split", "s2"]
    idx, cut = helpers.default_selection()
    jess = pyjess.Jess(helpers.product_library_templates())
    host_mols = [pyjess.Molecule.loads(t.decode(), id=i) for t, i in zip(batch_texts, ids)]
    q = lambda mols, **kw: jess.query_batch(mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True,
                                            template_index=idx, return_raw=True, **kw)
    for cutoff in (0.0, 60.0):
        pb = ingest.PdbBatch(batch_texts, ids, device=jess._device, conservation_cutoff=cutoff)
        want_mols = [m.conserved(cutoff) if cutoff else m for m in host_mols]
        assert [len(m) for m in pb.molecules] == [len(m) for m in want_mols]
        got, want = q(pb.molecules, batch=pb), q(want_mols)
        assert got.tobytes() == want.tobytes() and len(want) > (0 if cutoff else 3)
        assert cutoff or {int(s) for s in want["structure"]} >= {1, 2}  # the patched files have hits themselves
        counts = pb.native.residue_counts()
        for f, m in enumerate(want_mols):
            names = np.char.strip(m._cols["residue_name"].astype("U3"))
            keep = np.isin(names, list(jess_run._COUNTED_RESIDUES))
            assert counts[f] == len(set(m._cols["residue_number"][keep].tolist()))
        pb.close()
    pb = ingest.PdbBatch([twice, texts[0]], ["twice", "s0"], device=jess._device, ignore_endmdl=True)
    both = pyjess.Molecule.loads(twice.decode(), id="twice", ignore_endmdl=True)
    assert len(pb.molecules[0]) == len(both) == 2 * len(synth[2])
    assert q(pb.molecules, batch=pb).tobytes() == q([both, host_mols[0]]).tobytes()
    pb.close()
    # end to end through Matcher.run_files
    paths = []
    for name, text in zip(ids, batch_texts):
        p = tmp_path / f"{name}.pdb"
        p.write_bytes(text)
        paths.append(p)
    tpls = list(template.load_templates(TPL / "5_residues")) + list(template.load_templates(TPL / "4_residues"))
    matcher = jess_run.Matcher(tpls, filter_matches=False)
    rows = lambda res: [(mol.id, m.hit.template.id, round(m.hit.rmsd, 9), m.dumps()) for mol, ms in res.items() for m in ms]
    assert rows(matcher.run_files(paths)) == rows(matcher.run(jess_run.load_molecules(paths)))


def test_truncated_matches_are_flagged(amy):
    """max_candidates reached: which assignments were visited before the cut is this kernel's own order, so the best
    hit of such a pair may differ from PyJess's.  Matcher says so (a warning under warn=True, Match.truncated)."""
    idx = helpers.default_selection()[0][::29]
    lib = [helpers.product_library_templates()[i] for i in idx]
    tlib = [t for t in template.load_packed_library() if t.effective_size >= 3][::29]
    params = {s: {"rmsd": 3.0, "distance": 2.5, "max_dynamic_distance": 2.5} for s in range(3, 9)}
    matcher = jess_run.Matcher(tlib, jess_params=params, filter_matches=False, warn=False)
    matcher.MAX_CANDIDATES = 30
    with warnings.catch_warnings(record=True) as seen:
        warnings.simplefilter("always")
        matcher.warn = True
        out = matcher.run([amy])
    flagged = [m for m in out[amy] if m.truncated]
    assert flagged and any("max_candidates" in str(w.message) for w in seen)
    assert all(m.hit.truncated for m in flagged) and not all(m.truncated for m in out[amy])
    del lib


@pytest.mark.parametrize("options", [dict(), dict(filter_matches=False), dict(skip_smaller_hits=True),
                                     dict(filter_matches=False, skip_smaller_hits=True, match_small_templates=True)])
def test_column_replay_equals_object_replay(amy, af, options):
    """Matcher replays the reference's per-size loop (jess_run.py:930-988) on whole columns of the hit table and
    builds Match objects lazily; the object-by-object replay it replaced must give the same molecules in the same
    order with the same matches, completeness flags, distances and verdicts."""
    lib = template.load_packed_library()
    mols = [amy] + synthetic.structures(30) + [af, pyjess.Molecule()] + synthetic.structures(10, start=500)
    fast = jess_run.Matcher(lib, **options)
    slow = jess_run.Matcher(lib, **options)
    slow._collect_table = lambda *a, **k: None  # forces the object-by-object path
    a, b = fast.run(mols), slow.run(mols)
    assert type(next(iter(a.values()))).__name__ == "_LazyMatchList" and type(next(iter(b.values()))) is list
    assert [len(v) for v in a.values()] == [len(v) for v in b.values()]  # len() without materialising
    row = lambda m: (m.hit.template.id, m.complete, m.pairwise_distance, m.hit.rmsd, m.hit._atom_index,
                     m.hit.predicted, m.hit.truncated, m.hit.n_candidates)
    assert [mol.id for mol in a] == [mol.id for mol in b] and len(a) > 5
    for (ma, la), (mb, lb) in zip(a.items(), b.items()):
        assert ma is mb and [row(m) for m in la] == [row(m) for m in lb]
    assert sum(len(v) for v in a.values()) > 10
    if not options.get("filter_matches", True):
        assert any(not m.complete for v in a.values() for m in v) and any(m.complete for v in a.values() for m in v)


def test_device_hit_table_reaches_torch_without_a_host_copy(amy, af):
    """The RCCL leg of the multi-GPU gather (distributed.gather_result) sends rows straight from the device hit table
    (tm_result_device_hits).  What can be checked on one GPU: the table is visible to torch through
    __cuda_array_interface__, holds exactly the rows tm_result_copy_hits returns, and the single-rank path of
    gather_result gives the globally indexed table."""
    import torch
    idx, cut = helpers.default_selection()
    jess = pyjess.Jess(helpers.product_library_templates())
    packed = pyjess._packed(jess._templates, jess._device)
    batch = pyjess.make_batch([amy, af] + synthetic.structures(5), jess._device)
    flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
    res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags)
    n = res.hit_count()
    width = _native.HIT_DTYPE.itemsize
    rows = torch.as_tensor(distributed._DeviceBytes(res.device_hits_ptr(), n * width), device=f"cuda:{jess._device}")
    assert rows.is_cuda and rows.numel() == n * width
    raw = rows.cpu().numpy().view(_native.HIT_DTYPE)
    want = res.hits()
    order = np.lexsort((raw["template_sel"], raw["structure"]))
    assert n > 20 and raw[order].tobytes() == want.tobytes()
    shifted = distributed.gather_result(res, n, 1000)
    assert (shifted["structure"] == want["structure"] + 1000).all() and len(shifted) == n
    res.close()
    batch.close()


@pytest.mark.parametrize("options", [dict(), dict(filter_matches=False, skip_smaller_hits=True)])
def test_run_in_batches_equals_one_pass(amy, af, tmp_path, options):
    """A long molecule list goes through the GPU in pipelined batches (Matcher.run, batch_size) and so do files
    (Matcher.run_files); both must give the result of ONE pass over everything -- the same molecules in the
    reference's insertion order (jess_run.py:930-988: first the molecules that enter in the largest size bucket, input
    order within a bucket), with the same matches."""
    lib = template.load_packed_library()
    matcher = jess_run.Matcher(lib, **options)
    mols = synthetic.structures(12) + [amy, pyjess.Molecule()] + synthetic.structures(14, start=700) + [af] + \
        synthetic.structures(9, start=300)
    row = lambda m: (m.hit.template.id, m.complete, m.pairwise_distance, m.hit.rmsd, m.hit._atom_index,
                     m.hit.predicted, m.hit.truncated)
    whole = matcher.run(mols)
    first = [ms[0].hit.template.effective_size for ms in whole.values()]
    assert len(set(min(8, s) for s in first)) >= 2  # molecules enter in different buckets: the order is not input order
    for size in (1, 7, 16):
        parts = matcher.run(mols, batch_size=size)
        assert [m.id for m in parts] == [m.id for m in whole], size
        for (ma, la), (mb, lb) in zip(parts.items(), whole.items()):
            assert ma is mb and [row(m) for m in la] == [row(m) for m in lb]
    # the same through files
    paths = []
    for i, m in enumerate(mols):
        if len(m):
            p = tmp_path / f"m{i:03d}.pdb"
            p.write_text(synthetic.to_pdb(m))
            paths.append(p)
    host = matcher.run(jess_run.load_molecules(paths))
    dev = matcher.run_files(paths, batch_size=5)
    assert [m.id for m in dev] == [m.id for m in host] and len(host) > 10
    key = lambda m: (m.hit.template.id, m.complete, round(m.hit.rmsd, 9), m.hit._atom_index)
    for la, lb in zip(dev.values(), host.values()):
        assert [key(m) for m in la] == [key(m) for m in lb]


def test_first_query_of_a_small_batch_does_not_overflow():
    """The pair-table pools are sized before the build runs.  A batch too small to fill the GPU has every structure's
    tables dealt to several workgroups, each of whose waves ends on a partly used chunk of entries: the reserve must
    follow (a 64-structure batch used to overflow on its first query and silently run twice through the retry)."""
    idx, cut = helpers.default_selection()
    jess = pyjess.Jess(helpers.product_library_templates())
    packed = pyjess._packed(jess._templates, jess._device)
    flags = _native.TM_FLAG_BEST_MATCH | _native.TM_FLAG_IGNORE_CHAIN
    cases = [synthetic.structures(n, start=40 * n) for n in (1, 7, 64, 300)]
    cases.append([synthetic.complex_structure(2, 6000)])
    cases.append([synthetic.complex_structure(3, 6000)] + synthetic.structures(5))
    for mols in cases:
        batch = pyjess.make_batch(mols, jess._device)
        res = _native.query(packed.native, batch, idx, 2.0, cut, cut, 10000, flags, retry=False)
        res.wait()  # raises NativeError(-5) when a pool was too small
        assert res.hit_count() >= 0
        res.close()
        batch.close()


@pytest.mark.timeout(240)
def test_rccl_gather_world_size_1():
    """The RCCL leg of the multi-GPU gather (distributed.gather_result: counts by all_gather, rows by a padded gather,
    both from device memory) on the one GPU there is: a world-size-1 "nccl" group with the collectives forced.  Run in
    a child process -- a process group must not outlive the test in the pytest process."""
    import subprocess
    import sys
    import socket
    with socket.socket() as sock:  # (a port nobody holds: a fixed one may still be in TIME_WAIT from an earlier run)
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    out = helpers.run_child([sys.executable, str(helpers.ROOT / "tests" / "gpu_rccl_world1.py")], env, timeout=300)
    assert out.returncode == 0 and "RCCL world-size-1 gather ok" in out.stdout, out.stdout[-2000:] + out.stderr[-2000:]


def test_kind_lists_partly_outside_lds():
    """k_search keeps a structure's atom records in LDS up to about 6 300 atoms per structure, its kind lists only while
    they fit beside them (up to about 4 700 atoms): a batch whose largest structure lies in between runs the search
    body that reads kind lists from global memory for that structure and the LDS one for the small ones."""
    big = synthetic.complex_structure(5, 700)
    big = big._subset(np.arange(5500))
    mols = [synthetic.structure(3), big, synthetic.structure(4)]
    jess, olib = _every(13)
    table, cands = jess.query_batch(mols, 2.0, 1.5, 1.5, max_candidates=10000, ignore_chain=True, count_candidates=True,
                                    return_raw=True)
    total = 0
    for s, mol in enumerate(mols):
        r = oracle.query_library(helpers.to_oracle_molecule(mol), olib, 2.0, 1.5, 1.5, max_candidates=10000,
                                 ignore_chain=True)
        total += compare(table, cands, s, r, olib)
    assert total > 3
    # chains are compared on the other kernel variant
    table_c = jess.query_batch(mols, 2.0, 1.5, 1.5, max_candidates=10000, ignore_chain=False, return_raw=True)
    want = sum(int((oracle.query_library(helpers.to_oracle_molecule(m), olib, 2.0, 1.5, 1.5, max_candidates=10000,
                                         ignore_chain=False)["hit"]).sum()) for m in mols)
    assert len(table_c) == want


def test_shared_table_build_equals_one_wave_per_table(amy, af, monkeypatch):
    """k_tab_build builds a table of more than 256 Ki position pairs with all waves of its workgroup (row blocks, counts
    summed in LDS, one allocation).  Large complexes take that path with their lead records read through L2
    (test_config5_large_complex, test_kind_lists_partly_outside_lds); with the limit lowered (ENZYMM_AMD_TAB_SHARED)
    ordinary structures take it with the records in LDS -- the hit tables and candidate counts must not change, and
    they are checked against the oracle for one structure."""
    idx, cut = helpers.default_selection()
    jess = pyjess.Jess(helpers.product_library_templates())
    mols = [amy] + synthetic.structures(20, start=60) + [af]
    q = lambda: jess.query_batch(mols, 2.0, cut, cut, max_candidates=10000, ignore_chain=True, template_index=idx,
                                 count_candidates=True, return_raw=True)
    want, want_c = q()
    for limit in ("1", "900", "20000"):  # every table / most / the larger ones
        monkeypatch.setenv("ENZYMM_AMD_TAB_SHARED", limit)
        got, got_c = q()
        assert got.tobytes() == want.tobytes() and np.array_equal(got_c, want_c), limit
    jess9, olib = _every(9)
    table, cands = jess9.query_batch([amy], 2.0, 1.5, 1.5, max_candidates=10000, ignore_chain=True,
                                     count_candidates=True, return_raw=True)
    r = oracle.query_library(helpers.to_oracle_molecule(amy), olib, 2.0, 1.5, 1.5, max_candidates=10000, ignore_chain=True)
    assert compare(table, cands, 0, r, olib) > 0
    assert len(want) > 30
