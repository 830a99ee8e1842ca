"""The N>1 path on CPU: structure sharding + variable-length hit-table gather over gloo, world size 2.
The per-rank compute is the oracle (tests may use it); on the GPU box the same code runs over RCCL."""
import os
import socket

import numpy as np
import pytest

from tests import helpers
from enzymm_amd import _native, distributed


def test_shard_range_partitions():
    for n in (0, 1, 7, 1000, 100003):
        for world in (1, 2, 3, 8):
            spans = [distributed.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        distributed.shard_range(10, 2, 2)
    spans = distributed.shard_by_atoms([100, 100, 5000, 100, 100, 4000], 2)
    assert spans[0][0] == 0 and spans[-1][1] == 6 and spans[0][1] == spans[1][0]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_table(molecules, lo):
    """Hit table (HIT_DTYPE) of molecules against a few templates, computed with the oracle."""
    from enzymm_amd import template
    tdir = helpers.GOLDEN / "templates"
    tpls = list(template.load_templates(tdir / "5_residues")) + list(template.load_templates(tdir / "4_residues"))
    tpls.sort(key=lambda t: t.template_id_string + str(t.cluster))
    per_mol = helpers.OracleJess(tpls).query_batch(molecules, 2.0, 2.0, 2.0, max_candidates=10000,
                                                   ignore_chain=True, best_match=True)
    rows = []
    for s, hits in enumerate(per_mol):
        for h in hits:
            row = np.zeros((), dtype=_native.HIT_DTYPE)
            row["structure"], row["template_sel"] = s, tpls.index(h.template)
            row["rmsd"], row["rot"], row["trans"] = h.rmsd, h._rot.ravel(), h._trans
            row["n_atoms"] = len(h._atom_index)
            row["atoms"][: len(h._atom_index)] = h._atom_index
            rows.append(row)
    return np.array(rows, dtype=_native.HIT_DTYPE) if rows else np.zeros(0, dtype=_native.HIT_DTYPE)


def _worker(rank, world, port, out_path):
    import torch.distributed as dist
    from enzymm_amd import pyjess, synthetic
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mols = [pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb"),
                pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb")] + synthetic.structures(1)
        lo, hi = distributed.shard_range(len(mols), rank, world)
        local = _oracle_table(mols[lo:hi], lo)
        table = distributed.gather_tables(local, lo, device="cpu")
        if rank == 0:
            np.save(out_path, table)
        else:
            assert table is None
    finally:
        dist.destroy_process_group()


def test_gather_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    from enzymm_amd import pyjess, synthetic
    out = tmp_path / "table.npy"
    mp.spawn(_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    got = np.load(out)
    mols = [pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb"),
            pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb")] + synthetic.structures(1)
    want = _oracle_table(mols, 0)
    assert len(got) == len(want) > 0
    assert got.tobytes() == want.tobytes()  # same table as the unsharded run, globally indexed and sorted


def test_gather_without_process_group_is_identity():
    local = np.zeros(3, dtype=_native.HIT_DTYPE)
    local["structure"] = [0, 1, 1]
    out = distributed.gather_tables(local, 10)
    assert out["structure"].tolist() == [10, 11, 11]


def _shard_files(tmp_path):
    """The reference's two test structures + three synthetic ones as files, and the A2 template set."""
    import shutil
    from enzymm_amd import synthetic, template
    paths = []
    for name in ("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb"):
        shutil.copy(helpers.GOLDEN / name, tmp_path / name)
        paths.append(tmp_path / name)
    for i, m in enumerate(synthetic.structures(3)):
        p = tmp_path / f"synthetic_{i}.pdb"
        p.write_text(synthetic.to_pdb(m))
        paths.append(p)
    tdir = helpers.GOLDEN / "templates"
    tpls = list(template.load_templates(tdir / "5_residues")) + list(template.load_templates(tdir / "4_residues")) + \
        list(template.load_templates(tdir / "3_residues" / "results" / "csa3d_0285"))
    tpls.sort(key=lambda t: (t.effective_size, t.template_id_string, str(t.cluster)))
    return paths, tpls


def _file_worker(rank, world, port, tmp, out_path):
    import torch.distributed as dist
    from pathlib import Path
    from enzymm_amd import jess_run
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        paths, tpls = _shard_files(Path(tmp))
        matcher = jess_run.Matcher(tpls, filter_matches=False, jess_factory=helpers.OracleJess)
        local, lo, hi = distributed.run_files_sharded(matcher, paths, device_ingest=False)
        assert (lo, hi) == distributed.shard_range(len(paths), rank, world)
        text = distributed.gather_report(local)
        if rank == 0:
            Path(out_path).write_text(text)
        else:
            assert text is None
    finally:
        dist.destroy_process_group()


def test_sharded_file_run_world_size_2(tmp_path):
    """The multi-GPU file path (one process per shard of the path list, reports merged on rank 0 -- the analogue of
    /root/reference/nextflow/template_matcher.nf:43-78) with world size 2 over gloo: the merged TSV equals the TSV
    of one process over all files, byte for byte."""
    import io
    import torch.multiprocessing as mp
    from enzymm_amd import jess_run, report
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    out = tmp_path / "merged.tsv"
    mp.spawn(_file_worker, args=(2, _free_port(), str(tmp_path / "a"), str(out)), nprocs=2, join=True)
    paths, tpls = _shard_files(tmp_path / "a")  # same files (already written by the workers)
    whole = jess_run.Matcher(tpls, filter_matches=False, jess_factory=helpers.OracleJess).run(jess_run.load_molecules(paths))
    buf = io.StringIO()
    rows = report.write_tsv(whole, buf)
    got, want = out.read_text().splitlines(), buf.getvalue().splitlines()
    assert rows >= 4 and got[:2] == want[:2] and sorted(got[2:]) == sorted(want[2:])  # same rows under one header
    # molecule order is per shard (as with the reference's per-batch TSVs); inside a molecule the rows keep their order
    by_mol = lambda lines: {q: [l for l in lines if l.split("\t")[0] == q] for q in {l.split("\t")[0] for l in lines}}
    assert by_mol(got[2:]) == by_mol(want[2:])
    # one rank: the same text without a process group
    assert distributed.gather_report(whole) == buf.getvalue()


def _pass_table_worker(rank, world, port, out_path):
    """One gather per pass (distributed.DeviceHitTable): each rank appends the rows of TWO queries of its shard to its
    table (kept in host memory here; in HBM on a GPU), then the table goes to rank 0 in one exchange."""
    import torch.distributed as dist
    from enzymm_amd import pyjess, synthetic
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mols = [pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb"),
                pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb")] + synthetic.structures(2)
        lo, hi = distributed.shard_range(len(mols), rank, world)
        table = distributed.DeviceHitTable(64, device="cpu")
        for i in range(lo, hi):  # one "query" per molecule: several appends per pass
            table.append_rows(_oracle_table(mols[i:i + 1], i), i)
        merged = table.gather()
        if rank == 0:
            np.save(out_path, merged)
        else:
            assert merged is None
        table.reset()
        assert int(table.count.item()) == 0
    finally:
        dist.destroy_process_group()


def test_one_gather_per_pass_world_size_2(tmp_path):
    import torch.multiprocessing as mp
    from enzymm_amd import pyjess, synthetic
    out = tmp_path / "pass_table.npy"
    mp.spawn(_pass_table_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    got = np.load(out)
    mols = [pyjess.Molecule.load(helpers.GOLDEN / "1AMY.pdb"),
            pyjess.Molecule.load(helpers.GOLDEN / "AF-P0DUB6-F1-model_v4.pdb")] + synthetic.structures(2)
    want = _oracle_table(mols, 0)
    assert len(got) == len(want) > 0
    assert got.tobytes() == want.tobytes()
    assert distributed.table_digest(got) == distributed.table_digest(want[::-1])  # the digest ignores the order of rows


def test_pass_table_without_process_group_and_overflow():
    rows = np.zeros(5, dtype=_native.HIT_DTYPE)
    rows["structure"] = [3, 1, 1, 0, 2]
    rows["template_sel"] = [0, 7, 2, 5, 1]
    t = distributed.DeviceHitTable(8, device="cpu")
    t.append_rows(rows[:2], 10)
    t.append_rows(rows[2:], 10)
    out = t.gather()
    assert out["structure"].tolist() == [10, 11, 11, 12, 13] and out["template_sel"].tolist() == [5, 2, 7, 1, 0]
    small = distributed.DeviceHitTable(4, device="cpu")
    small.append_rows(rows)
    with pytest.raises(_native.NativeError):
        small.gather()


def _overflow_worker(rank, world, port, out_path):
    """Rank 1's table overflows, rank 0's does not: BOTH must raise TM_E_OVERFLOW from gather() -- rank 1 takes part in
    the exchange of the counts before it raises, so rank 0 is not left waiting in a collective."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world, timeout=__import__("datetime").timedelta(seconds=60))
    try:
        rows = np.zeros(6, dtype=_native.HIT_DTYPE)
        rows["structure"] = np.arange(6)
        table = distributed.DeviceHitTable(4, device="cpu")
        table.append_rows(rows[:2] if rank == 0 else rows, 100 * rank)  # 2 rows fit, 6 do not
        try:
            table.gather()
            outcome = "no error"
        except _native.NativeError as exc:
            outcome = f"{exc.code}: {exc}"
        with open(f"{out_path}.{rank}", "w") as handle:
            handle.write(outcome)
    finally:
        dist.destroy_process_group()


def test_overflowed_rank_does_not_hang_the_gather(tmp_path):
    import torch.multiprocessing as mp
    out = tmp_path / "outcome"
    mp.spawn(_overflow_worker, args=(2, _free_port(), str(out)), nprocs=2, join=True)
    for rank in (0, 1):
        text = (tmp_path / f"outcome.{rank}").read_text()
        assert text.startswith(f"{_native.TM_E_OVERFLOW}:") and "rank(s) 1 (6 rows, room for 4)" in text, text
