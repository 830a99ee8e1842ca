"""Sharding of the query-structure batch over the GPUs of one node and the final hit-table gather.

The (structure x template) grid is embarrassingly parallel: nothing in
``Matcher.run`` couples molecules (/root/reference/enzymm/jess_run.py:896-988; the reference's
own analogue is Nextflow's file sharding, nextflow/template_matcher.nf:43-46).  So each rank owns
a contiguous range of structures, holds a full copy of the template library, and the only
exchange is one variable-length gather of the per-rank hit tables to rank 0 -- RCCL when the
tables live in HBM (backend "nccl"), gloo in the CPU tests.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np

from . import _native


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) range of rank ``rank``; sizes differ by at most one."""
    if world < 1 or not 0 <= rank < world:
        raise ValueError(f"bad rank {rank} of {world}")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard_by_atoms(atom_counts, world: int):
    """Contiguous ranges balanced by atom count (large complexes, BASELINE.json configs[4])."""
    counts = np.asarray(atom_counts, dtype=np.float64)
    total = counts.sum()
    bounds = [0]
    csum = np.cumsum(counts)
    for r in range(1, world):
        bounds.append(int(np.searchsorted(csum, total * r / world, side="left")))
    bounds.append(len(counts))
    bounds = np.maximum.accumulate(bounds)
    return [(int(bounds[r]), int(bounds[r + 1])) for r in range(world)]


def gather_tables(local: np.ndarray, structure_offset: int, group=None, device=None) -> Optional[np.ndarray]:
    """Gather structured hit tables (``_native.HIT_DTYPE``) to rank 0.

    ``structure`` indices are made global by adding ``structure_offset``.  Returns the
    concatenated table (sorted by structure, template) on rank 0 and ``None`` elsewhere.  Works
    on any torch.distributed backend: tensors are staged on ``device`` ("cuda" for RCCL).
    """
    import torch
    import torch.distributed as dist

    local = np.ascontiguousarray(local, dtype=_native.HIT_DTYPE).copy()
    local["structure"] += np.uint32(structure_offset)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device(device) if device is not None else torch.device("cpu")
    count = torch.tensor([len(local)], dtype=torch.int64, device=dev)
    counts = [torch.zeros_like(count) for _ in range(world)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    width = _native.HIT_DTYPE.itemsize
    pad = max(counts + [1])
    buf = torch.zeros(pad * width, dtype=torch.uint8, device=dev)
    if len(local):
        buf[: len(local) * width] = torch.from_numpy(local.view(np.uint8).reshape(-1)).to(dev)
    out = [torch.zeros_like(buf) for _ in range(world)] if rank == 0 else None
    dist.gather(buf, out, dst=0, group=group)
    if rank != 0:
        return None
    return merge_tables([out[r][: counts[r] * width].cpu().numpy().view(_native.HIT_DTYPE) for r in range(world)])


def merge_tables(parts) -> np.ndarray:
    """Per-rank hit tables (``structure`` already global) -> one table in (structure, template) order: what rank 0
    holds after the gather.  Several hits of one pair (best_match off) keep their per-rank order."""
    parts = [np.ascontiguousarray(p, dtype=_native.HIT_DTYPE) for p in parts]
    table = np.concatenate(parts) if parts else np.zeros(0, dtype=_native.HIT_DTYPE)
    order = np.lexsort((table["template_sel"], table["structure"]))  # stable
    return table[order]


class _DeviceBytes:
    """Device memory owned by someone else, presented to torch through ``__cuda_array_interface__``."""

    def __init__(self, pointer: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (pointer, False), "version": 2}


def gather_result(result: "_native.Result", n_hits: int, structure_offset: int, group=None,
                  device=None, force_collective: bool = False) -> Optional[np.ndarray]:
    """Gather a query's hit table to rank 0 straight from HBM: the rows leave the device table the search wrote
    (``tm_result_device_hits``) over RCCL, with no device-to-host-to-device round trip on the sending ranks.  The
    device table is unsorted and carries batch-local structure indices; rank 0 adds every rank's
    ``structure_offset`` and sorts.  Returns the merged table on rank 0, None elsewhere.  Needs a backend that moves
    device tensors ("nccl" = RCCL); with one rank it is ``result.hits()`` plus the offset (``force_collective``:
    go through the collectives all the same -- how a one-GPU box tests the RCCL leg)."""
    import torch
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force_collective):
        local = result.hits()
        local["structure"] += np.uint32(structure_offset)
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    width = _native.HIT_DTYPE.itemsize
    meta = torch.tensor([n_hits, structure_offset], dtype=torch.int64, device=dev)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    counts = [int(m[0].item()) for m in metas]
    offsets = [int(m[1].item()) for m in metas]
    pad = max(counts + [1]) * width
    send = torch.zeros(pad, dtype=torch.uint8, device=dev)
    if n_hits:
        rows = torch.as_tensor(_DeviceBytes(result.device_hits_ptr(), n_hits * width), device=dev)
        send[: n_hits * width] = rows  # device-to-device
    out = [torch.zeros_like(send) for _ in range(world)] if rank == 0 else None
    dist.gather(send, out, dst=0, group=group)
    if rank != 0:
        return None
    parts = []
    for r in range(world):
        part = out[r][: counts[r] * width].cpu().numpy().view(_native.HIT_DTYPE).copy()
        part["structure"] += np.uint32(offsets[r])
        parts.append(part)
    table = np.concatenate(parts) if parts else np.zeros(0, dtype=_native.HIT_DTYPE)
    # (structure, template) order; several hits of one pair (best_match off) by ascending atom tuple, like
    # tm_result_copy_hits
    keys = [table["atoms"][:, c] for c in range(table["atoms"].shape[1] - 1, -1, -1)]
    return table[np.lexsort(keys + [table["template_sel"], table["structure"]])]


def run_files_sharded(matcher, paths, group=None, device_ingest: bool = True, batch_size: int = 1000, **ingest_options):
    """One process per GPU over a list of structure files -- the single-node counterpart of the reference's Nextflow
    sharding (/root/reference/nextflow/template_matcher.nf:43-62: ``splitText(by: batch_size)`` -> one CLI run per
    shard): rank r takes the contiguous range ``shard_range(len(paths), r, world)`` and runs the matcher on it.
    Returns ``(matches of this rank's shard, lo, hi)``; merge reports with ``gather_report``.  ``device_ingest``
    False loads the shard through ``load_molecules`` (host parser) instead of ``Matcher.run_files``."""
    import torch.distributed as dist

    paths = list(paths)
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
    rank = dist.get_rank(group) if world > 1 else 0
    lo, hi = shard_range(len(paths), rank, world)
    if device_ingest:
        local = matcher.run_files(paths[lo:hi], batch_size=batch_size, **ingest_options)
    else:
        from . import jess_run
        local = matcher.run(jess_run.load_molecules(paths[lo:hi]))
    return local, lo, hi


def gather_report(processed, group=None, predict_correctness: bool = True) -> Optional[str]:
    """The TSV of a sharded run on rank 0 (None elsewhere): every rank renders its rows exactly as ``report.write_tsv``
    would, rank 0 concatenates them in rank order under one header -- what the reference's ``collectFile`` merge does
    with its per-batch TSVs (nextflow/template_matcher.nf:65-78).  As there, the rows are those of a single run over
    all files; the ORDER of the molecules is per shard (inside a shard: first the molecules matched by the largest
    templates, jess_run.py:930), so it can differ from the single run's.  Python objects travel, so the group needs an
    object collective (gloo, or RCCL's via torch)."""
    import io

    import torch.distributed as dist

    from . import report

    buf = io.StringIO()
    report.write_tsv(processed, buf, predict_correctness=predict_correctness)
    lines = buf.getvalue().splitlines(keepends=True)
    version, header, rows = lines[0], (lines[1] if len(lines) > 1 else None), lines[2:]
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return buf.getvalue()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    gathered = [None] * world if rank == 0 else None
    dist.gather_object((header, rows), gathered, dst=0, group=group)
    if rank != 0:
        return None
    header = next((h for h, _ in gathered if h is not None), None)
    return version + (header or "") + "".join("".join(r) for _, r in gathered)


class DeviceHitTable:
    """A rank's hit table, collected in HBM over the queries of one pass and sent to rank 0 in ONE gather.

    ``append(result, structure_offset)`` enqueues a device-side copy of the query's rows behind its kernels
    (``tm_result_append_hits``: no host wait, no host copy); ``gather()`` is the pass's only exchange: the row counts
    in one ``all_gather``, the rows -- straight from this table's memory -- in one padded ``gather`` to rank 0, which
    sorts them into (structure, template) order.  This is north_star's "only a final RCCL gather of the hit table"
    and the single merge of the reference's sharded flow (/root/reference/nextflow/template_matcher.nf:65-78).
    With a backend that cannot move device tensors (gloo in the CPU/one-GPU tests) the rows are staged on the host
    for the collective only."""

    def __init__(self, capacity_rows: int, device=None):
        import torch

        self.capacity = int(capacity_rows)
        self.device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.width = _native.HIT_DTYPE.itemsize
        self.rows = torch.zeros(max(1, self.capacity) * self.width, dtype=torch.uint8, device=self.device)
        self.count = torch.zeros(1, dtype=torch.int64, device=self.device)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)  # the zero fills ran on torch's stream; appends come on the library's

    def reset(self):
        """Empty the table (enqueued on torch's current stream; call when no append is in flight)."""
        self.count.zero_()

    def append(self, result: "_native.Result", structure_offset: int = 0):
        result.append_hits(self.rows.data_ptr(), self.capacity, self.count.data_ptr(), structure_offset)

    def append_rows(self, table: np.ndarray, structure_offset: int = 0):
        """Host rows (``_native.HIT_DTYPE``) into the table -- for hit tables that already left the device, and for the
        CPU tests of the gather (a table on ``device="cpu"``)."""
        import torch

        table = np.ascontiguousarray(table, dtype=_native.HIT_DTYPE).copy()
        table["structure"] += np.uint32(structure_offset)
        n0, n = int(self.count.item()), len(table)
        if n0 + n <= self.capacity and n:
            raw = torch.from_numpy(table.view(np.uint8).reshape(-1))
            self.rows[n0 * self.width: (n0 + n) * self.width] = raw.to(self.device)
        self.count += n

    def gather(self, group=None, force_collective: bool = False, sort: bool = True):
        """Every append must have completed (wait for the results first).  Returns rank 0's merged table (None on the
        other ranks); ``sort=False`` leaves the rows in rank order, unsorted (rank 0 can sort later, off the
        critical path, with ``sort_rows``)."""
        import torch
        import torch.distributed as dist

        n_local = int(self.count.item())  # the pass's one read-back of a count
        overflow = _native.NativeError(_native.TM_E_OVERFLOW,
                                       f"device hit table holds {self.capacity} rows, the pass produced {n_local}")
        active = dist.is_available() and dist.is_initialized()
        if not active or (dist.get_world_size(group) == 1 and not force_collective):
            if n_local > self.capacity:
                raise overflow
            table = self.rows[: n_local * self.width].cpu().numpy().view(_native.HIT_DTYPE).copy()
            return sort_rows(table) if sort else table
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        on_device = dist.get_backend(group) != "gloo"
        comm = self.device if on_device else torch.device("cpu")
        # every rank takes part in the exchange of the counts -- an overflowed rank included (it sends its count and its
        # capacity) -- and all of them raise together: a rank that raised BEFORE the collective would leave the others
        # waiting in it until the watchdog fires
        mine = torch.tensor([n_local, self.capacity], dtype=torch.int64, device=comm)
        counts = torch.zeros(2 * world, dtype=torch.int64, device=comm)
        dist.all_gather_into_tensor(counts, mine, group=group)
        counts = counts.cpu().tolist()
        counts, capacities = counts[0::2], counts[1::2]
        full = [r for r in range(world) if counts[r] > capacities[r]]
        if full:
            raise _native.NativeError(_native.TM_E_OVERFLOW, "device hit table too small on rank(s) " +
                                      ", ".join(f"{r} ({counts[r]} rows, room for {capacities[r]})" for r in full))
        pad = max(counts + [1]) * self.width
        if pad > self.rows.numel():  # (equal capacities on every rank rule this out; unequal ones: every rank sees the same counts)
            raise _native.NativeError(_native.TM_E_OVERFLOW, "a rank's hit table is larger than this rank's capacity")
        send = self.rows[:pad] if on_device else self.rows[:pad].cpu()  # a view of the table itself: no staging copy
        out = [torch.empty_like(send) for _ in range(world)] if rank == 0 else None
        dist.gather(send, out, dst=0, group=group)
        if rank != 0:
            return None
        parts = [out[r][: counts[r] * self.width].cpu().numpy().view(_native.HIT_DTYPE) for r in range(world)]
        table = np.concatenate(parts) if parts else np.zeros(0, dtype=_native.HIT_DTYPE)
        return sort_rows(table) if sort else table


def sort_rows(table: np.ndarray) -> np.ndarray:
    """Hit rows in (structure, template) order; several hits of one pair (best_match off) by ascending atom tuple,
    like ``tm_result_copy_hits``."""
    if len(table) < 2:
        return table
    keys = [table["atoms"][:, c] for c in range(table["atoms"].shape[1] - 1, -1, -1)]
    return table[np.lexsort(keys + [table["template_sel"], table["structure"]])]


def table_digest(table: np.ndarray) -> str:
    """Order-independent fingerprint of a hit table's identity columns (structure, template, matched atoms): what a
    sharded run and the unsharded run of the same structures must agree on."""
    import hashlib

    t = sort_rows(np.ascontiguousarray(table, dtype=_native.HIT_DTYPE))
    h = hashlib.sha256()
    for col in ("structure", "template_sel", "n_atoms", "atoms"):
        h.update(np.ascontiguousarray(t[col]).tobytes())
    return h.hexdigest()[:16]
