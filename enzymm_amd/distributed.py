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


def gather_hit_table(result: "_native.Result", n_hits: int, structure_offset: int, world: int, rank: int,
                     device: str):
    """bench.py's per-step exchange: this rank's (sorted) hit table, gathered to rank 0 when world > 1
    (``device`` "cuda:N" = staged in HBM and moved by RCCL)."""
    table = result.hits()
    if world == 1:
        return table
    return gather_tables(table, structure_offset, device=device)
