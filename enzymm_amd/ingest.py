"""Batch ingestion of PDB files on the GPU.

``load_molecules`` (/root/reference/enzymm/jess_run.py:523-556) parses one file at a time through
``pyjess.Molecule.load``; at tens of thousands of structures per second that parse, not the
search, is the wall clock.  ``PdbBatch`` uploads the raw file bytes once and lets
``tm_batch_create_from_pdb`` do the fixed-column parse, the ENDMDL cut, the temperature-factor
filter and the structure-of-arrays packing on the device.  The ``Molecule`` objects it hands out
are lazy: their atom table is parsed on the host only when something (a ``Hit``'s atoms, a TSV
row) actually needs it.
"""
from __future__ import annotations

import io
import threading
from pathlib import Path
from typing import List, Optional, Sequence

import numpy as np

from . import _native, pyjess
from ._vocab import VOCAB


class LazyMolecule(pyjess.Molecule):
    """A ``Molecule`` backed by PDB text; columns appear on first use."""

    _lock = threading.Lock()  # shared: the lazy ones are rarely flattened on the host (Molecule._encoded)

    # Defaults live on the class: a batch of a thousand files makes a thousand of these on an ingest thread with the
    # GIL held, while the main thread wants it to launch and collect searches.
    _loaded = None
    _atoms = None
    _enc = None
    #: filled by PdbBatch.annotate(): byte offset of every kept record in the text, and the number of
    #: distinct residue numbers among the counted residue names (what a TSV row needs) -- with those a
    #: match is reported without parsing the whole file on the host
    _line_off = None
    _residue_count = None
    _picked = None  # {kept-record index: Atom}, made on first use
    #: PdbBatch.annotate_hits: the records of the matched atoms, copied out of the batch's text buffer -- sorted
    #: kept-record indices and, row for row, 96 bytes from the start of each record's line
    _picked_idx = None
    _picked_lines = None
    _row = None

    def __init__(self, text: Optional[bytes], id: Optional[str], n_atoms: int, min_bfactor: float, ignore_endmdl: bool,
                 path: Optional[str] = None):
        #: the file's bytes; None for a molecule of a batch read natively (PdbBatch.from_paths) until something
        #: needs them -- PdbBatch.annotate hands them to molecules with matches, anything else re-reads ``path``
        self._text_bytes = text
        self._path = path
        self.id = id
        self._n = n_atoms
        self._min_bfactor = min_bfactor
        self._ignore_endmdl = ignore_endmdl

    @property
    def _text(self):
        if self._text_bytes is None:
            if self._path is None:
                raise RuntimeError(f"the text of molecule {self.id!r} is gone with its batch")
            with open(self._path, "rb") as handle:
                self._text_bytes = handle.read()
        return self._text_bytes

    @_text.setter
    def _text(self, value):
        self._text_bytes = value

    def _atoms_at(self, indices) -> List[pyjess.Atom]:
        """The atoms with these (kept-record) indices; only their lines are parsed."""
        if self._loaded is not None or (self._line_off is None and self._picked_idx is None):
            return [self._materialise()._atom(int(i)) for i in indices]
        if self._picked is None:
            self._picked = {}
        missing = [int(i) for i in indices if int(i) not in self._picked]
        if missing:
            lines = []
            for i in missing:
                if self._picked_idx is not None:
                    row = int(np.searchsorted(self._picked_idx, i))
                    if row >= len(self._picked_idx) or int(self._picked_idx[row]) != i:  # not a matched atom
                        return [self._materialise()._atom(int(j)) for j in indices]
                    piece = self._picked_lines[row].tobytes()
                else:
                    lo = int(self._line_off[i])
                    piece = bytes(self._text[lo: lo + 96])  # a record is at most 80 columns + its line end
                hi = piece.find(b"\n")
                lines.append(piece if hi < 0 else piece[:hi])
            cols, xyz, _ = pyjess._parse_pdb_bytes(b"\n".join(lines) + b"\n", True)
            part = pyjess.Molecule.__new__(pyjess.Molecule)
            part._init_columns(cols, xyz, self.id)
            for j, i in enumerate(missing):
                self._picked[i] = part._atom(j)
        return [self._picked[int(i)] for i in indices]

    def _materialise(self) -> pyjess.Molecule:
        if self._loaded is None:
            mol = pyjess.Molecule.load(io.StringIO(bytes(self._text).decode("ascii", "replace")), id=self.id,
                                       ignore_endmdl=self._ignore_endmdl)
            if self._min_bfactor > 0:
                mol = mol.conserved(self._min_bfactor)
            if len(mol) != self._n:
                raise RuntimeError(f"host and device parsers disagree on {self.id!r}: {len(mol)} vs {self._n} atoms")
            self._loaded = mol
        return self._loaded

    _cols = property(lambda self: self._materialise()._cols)
    _xyz = property(lambda self: self._materialise()._xyz)

    def __len__(self):
        return self._n

    def __bool__(self):
        return self._n > 0


class PdbBatch:
    """PDB files resident in HBM, ready for ``Jess.query_batch(..., batch=pdb_batch)``."""

    def __init__(self, texts: Optional[Sequence[bytes]], ids: Sequence[Optional[str]], device: Optional[int] = None,
                 conservation_cutoff: float = 0.0, ignore_endmdl: bool = False, paths: Optional[Sequence[str]] = None):
        self.device = pyjess.default_device() if device is None else int(device)
        tables = VOCAB.native_tables()
        self.vocab_version = tables[-1]
        from .jess_run import _COUNTED_RESIDUES
        if texts is None:  # read natively: one page-locked buffer, one upload; the text stays on the native side
            self.native = _native.Batch.from_files(self.device, paths, tables, min_bfactor=conservation_cutoff,
                                                   ignore_endmdl=ignore_endmdl, counted_residues=sorted(_COUNTED_RESIDUES))
            texts = [None] * len(paths)
        else:
            if not isinstance(texts, _native.TextBlock):
                texts = list(texts)
            self.native = _native.Batch.from_pdb(self.device, texts, tables, min_bfactor=conservation_cutoff,
                                                 ignore_endmdl=ignore_endmdl, counted_residues=sorted(_COUNTED_RESIDUES))
        self._from_files = paths is not None and texts[:1] == [None]
        self.native.vocab_version = self.vocab_version
        counts = np.diff(self.native.atom_offsets).tolist()
        paths = list(paths) if paths is not None else [None] * len(counts)
        # files beyond the engine's atom limit: empty structures in this batch (no hits), reported, the rest searched
        self.refused = self.native.refused()
        pyjess.warn_refused([(paths[k] if paths[k] is not None else ids[k], n) for k, n in self.refused])
        self.molecules: List[LazyMolecule] = [
            LazyMolecule(texts[k], ids[k], counts[k], conservation_cutoff, ignore_endmdl, paths[k])
            for k in range(len(counts))]
        self._annotated = False

    def annotate(self, indices) -> None:
        """Give the molecules at ``indices`` (the ones with matches) what a report row needs from the device --
        record offsets for parsing only the matched atoms, and the counted-residue total -- while the batch is
        still alive.  One device-to-host copy per batch."""
        indices = list(indices)
        if not indices:
            return
        if not self._annotated:
            self._line_off = self.native.line_offsets()
            self._res_counts = self.native.residue_counts() if self.native.has_residue_counts else None
            self._annotated = True
        off = self.native.atom_offsets
        for i in indices:
            mol = self.molecules[i]
            # a molecule with matches outlives the batch (Matcher.run_files keeps it as a dict key): give it its own
            # copy of its file's text -- a view would pin the whole batch buffer (about 1000 files) per survivor
            if mol._text_bytes is None and self._from_files:
                mol._text_bytes = self.native.file_text(i)
            elif isinstance(mol._text_bytes, memoryview):
                mol._text_bytes = bytes(mol._text_bytes)
            mol._line_off = self._line_off[int(off[i]): int(off[i + 1])].copy()
            if self._res_counts is not None:
                mol._residue_count = int(self._res_counts[i])

    def annotate_hits(self, table: np.ndarray) -> None:
        """The bulk form of ``annotate`` for batches read natively: from the raw hit table, copy the records of every
        matched atom out of the batch's text buffer (one gather for the whole batch) and hand each molecule with
        hits its rows and its counted-residue total.  A matched molecule then never needs its file again unless
        somebody asks for atoms that were not matched (it is re-read from its path in that case)."""
        if len(table) == 0:
            return
        if not self._from_files:
            return self.annotate(np.unique(table["structure"]).tolist())
        if getattr(self, "_res_counts_read", None) is None:
            self._res_counts = self.native.residue_counts() if self.native.has_residue_counts else None
            self._res_counts_read = True
        # per structure the distinct matched atoms, ascending, with 96 bytes from the start of each one's record: found
        # and copied by the library on host threads (text and record offsets are on the host with the batch)
        keys, lines = self.native.pick_records(table, 96)
        structure = (keys >> np.uint64(32)).astype(np.int64)
        atom = (keys & np.uint64(0xFFFFFFFF)).astype(np.int64)
        matched, lo = np.unique(structure, return_index=True)
        hi = np.append(lo[1:], len(structure)).tolist()
        counts = self._res_counts.tolist() if self._res_counts is not None else None
        molecules = self.molecules
        for mi, a, b in zip(matched.tolist(), lo.tolist(), hi):
            mol = molecules[mi]
            mol._picked_idx = atom[a:b]
            mol._picked_lines = lines[a:b]
            if counts is not None:
                mol._residue_count = counts[mi]

    @classmethod
    def from_paths(cls, paths: Sequence, **kw) -> "PdbBatch":
        """Same id rule as load_molecules: the file stem, with _2, _3 ... for repeated stems."""
        import os

        paths = [p if type(p) is str else os.fspath(p) for p in paths]
        seen: dict = {}
        ids = []
        for p in paths:
            name = p[p.rfind(os.sep) + 1:]
            dot = name.rfind(".")
            stem = name[:dot] if 0 < dot < len(name) - 1 else name  # pathlib's PurePath.stem
            count = seen[stem] = seen.get(stem, 0) + 1
            ids.append(stem if count == 1 else f"{stem}_{count}")
        return cls(None, ids, paths=paths, **kw)  # read, uploaded and parsed without passing through Python

    def __len__(self):
        return len(self.molecules)

    def close(self):
        self.native.close()


class MoleculeBatch:
    """Host-parsed molecules flattened into one device-resident batch (``pyjess.make_batch``): what
    ``Matcher._run_pipelined`` needs of a batch -- ``native``, ``molecules``, ``close()``."""

    def __init__(self, molecules: Sequence[pyjess.Molecule], device: Optional[int] = None):
        self.molecules = list(molecules)
        self.native = pyjess.make_batch(self.molecules, pyjess.default_device() if device is None else int(device))

    def __len__(self):
        return len(self.molecules)

    def close(self):
        self.native.close()


def _prefetched(items, build, prefetch: int, workers: int):
    """``build(item)`` for every item, in order, on ``workers`` threads up to ``prefetch`` results ahead of the
    consumer.  A build error is raised in the consumer; results it never takes are closed."""
    from collections import deque
    from concurrent.futures import ThreadPoolExecutor

    items = iter(items)
    prefetch = max(1, int(prefetch))
    pool = ThreadPoolExecutor(max_workers=max(1, min(int(workers), prefetch)), thread_name_prefix="enzymm_amd-ingest")
    pending: "deque" = deque()

    def launch():
        item = next(items, None)
        if item is not None:
            pending.append(pool.submit(build, item))

    try:
        for _ in range(prefetch):
            launch()
        while pending:
            batch = pending.popleft().result()  # an ingestion error is raised here, in the consumer
            launch()
            yield batch
    finally:
        # the consumer stopped early or an error came through: close what it will not take, leave no thread behind
        for fut in pending:
            if not fut.cancel():
                try:
                    fut.result().close()
                except BaseException:
                    pass
        pool.shutdown(wait=True)


def iter_molecule_batches(molecules: Sequence[pyjess.Molecule], batch_size: int = 2000, max_atoms: int = 8_000_000,
                          prefetch: int = 3, workers: int = 2, device: Optional[int] = None):
    """Yield ``MoleculeBatch`` objects over consecutive slices of ``molecules`` (at most ``batch_size`` molecules
    or ``max_atoms`` atoms each, at least one molecule), flattened and uploaded on a worker thread ahead of the
    consumer (the native call releases the GIL).  The consumer closes each batch."""
    chunks, chunk, atoms = [], [], 0
    for m in molecules:
        if chunk and (len(chunk) >= batch_size or atoms + len(m) > max_atoms):
            chunks.append(chunk)
            chunk, atoms = [], 0
        chunk.append(m)
        atoms += len(m)
    if chunk:
        chunks.append(chunk)
    return _prefetched(chunks, lambda c: MoleculeBatch(c, device), prefetch, workers)


def iter_batches(paths: Sequence, batch_size: int = 1000, prefetch: int = 3, workers: int = 3, **kw):
    """Yield ``PdbBatch`` objects over ``paths`` in slices of ``batch_size`` files, in order, built on ``workers``
    ingest threads up to ``prefetch`` batches ahead: reading, uploading and parsing batch i+1 (and i+2) overlaps
    the search of batch i, and with two workers the file reads of one batch overlap the upload and the device parse
    of the other (every host thread ingests on a HIP stream of its own, and the native calls release the GIL).
    The consumer closes each batch when it is done with it."""
    paths = list(paths)
    slices = [paths[i:i + batch_size] for i in range(0, len(paths), batch_size)]
    return _prefetched(slices, lambda chunk: PdbBatch.from_paths(chunk, **kw), prefetch, workers)
