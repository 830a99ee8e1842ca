"""ctypes binding of ``libenzymm_amd.so`` (C ABI: ``include/enzymm_amd.h``).

There is deliberately no CPU fallback: if the shared library is missing or no
MI355X is visible, every entry point raises.
"""
from __future__ import annotations

import ctypes as C
import sys as _sys
import threading
from pathlib import Path

import numpy as np

TM_MAX_TEMPLATE_ATOMS = 24
TM_MAX_TEMPLATE_GROUPS = 8
TM_MAX_STRUCTURE_ATOMS = 16777215   # beyond it a structure is refused (kept in its batch as an empty one)
TM_MAX_STRUCTURE_RESIDUES = 65533   # residues of one structure (the kernels number them with 16 bits); beyond: refused like the above
TM_NARROW_STRUCTURE_ATOMS = 65535  # up to here the 16-bit atom indices of the tuned kernels; larger: the template-at-a-time launch, 24-bit indices
TM_RES_OTHER = 31
TM_NAME_OTHER = 63

TM_OK, TM_E_INVALID, TM_E_UNSUPPORTED, TM_E_DEVICE, TM_E_NOMEM, TM_E_OVERFLOW = 0, -1, -2, -3, -4, -5
TM_FLAG_IGNORE_CHAIN, TM_FLAG_BEST_MATCH, TM_FLAG_COUNT_CANDIDATES, TM_FLAG_REBUILD_INDEX = 1, 2, 4, 8
TM_HIT_TRUNCATED, TM_HIT_PREDICTED, TM_HIT_CORRECT = 1, 2, 4

import os

# $ENZYMM_AMD_LIB selects another build of the same ABI (the diagnostic `make stats` library)
LIB_PATH = Path(os.environ.get("ENZYMM_AMD_LIB") or Path(__file__).resolve().parent / "libenzymm_amd.so")

#: numpy view of ``struct tm_hit`` (232 bytes)
HIT_DTYPE = np.dtype(
    [
        ("structure", "<u4"), ("template_sel", "<u4"), ("n_candidates", "<u4"), ("flags", "<u4"),
        ("rmsd", "<f8"), ("rot", "<f8", (9,)), ("trans", "<f8", (3,)),
        ("n_atoms", "<u4"), ("atoms", "<u4", (TM_MAX_TEMPLATE_ATOMS,)), ("reserved", "<u4"),
        ("orientation", "<f8"),
    ],
    align=False,
)
assert HIT_DTYPE.itemsize == 232


class _LibraryDesc(C.Structure):
    _fields_ = [
        ("n_templates", C.c_uint32), ("atom_offsets", C.c_void_p), ("xyz", C.c_void_p), ("dist_weight", C.c_void_p),
        ("res_mask", C.c_void_p), ("name_mask", C.c_void_p), ("any_residue", C.c_void_p), ("group", C.c_void_p),
        ("chain", C.c_void_p), ("residue_offsets", C.c_void_p), ("orient_first", C.c_void_p),
        ("orient_second", C.c_void_p), ("orient_vec", C.c_void_p),
    ]


class _BatchDesc(C.Structure):
    _fields_ = [
        ("n_structures", C.c_uint32), ("atom_offsets", C.c_void_p), ("xyz", C.c_void_p), ("res_type", C.c_void_p),
        ("name_id", C.c_void_p), ("residue", C.c_void_p), ("chain", C.c_void_p),
    ]


class _ColumnsDesc(C.Structure):
    _fields_ = [
        ("n_structures", C.c_uint32), ("atom_offsets", C.c_void_p), ("xyz", C.c_void_p), ("name", C.c_void_p),
        ("residue_name", C.c_void_p), ("chain_id", C.c_void_p), ("residue_number", C.c_void_p),
        ("insertion_code", C.c_void_p),
    ]


class _Vocab(C.Structure):
    _fields_ = [
        ("n_atom_names", C.c_uint32), ("atom_names", C.c_void_p), ("atom_name_ids", C.c_void_p),
        ("n_elements", C.c_uint32), ("elements", C.c_void_p), ("element_ids", C.c_void_p),
        ("n_residues", C.c_uint32), ("residue_names", C.c_void_p), ("residue_ids", C.c_void_p),
        ("n_counted_residues", C.c_uint32), ("counted_residue_names", C.c_void_p),
    ]


TM_PDB_IGNORE_ENDMDL = 1


class _QueryParams(C.Structure):
    _fields_ = [
        ("n_selected", C.c_uint32), ("template_index", C.c_void_p), ("rmsd_threshold", C.c_void_p),
        ("distance_cutoff", C.c_void_p), ("max_dynamic_distance", C.c_void_p), ("max_candidates", C.c_uint32),
        ("flags", C.c_uint32), ("hit_capacity", C.c_uint64), ("stream", C.c_void_p),
        ("candidate_capacity", C.c_uint64), ("n_model_sets", C.c_uint32), ("model_set", C.c_void_p),
        ("model_offsets", C.c_void_p), ("model_votes", C.c_void_p), ("models", C.c_void_p),
    ]


EXPORTS = (
    "tm_abi_version", "tm_last_error", "tm_device_count", "tm_library_create", "tm_library_destroy",
    "tm_batch_create", "tm_batch_create_from_columns", "tm_batch_create_from_structures", "tm_batch_create_from_pdb", "tm_batch_atom_counts", "tm_batch_line_offsets",
    "tm_batch_residue_counts",
    "tm_batch_destroy", "tm_query", "tm_result_wait", "tm_result_needed", "tm_result_unsearched", "tm_result_hit_count",
    "tm_batch_refused", "tm_result_copy_hits", "tm_result_copy_candidates", "tm_result_timing", "tm_result_search_split", "tm_result_page_census", "tm_result_wait_appended", "tm_result_device_hits",
    "tm_result_append_hits", "tm_result_destroy", "tm_pdb_count_host", "tm_pdb_parse_host", "tm_pdb_count_host_many",
    "tm_pdb_parse_host_many", "tm_files_sizes", "tm_files_read", "tm_batch_create_from_files", "tm_batch_file_text",
    "tm_stream_create", "tm_stream_destroy", "tm_batch_text_layout",
    "tm_batch_wait", "tm_batch_upload_bytes", "tm_batch_copy_layout", "tm_batch_line_offsets_at",
    "tm_batch_pick_records", "tm_host_threads",
)

_lock = threading.Lock()
_lib = None


class NativeError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libenzymm_amd: {msg} (code {code})")
        self.code = code


def load() -> C.CDLL:
    """Load the shared library (no GPU needed for this step)."""
    global _lib
    with _lock:
        if _lib is None:
            if not LIB_PATH.exists():
                raise ImportError(
                    f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C enzymm_amd/csrc`. There is no CPU fallback."
                )
            # PyTorch bundles its own HIP/HSA runtime; a process can only hold one.  Loading torch first makes this
            # library bind to the runtime torch uses (same sonames) -- the other order leaves torch.distributed / RCCL
            # with a second runtime that finds no device ("no ROCm-capable device is detected").
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
            lib = C.CDLL(str(LIB_PATH))
            lib.tm_last_error.restype = C.c_char_p
            lib.tm_result_device_hits.restype = C.c_void_p
            for name in EXPORTS:
                getattr(lib, name)  # AttributeError if the ABI is incomplete
            if lib.tm_abi_version() != 1:
                raise ImportError("libenzymm_amd.so ABI version mismatch")
            _lib = lib
    return _lib


def _finalizing() -> bool:
    """True while the interpreter shuts down: destructors then leave device objects to the process exit (the HIP runtime
    may already be gone, and calling into it crashes)."""
    try:
        return _sys.is_finalizing()
    except Exception:  # module globals already torn down
        return True


def _check(rc: int):
    if rc != TM_OK:
        raise NativeError(rc, (load().tm_last_error() or b"").decode(errors="replace"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


class _PdbColumns(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in (
        "serial", "name", "altloc", "residue_name", "chain_id", "residue_number", "insertion_code", "xyz",
        "occupancy", "temperature_factor", "segment", "element", "charge")]


_PDB_COLUMN_DTYPES = (
    ("serial", "<i4"), ("name", "S4"), ("altloc", "S1"), ("residue_name", "S3"), ("chain_id", "S2"),
    ("residue_number", "<i4"), ("insertion_code", "S1"), ("occupancy", "<f8"), ("temperature_factor", "<f8"),
    ("segment", "S4"), ("element", "S2"), ("charge", "i1"),
)


def parse_pdb_host(data: bytes, ignore_endmdl: bool = False):
    """One PDB file -> (columns, xyz, HEADER idCode) with the native host reader (``tm_pdb_parse_host``).
    Raises ValueError quoting the offending record, like a line-by-line parse would."""
    lib = load()
    flags = C.c_uint32(TM_PDB_IGNORE_ENDMDL if ignore_endmdl else 0)
    n = C.c_uint64(0)
    head = C.create_string_buffer(4)
    has_head = C.c_int(0)
    _check(lib.tm_pdb_count_host(data, C.c_uint64(len(data)), flags, C.byref(n), head, C.byref(has_head)))
    n = n.value
    cols = {k: np.zeros(n, dtype=dt) for k, dt in _PDB_COLUMN_DTYPES}
    xyz = np.zeros((n, 3), dtype=np.float64)
    header_id = None
    if has_head.value:
        header_id = head.raw.decode("ascii", "replace").strip() or None
    if n:
        out = _PdbColumns(**{k: _ptr(cols[k]) for k, _ in _PDB_COLUMN_DTYPES}, xyz=_ptr(xyz))
        bad_begin, bad_len = C.c_uint64(0), C.c_uint64(0)
        rc = lib.tm_pdb_parse_host(data, C.c_uint64(len(data)), flags, C.c_uint64(n), C.byref(out),
                                   C.byref(bad_begin), C.byref(bad_len))
        if rc == TM_E_INVALID and (lib.tm_last_error() or b"") == b"malformed PDB record":
            line = data[bad_begin.value: bad_begin.value + min(bad_len.value, 80)]
            raise ValueError(f"malformed PDB record: {line.decode('ascii', 'replace').rstrip()!r}")
        _check(rc)
    return cols, xyz, header_id


class TextBlock:
    """Many files' bytes in one buffer (``tm_files_read``): ``buffer`` uint8, file f at offsets[f]:offsets[f+1]."""

    def __init__(self, buffer: np.ndarray, offsets: np.ndarray):
        self.buffer, self.offsets = buffer, offsets

    def __len__(self):
        return len(self.offsets) - 1

    def __getitem__(self, i) -> memoryview:
        return memoryview(self.buffer)[int(self.offsets[i]): int(self.offsets[i + 1])]

    @property
    def pointers(self) -> np.ndarray:
        return (np.uint64(self.buffer.ctypes.data) + self.offsets[:-1]).astype(np.uint64)

    @property
    def lengths(self) -> np.ndarray:
        return np.diff(self.offsets).astype(np.uint64)

    @classmethod
    def from_bytes(cls, datas) -> "TextBlock":
        offsets = np.zeros(len(datas) + 1, dtype=np.uint64)
        np.cumsum([len(d) for d in datas], out=offsets[1:])
        return cls(np.frombuffer(b"".join(datas), dtype=np.uint8), offsets)


def read_files(paths) -> TextBlock:
    """Read many files on a team of native host threads (``tm_files_sizes`` + ``tm_files_read``)."""
    lib = load()
    raw = [os.fsencode(p) for p in paths]
    nf = len(raw)
    offsets = np.zeros(nf + 1, dtype=np.uint64)
    if nf == 0:
        return TextBlock(np.zeros(0, np.uint8), offsets)
    cpaths = (C.c_char_p * nf)(*raw)
    sizes = np.zeros(nf, dtype=np.uint64)
    bad = C.c_int64(-1)
    rc = lib.tm_files_sizes(C.c_uint32(nf), cpaths, _ptr(sizes), C.byref(bad))
    if rc == TM_OK:
        np.cumsum(sizes, out=offsets[1:])
        buffer = np.empty(int(offsets[-1]), dtype=np.uint8)
        rc = lib.tm_files_read(C.c_uint32(nf), cpaths, _ptr(offsets), _ptr(buffer) if len(buffer) else None, C.byref(bad))
        if rc == TM_OK:
            return TextBlock(buffer, offsets)
    if bad.value >= 0:
        open(paths[bad.value], "rb").close()  # raises the ordinary OSError (FileNotFoundError, IsADirectoryError ...)
    _check(rc)
    raise OSError(f"cannot read {paths[max(bad.value, 0)]}")


def parse_pdb_host_many(datas, ignore_endmdl: bool = False):
    """Many PDB files -> [(columns, xyz, HEADER idCode), ...] in two native calls that parse on a team of host
    threads (``tm_pdb_parse_host_many``); the per-file tables are views into shared arrays.  Raises the
    ValueError of the first malformed file, like the sequential loop would."""
    lib = load()
    nf = len(datas)
    if nf == 0:
        return []
    block = datas if isinstance(datas, TextBlock) else TextBlock.from_bytes(datas)
    flags = C.c_uint32(TM_PDB_IGNORE_ENDMDL if ignore_endmdl else 0)
    ptr_arr, len_arr = block.pointers, block.lengths
    ptrs, lens = _ptr(ptr_arr), _ptr(len_arr)
    counts = np.zeros(nf, dtype=np.uint64)
    heads = np.zeros(nf, dtype="S4")
    head_raw = C.create_string_buffer(4 * nf)
    has_head = np.zeros(nf, dtype=np.int32)
    _check(lib.tm_pdb_count_host_many(C.c_uint32(nf), ptrs, lens, flags, _ptr(counts), head_raw, _ptr(has_head)))
    offsets = np.zeros(nf + 1, dtype=np.uint64)
    np.cumsum(counts, out=offsets[1:])
    n = int(offsets[-1])
    # (uninitialised: the reader writes every column of every record it counted, or reports the file as malformed -- zeroing
    # 100 MB per thousand files on one thread was a fifth of load_molecules)
    cols = {k: np.empty(n, dtype=dt) for k, dt in _PDB_COLUMN_DTYPES}
    xyz = np.empty((n, 3), dtype=np.float64)
    out = _PdbColumns(**{k: _ptr(cols[k]) for k, _ in _PDB_COLUMN_DTYPES}, xyz=_ptr(xyz))
    bad_file, bad_begin, bad_len = C.c_int64(-1), C.c_uint64(0), C.c_uint64(0)
    rc = lib.tm_pdb_parse_host_many(C.c_uint32(nf), ptrs, lens, flags, _ptr(offsets), C.byref(out), C.byref(bad_file),
                                    C.byref(bad_begin), C.byref(bad_len))
    if rc == TM_E_INVALID and bad_file.value >= 0:
        data = bytes(block[bad_file.value])
        line = data[bad_begin.value: bad_begin.value + min(bad_len.value, 80)]
        raise ValueError(f"malformed PDB record: {line.decode('ascii', 'replace').rstrip()!r}")
    _check(rc)
    raw = head_raw.raw
    result = []
    for f in range(nf):
        lo, hi = int(offsets[f]), int(offsets[f + 1])
        header_id = (raw[4 * f: 4 * f + 4].decode("ascii", "replace").strip() or None) if has_head[f] else None
        result.append(({k: v[lo:hi] for k, v in cols.items()}, xyz[lo:hi], header_id))
    return result


class Stream:
    """A HIP stream owned by the library (``tm_stream_create``); ``handle`` goes into ``query(stream=...)``."""

    def __init__(self, device: int):
        self.device = device
        h = C.c_void_p()
        _check(load().tm_stream_create(C.c_int(device), C.byref(h)))
        self.handle = h.value

    def close(self):
        if getattr(self, "handle", None) and not _finalizing():
            load().tm_stream_destroy(C.c_int(self.device), C.c_void_p(self.handle))
            self.handle = None

    __del__ = close


def host_threads() -> int:
    """Host threads one native call of this process may use (``tm_host_threads``)."""
    return int(load().tm_host_threads())


def device_count() -> int:
    n = C.c_int(0)
    _check(load().tm_device_count(C.byref(n)))
    return n.value


class Library:
    """Device-resident template library (``tm_library``)."""

    def __init__(self, device, atom_offsets, xyz, dist_weight, res_mask, name_mask, any_residue, group, chain,
                 orientation=None):
        """``orientation`` = (residue_offsets, first, second, vectors) or None (see tm_library_desc)."""
        lib = load()
        self._h = C.c_void_p()
        self.n_templates = len(atom_offsets) - 1
        keep = [
            _arr(atom_offsets, np.uint32), _arr(xyz, np.float64), _arr(dist_weight, np.float64),
            _arr(res_mask, np.uint32), _arr(name_mask, np.uint64), _arr(any_residue, np.uint8),
            _arr(group, np.uint8), _arr(chain, np.uint8),
        ]
        if orientation is not None:
            keep += [_arr(orientation[0], np.uint32), _arr(orientation[1], np.uint8), _arr(orientation[2], np.uint8),
                     _arr(orientation[3], np.float64)]
        else:
            keep += [None, None, None, None]
        desc = _LibraryDesc(self.n_templates, *[_ptr(a) for a in keep])
        _check(lib.tm_library_create(C.c_int(device), C.byref(desc), C.byref(self._h)))
        self.device = device

    def close(self):
        if getattr(self, "_h", None) and not _finalizing():
            load().tm_library_destroy(self._h)
            self._h = None

    __del__ = close


class Batch:
    """Device-resident batch of target structures (``tm_batch``)."""

    def __init__(self, device, atom_offsets, xyz, res_type, name_id, residue, chain):
        lib = load()
        self._h = C.c_void_p()
        self.n_structures = len(atom_offsets) - 1
        self.atom_offsets = _arr(atom_offsets, np.uint32)
        keep = [
            self.atom_offsets, _arr(xyz, np.float64), _arr(res_type, np.uint8), _arr(name_id, np.uint8),
            _arr(residue, np.uint32), _arr(chain, np.uint8),
        ]
        desc = _BatchDesc(self.n_structures, *[_ptr(a) for a in keep])
        _check(lib.tm_batch_create(C.c_int(device), C.byref(desc), C.byref(self._h)))
        self.device = device

    @classmethod
    def from_columns(cls, device, atom_offsets, xyz, name, residue_name, chain_id, residue_number, insertion_code,
                     vocab_tables) -> "Batch":
        """Raw atom-table columns of many molecules (concatenated); vocabulary encoding, residue/chain ordinals and
        flattening happen in the library on host threads (``tm_batch_create_from_columns``)."""
        lib = load()
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.n_structures = len(atom_offsets) - 1
        names, name_ids, els, el_ids, res, res_ids, _ = vocab_tables
        vk = [_arr(names, "S4"), _arr(name_ids, np.uint8), _arr(els, "S2"), _arr(el_ids, np.uint8),
              _arr(res, "S3"), _arr(res_ids, np.uint8)]
        vocab = _Vocab(len(vk[0]), _ptr(vk[0]), _ptr(vk[1]), len(vk[2]), _ptr(vk[2]), _ptr(vk[3]),
                       len(vk[4]), _ptr(vk[4]), _ptr(vk[5]), 0, None)
        keep = [_arr(atom_offsets, np.uint32), _arr(xyz, np.float64), _arr(name, "S4"), _arr(residue_name, "S3"),
                _arr(chain_id, "S2"), _arr(residue_number, np.int32), _arr(insertion_code, "S1")]
        desc = _ColumnsDesc(self.n_structures, *[_ptr(a) for a in keep])
        _check(lib.tm_batch_create_from_columns(C.c_int(device), C.byref(desc), C.byref(vocab), C.byref(self._h)))
        self.device = device
        self.atom_offsets = keep[0]
        self.has_residue_counts = False
        return self

    @classmethod
    def from_structures(cls, device, rows: np.ndarray, vocab_tables) -> "Batch":
        """Raw atom-table columns of many molecules, each wherever it lies: ``rows`` is a ``[n, 7]`` uint64 table of
        ``tm_structure_columns`` records (atom count, then the addresses of xyz float64[n,3], name S4, residue_name S3,
        chain_id S2, residue_number int32, insertion_code S1 -- ``Molecule._pointer_row``).  The caller keeps the
        arrays alive for the duration of the call (``tm_batch_create_from_structures``)."""
        lib = load()
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        rows = _arr(rows, np.uint64).reshape(-1, 7)
        self.n_structures = len(rows)
        names, name_ids, els, el_ids, res, res_ids, _ = vocab_tables
        vk = [_arr(names, "S4"), _arr(name_ids, np.uint8), _arr(els, "S2"), _arr(el_ids, np.uint8),
              _arr(res, "S3"), _arr(res_ids, np.uint8)]
        vocab = _Vocab(len(vk[0]), _ptr(vk[0]), _ptr(vk[1]), len(vk[2]), _ptr(vk[2]), _ptr(vk[3]),
                       len(vk[4]), _ptr(vk[4]), _ptr(vk[5]), 0, None)
        _check(lib.tm_batch_create_from_structures(C.c_int(device), C.c_uint32(len(rows)), _ptr(rows), C.byref(vocab),
                                                   C.byref(self._h)))
        self.device = device
        self.atom_offsets = np.zeros(len(rows) + 1, dtype=np.uint32)
        np.cumsum(rows[:, 0], out=self.atom_offsets[1:], dtype=np.uint32)
        self.has_residue_counts = False
        return self

    @classmethod
    def from_pdb(cls, device, texts, vocab_tables, min_bfactor: float = 0.0, ignore_endmdl: bool = False,
                 counted_residues=()) -> "Batch":
        """Upload raw PDB texts (bytes) and parse them on the GPU (``tm_batch_create_from_pdb``).
        ``counted_residues``: residue names (3 characters) whose distinct residue numbers are counted per file."""
        lib = load()
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.n_structures = len(texts)
        names, name_ids, els, el_ids, res, res_ids, _ = vocab_tables
        keep = [_arr(names, "S4"), _arr(name_ids, np.uint8), _arr(els, "S2"), _arr(el_ids, np.uint8),
                _arr(res, "S3"), _arr(res_ids, np.uint8)]
        keep.append(_arr(np.array([str(r).rjust(3)[:3].encode() for r in counted_residues], dtype="S3"), "S3"))
        vocab = _Vocab(len(keep[0]), _ptr(keep[0]), _ptr(keep[1]), len(keep[2]), _ptr(keep[2]), _ptr(keep[3]),
                       len(keep[4]), _ptr(keep[4]), _ptr(keep[5]), len(keep[6]), _ptr(keep[6]) if len(keep[6]) else None)
        self.has_residue_counts = len(keep[6]) > 0
        if isinstance(texts, TextBlock):
            keep += [texts.pointers, texts.lengths, texts.buffer]
            ptrs, lens = _ptr(keep[-3]), _ptr(keep[-2])
        else:
            ptrs = (C.c_char_p * max(1, len(texts)))(*texts)
            lens = (C.c_uint64 * max(1, len(texts)))(*[len(t) for t in texts])
        _check(lib.tm_batch_create_from_pdb(C.c_int(device), C.c_uint32(len(texts)), ptrs, lens, C.byref(vocab),
                                            C.c_uint32(TM_PDB_IGNORE_ENDMDL if ignore_endmdl else 0),
                                            C.c_double(min_bfactor), C.byref(self._h)))
        self.device = device
        self.atom_offsets = np.zeros(len(texts) + 1, dtype=np.uint32)
        _check(lib.tm_batch_atom_counts(self._h, _ptr(self.atom_offsets)))
        return self

    @classmethod
    def from_files(cls, device, paths, vocab_tables, min_bfactor: float = 0.0, ignore_endmdl: bool = False,
                   counted_residues=()) -> "Batch":
        """Read the files natively (host threads, one page-locked buffer, one upload) and parse them on the GPU
        (``tm_batch_create_from_files``).  ``file_text(i)`` returns a file's bytes while the batch is open."""
        lib = load()
        self = cls.__new__(cls)
        self._h = C.c_void_p()
        self.n_structures = len(paths)
        names, name_ids, els, el_ids, res, res_ids, _ = vocab_tables
        keep = [_arr(names, "S4"), _arr(name_ids, np.uint8), _arr(els, "S2"), _arr(el_ids, np.uint8),
                _arr(res, "S3"), _arr(res_ids, np.uint8)]
        keep.append(_arr(np.array([str(r).rjust(3)[:3].encode() for r in counted_residues], dtype="S3"), "S3"))
        vocab = _Vocab(len(keep[0]), _ptr(keep[0]), _ptr(keep[1]), len(keep[2]), _ptr(keep[2]), _ptr(keep[3]),
                       len(keep[4]), _ptr(keep[4]), _ptr(keep[5]), len(keep[6]), _ptr(keep[6]) if len(keep[6]) else None)
        self.has_residue_counts = len(keep[6]) > 0
        encoded = [os.fsencode(p) for p in paths]
        arr = (C.c_char_p * max(1, len(encoded)))(*encoded)
        bad = C.c_int64(-1)
        rc = lib.tm_batch_create_from_files(C.c_int(device), C.c_uint32(len(encoded)), arr, C.byref(vocab),
                                            C.c_uint32(TM_PDB_IGNORE_ENDMDL if ignore_endmdl else 0),
                                            C.c_double(min_bfactor), C.byref(self._h), C.byref(bad))
        if rc != TM_OK and bad.value >= 0:
            open(paths[bad.value], "rb").close()  # raises the ordinary OSError of that file, if it still fails
        _check(rc)
        self.device = device
        self.atom_offsets = np.zeros(len(paths) + 1, dtype=np.uint32)
        _check(lib.tm_batch_atom_counts(self._h, _ptr(self.atom_offsets)))
        return self

    def wait(self):
        """Block until the batch is resident (``from_structures`` / ``from_columns`` return with their copies queued;
        queries wait for them on their own stream, so nothing needs to call this)."""
        _check(load().tm_batch_wait(self._h))

    def upload_bytes(self) -> int:
        """Host-to-device bytes the constructor sent (0 for constructors that do not report it)."""
        n = C.c_uint64(0)
        _check(load().tm_batch_upload_bytes(self._h, C.byref(n)))
        return n.value

    def copy_layout(self):
        """Diagnostic: the batch as it lies in HBM -- dict of xyz [n, 3] f8, records [n, 4] f4 (centred xyz + the packed
        word's bits), chain u1, orig u4, residue_offsets [structures + 1] u4, residue_start u2 (``tm_batch_copy_layout``)."""
        lib = load()
        n, ns = int(self.atom_offsets[-1]), self.n_structures
        roff = np.zeros(ns + 1, dtype=np.uint32)
        _check(lib.tm_batch_copy_layout(self._h, None, None, None, None, _ptr(roff), None))
        out = dict(xyz=np.zeros((n, 3), np.float64), records=np.zeros((n, 4), np.float32), chain=np.zeros(n, np.uint8),
                   orig=np.zeros(n, np.uint32), residue_offsets=roff, residue_start=np.zeros(int(roff[-1]), np.uint16))
        _check(lib.tm_batch_copy_layout(self._h, _ptr(out["xyz"]), _ptr(out["records"]), _ptr(out["chain"]), _ptr(out["orig"]),
                                        None, _ptr(out["residue_start"])))
        return out

    def refused(self):
        """``[(structure index, atoms it came with)]``: structures beyond TM_MAX_STRUCTURE_ATOMS.  They stay in the batch
        as empty structures (no hits); the others are searched as usual."""
        lib = load()
        count = C.c_uint32(0)
        _check(lib.tm_batch_refused(self._h, None, None, C.c_uint32(0), C.byref(count)))
        if count.value == 0:
            return []
        idx, atoms = np.zeros(count.value, dtype=np.uint32), np.zeros(count.value, dtype=np.uint32)
        _check(lib.tm_batch_refused(self._h, _ptr(idx), _ptr(atoms), C.c_uint32(count.value), C.byref(count)))
        return list(zip(idx.tolist(), atoms.tolist()))

    def file_text(self, index: int) -> bytes:
        """A private copy of file ``index``'s bytes (batches made by ``from_files``, while open)."""
        text, length = C.c_char_p(), C.c_uint64()
        ptr = C.c_void_p()
        _check(load().tm_batch_file_text(self._h, C.c_uint32(index), C.byref(ptr), C.byref(length)))
        return C.string_at(ptr, length.value)

    def text_view(self):
        """(uint8 view of the native text buffer, file offsets [n + 1]) of a batch made by ``from_files``; the view is
        valid until the batch is closed."""
        base = C.c_void_p()
        offsets = np.zeros(self.n_structures + 1, dtype=np.uint64)
        _check(load().tm_batch_text_layout(self._h, C.byref(base), _ptr(offsets)))
        total = int(offsets[-1])
        view = np.ctypeslib.as_array((C.c_ubyte * max(total, 1)).from_address(base.value)) if total else np.zeros(0, np.uint8)
        return view, offsets

    def line_offsets(self) -> np.ndarray:
        """Byte offset of every kept ATOM/HETATM record inside its file (device-parsed batches)."""
        out = np.zeros(int(self.atom_offsets[-1]), dtype=np.uint32)
        _check(load().tm_batch_line_offsets(self._h, _ptr(out), C.c_uint64(out.size)))
        return out

    def line_offsets_at(self, records: np.ndarray) -> np.ndarray:
        """``line_offsets()[records]`` without copying every offset back: a gather on the device."""
        records = _arr(records, np.uint32)
        out = np.zeros(len(records), dtype=np.uint32)
        _check(load().tm_batch_line_offsets_at(self._h, _ptr(records), C.c_uint64(len(records)), _ptr(out)))
        return out

    def pick_records(self, table: np.ndarray, width: int = 96):
        """(keys uint64 = structure << 32 | atom, lines uint8 [n, width]): the records of the atoms ``table`` (HIT_DTYPE)
        matched, per structure distinct and ascending, copied out of the batch's text buffer by the library
        (``tm_batch_pick_records``; batches made by ``from_files``)."""
        table = np.ascontiguousarray(table)
        cap = int(table["n_atoms"].sum())
        keys = np.empty(cap, dtype=np.uint64)
        lines = np.empty((cap, width), dtype=np.uint8)
        count = C.c_uint64(0)
        _check(load().tm_batch_pick_records(self._h, _ptr(table) if len(table) else None, C.c_uint64(len(table)), C.c_uint32(width),
                                            C.c_uint64(cap), _ptr(keys), _ptr(lines), C.byref(count)))
        return keys[:count.value], lines[:count.value]

    def residue_counts(self) -> np.ndarray:
        """Per file: distinct residue numbers among the atoms of the counted residue names."""
        out = np.zeros(self.n_structures, dtype=np.uint32)
        _check(load().tm_batch_residue_counts(self._h, _ptr(out)))
        return out

    def close(self):
        if getattr(self, "_h", None) and not _finalizing():
            load().tm_batch_destroy(self._h)
            self._h = None

    __del__ = close


class Result:
    """Outcome of one ``tm_query`` launch (owns the device hit table)."""

    def __init__(self, handle, n_structures, n_selected):
        self._h = handle
        self.n_structures, self.n_selected = n_structures, n_selected

    def wait(self):
        _check(load().tm_result_wait(self._h))

    def hit_count(self) -> int:
        n = C.c_uint64(0)
        _check(load().tm_result_hit_count(self._h, C.byref(n)))
        return n.value

    def hits(self) -> np.ndarray:
        """Structured array (HIT_DTYPE) sorted by (structure, template_sel)."""
        n = self.hit_count()
        out = np.zeros(n, dtype=HIT_DTYPE)
        _check(load().tm_result_copy_hits(self._h, _ptr(out), C.c_uint64(n)))
        return out

    def candidates(self) -> np.ndarray:
        out = np.zeros((self.n_structures, self.n_selected), dtype=np.uint32)
        _check(load().tm_result_copy_candidates(self._h, _ptr(out), C.c_uint64(out.size)))
        return out

    def timing(self):
        """(index_ms, search_ms, finish_ms) measured with HIP events on the launch stream."""
        a, b, c = C.c_float(0), C.c_float(0), C.c_float(0)
        _check(load().tm_result_timing(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def search_split(self):
        """(front_ms, deep_ms): the search time of ``timing()`` split into the block search over the pair tables and the
        depth-first part below the surviving triples."""
        a, b = C.c_float(0), C.c_float(0)
        _check(load().tm_result_search_split(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def page_census(self):
        """(most pages of triples one structure listed, triples in all pages) between the two search kernels."""
        a, b = C.c_uint32(0), C.c_uint64(0)
        _check(load().tm_result_page_census(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def needed(self):
        """(candidates, pairs) this query produced -- the capacities a retry after TM_E_OVERFLOW needs."""
        a, b = C.c_uint64(0), C.c_uint64(0)
        _check(load().tm_result_needed(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def unsearched(self):
        """Structure numbers of the batch this query could not search with its library (a wide structure with more than
        65 535 atoms admissible to one template-atom kind): no hits for them, the rest of the batch searched."""
        lib = load()
        count = C.c_uint32(0)
        _check(lib.tm_result_unsearched(self._h, None, C.c_uint32(0), C.byref(count)))
        if count.value == 0:
            return []
        idx = np.zeros(count.value, dtype=np.uint32)
        _check(lib.tm_result_unsearched(self._h, _ptr(idx), C.c_uint32(count.value), C.byref(count)))
        return [int(i) for i in idx[:count.value]]

    def device_hits_ptr(self) -> int:
        return load().tm_result_device_hits(self._h) or 0

    def append_hits(self, rows_ptr: int, capacity: int, count_ptr: int, structure_offset: int = 0):
        """Enqueue (on the query's stream, no host wait) the copy of this query's hit rows into a caller-owned device
        table: ``rows_ptr`` tm_hit[capacity], ``count_ptr`` a uint64 row counter, both device addresses."""
        _check(load().tm_result_append_hits(self._h, C.c_void_p(rows_ptr), C.c_uint64(capacity), C.c_void_p(count_ptr),
                                            C.c_uint32(structure_offset)))

    def wait_appended(self):
        """Waits for the rows of the last ``append_hits`` to be in the table (not for later work on the stream)."""
        _check(load().tm_result_wait_appended(self._h))

    def close(self):
        if getattr(self, "_h", None) and not _finalizing():
            load().tm_result_destroy(self._h)
            self._h = None

    __del__ = close


def query(library: Library, batch: Batch, template_index, rmsd_threshold, distance_cutoff, max_dynamic_distance,
          max_candidates: int, flags: int, hit_capacity: int = 0, stream: int = 0,
          candidate_capacity: int = 0, retry: bool = True, models=None) -> Result:
    """Launch one query.  With ``retry`` the call waits for the result and, if a result table
    overflowed, re-runs with the sizes the first run reported."""
    n = len(template_index)
    keep = [
        _arr(template_index, np.uint32),
        _arr(np.broadcast_to(np.asarray(rmsd_threshold, dtype=np.float64), (n,)), np.float64),
        _arr(np.broadcast_to(np.asarray(distance_cutoff, dtype=np.float64), (n,)), np.float64),
        _arr(np.broadcast_to(np.asarray(max_dynamic_distance, dtype=np.float64), (n,)), np.float64),
    ]
    if models is not None:  # (model_set [n], model_offsets [sets+1], model_votes [sets], models [n_models, 4])
        mk = [_arr(models[0], np.int32), _arr(models[1], np.uint32), _arr(models[2], np.uint32),
              _arr(models[3], np.float64)]
        if len(mk[0]) != n:
            raise ValueError("models[0] needs one entry per selected template")
        keep += mk
        margs = (len(mk[2]), _ptr(mk[0]), _ptr(mk[1]), _ptr(mk[2]), _ptr(mk[3]))
    else:
        margs = (0, None, None, None, None)
    params = _QueryParams(n, *[_ptr(a) for a in keep[:4]], int(max_candidates), int(flags), int(hit_capacity),
                          C.c_void_p(stream or None), int(candidate_capacity), *margs)
    for _ in range(4):
        handle = C.c_void_p()
        _check(load().tm_query(library._h, batch._h, C.byref(params), C.byref(handle)))
        res = Result(handle, batch.n_structures, n)
        if not retry:
            return res
        try:
            res.wait()
            return res
        except NativeError as exc:
            if exc.code != TM_E_OVERFLOW:
                raise
            cands, pairs = res.needed()
            res.close()
            params.candidate_capacity = max(int(1.25 * cands) + 1024, 65536)
            params.hit_capacity = max(int(1.25 * pairs) + 1024, 4096)
    raise NativeError(TM_E_OVERFLOW, "result tables kept overflowing")
