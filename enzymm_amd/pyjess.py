"""Drop-in for the slice of the ``pyjess`` API that enzymm touches, backed by the MI355X matcher.

Mirrors (names, argument meaning, error behaviour) what enzymm uses of ``pyjess ~=0.5``
(/root/reference/pyproject.toml:30):

=================  ===========================================================================
``Jess``           ``Jess(templates)``; ``.query(molecule, rmsd_threshold, distance_cutoff,
                   max_dynamic_distance, *, max_candidates, ignore_chain, best_match)``
                   -- /root/reference/enzymm/jess_run.py:800-811, tests/test_jess_run.py:32-40
``Hit``            ``.rmsd .log_evalue .template .molecule(transform) .atoms(transform)``
                   -- jess_run.py:148-182, 243-267, 357, 397, 485-495
``Atom``           PDB fields + ``!=``           -- jess_run.py:142-145, 365, 441
``Molecule``       ``load(path_or_file, id=)``, ``.id``, ``.conserved(cutoff)``, ``len``, iteration,
                   hashable                       -- jess_run.py:538-556, 915-917, 954-958
``Template``       ``__init__(atoms, id=None)``, ``.id``, ``.dimension``, iterable, subclassable
                   -- enzymm/template.py:455, 535
``TemplateAtom``   ``loads(line)``, keyword constructor, attributes, ``==``/hash
                   -- template.py:679, 258-319, 374-407; tests/test_template.py:234-268
=================  ===========================================================================

Install as a replacement with ``sys.modules["pyjess"] = enzymm_amd.pyjess``.

The search itself runs in ``libenzymm_amd.so`` (hand-written HIP, gfx950).  There is no CPU
fallback: without the library and a GPU, ``Jess.query`` raises.
"""
from __future__ import annotations

import collections
import collections.abc
import ctypes
import io
import math
import os
import threading
import time
from typing import Iterable, Iterator, List, Optional, Sequence

import numpy as np

from . import _native
from ._vocab import ONE_LETTER, VOCAB, raw_atom_name, raw_residue_name

__version__ = "0.5.0+enzymm_amd.1"

__all__ = ["Atom", "Molecule", "TemplateAtom", "Template", "Jess", "Query", "Hit", "__version__"]

_UNLIMITED = 0xFFFFFFFF


_default_device = None  # (read once: Jess objects are built per query in the reference's loop)


def default_device() -> int:
    """GPU ordinal used by this process: $ENZYMM_AMD_DEVICE, else $LOCAL_RANK, else 0 (as they stand at the first use)."""
    global _default_device
    if _default_device is None:
        _default_device = int(os.environ.get("ENZYMM_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    return _default_device


_molecule_created = None  # set once _Sweeps exists (end of this module)


# ---------------------------------------------------------------------------------- Atom


class Atom:
    """One ATOM/HETATM record of a target structure."""

    __slots__ = (
        "serial", "name", "altloc", "residue_name", "chain_id", "residue_number", "insertion_code",
        "x", "y", "z", "occupancy", "temperature_factor", "segment", "element", "charge", "_raw_name",
    )

    def __init__(self, *, serial: int, name: str, residue_name: str, chain_id: str, residue_number: int,
                 x: float, y: float, z: float, altloc: Optional[str] = None, insertion_code: str = " ",
                 occupancy: float = 0.0, temperature_factor: float = 0.0, segment: str = "", element: str = "",
                 charge: int = 0, _raw_name: Optional[bytes] = None):
        self.serial = int(serial)
        self.name = name
        self.altloc = altloc
        self.residue_name = residue_name
        self.chain_id = chain_id
        self.residue_number = int(residue_number)
        self.insertion_code = insertion_code
        self.x, self.y, self.z = float(x), float(y), float(z)
        self.occupancy = float(occupancy)
        self.temperature_factor = float(temperature_factor)
        self.segment = segment
        self.element = element
        self.charge = int(charge)
        self._raw_name = _raw_name if _raw_name is not None else raw_atom_name(name)

    def _key(self):
        return (self.serial, self.name, self.altloc, self.residue_name, self.chain_id, self.residue_number,
                self.insertion_code, self.x, self.y, self.z, self.occupancy, self.temperature_factor,
                self.segment, self.element, self.charge)

    def __eq__(self, other):
        return self._key() == other._key() if isinstance(other, Atom) else NotImplemented

    def __ne__(self, other):
        return self._key() != other._key() if isinstance(other, Atom) else NotImplemented

    def __hash__(self):
        return hash(self._key())

    def __repr__(self):
        return (f"Atom(serial={self.serial}, name={self.name!r}, residue_name={self.residue_name!r}, "
                f"chain_id={self.chain_id!r}, residue_number={self.residue_number}, "
                f"x={self.x}, y={self.y}, z={self.z})")

    def copy(self) -> "Atom":
        new = Atom.__new__(Atom)
        for f in Atom.__slots__:
            setattr(new, f, getattr(self, f))
        return new

    def _moved(self, x, y, z) -> "Atom":
        new = self.copy()
        new.x, new.y, new.z = float(x), float(y), float(z)
        return new


# ---------------------------------------------------------------------------------- Molecule


def _address_of(a: np.ndarray) -> int:
    """Address of a contiguous array's first byte.  Through the buffer protocol where the array is writable (0.7 us;
    ``__array_interface__`` builds a dictionary: 2 us -- six of them per molecule were most of the 17 us a fresh
    ``Molecule`` cost ``make_batch`` under the GIL, round 6)."""
    if a.size:
        try:
            return ctypes.addressof(ctypes.c_char.from_buffer(a))
        except (TypeError, ValueError, BufferError):  # read-only or exotic: the slow way
            pass
    return a.__array_interface__["data"][0]


_MOL_COLUMNS = (
    ("serial", "<i4"), ("name", "S4"), ("altloc", "S1"), ("residue_name", "S3"), ("chain_id", "S2"),
    ("residue_number", "<i4"), ("insertion_code", "S1"), ("occupancy", "<f8"), ("temperature_factor", "<f8"),
    ("segment", "S4"), ("element", "S2"), ("charge", "i1"),
)


_POW10 = (10.0 ** np.arange(19)).astype(np.float64)  # exact up to 1e18 < 2^63... as doubles exact up to 1e22


def _fixed_decimals(fields: np.ndarray, blank_ok: np.ndarray, integer: np.ndarray):
    """Decimal text fields [m, w] (uint8) -> (int64 mantissa with sign, digits after the point, ok flags).

    One pass over the w columns with whole-array operations: [spaces][sign]digits[.digits][spaces].  The
    value is mantissa / 10^frac; both are exact, so the quotient is the correctly rounded double --
    exactly float(text)."""
    m, w = fields.shape
    mant = np.zeros(m, np.int64)
    frac = np.zeros(m, np.int64)
    ndig = np.zeros(m, np.int64)
    seen_dot = np.zeros(m, bool)
    started = np.zeros(m, bool)
    ended = np.zeros(m, bool)
    neg = np.zeros(m, bool)
    bad = np.zeros(m, bool)
    for k in range(w):
        c = fields[:, k]
        is_d = (c >= 48) & (c <= 57)
        is_dot = c == 46
        is_sign = (c == 45) | (c == 43)
        is_sp = c == 32
        bad |= ~(is_d | is_dot | is_sign | is_sp)
        bad |= ended & ~is_sp          # something after the trailing blanks
        bad |= is_sign & started       # a sign in the middle
        bad |= is_dot & seen_dot       # a second point
        ended |= started & is_sp
        mant = np.where(is_d, mant * 10 + (c.astype(np.int64) - 48), mant)
        frac += is_d & seen_dot
        ndig += is_d
        neg |= c == 45
        seen_dot |= is_dot
        started |= is_d | is_dot | is_sign
    ok = ~bad & (ndig <= 18) & ((ndig > 0) | (~started & blank_ok)) & ~(integer & seen_dot)
    return np.where(neg, -mant, mant), frac, ok


def _parse_pdb_bytes(data: bytes, ignore_endmdl: bool):
    """Fixed-column parse of the ATOM/HETATM records of a PDB file (up to the first ENDMDL) by the native host
    reader (``tm_pdb_parse_host``, 30x the numpy version below, which remains as its test reference)."""
    return _native.parse_pdb_host(bytes(data), ignore_endmdl)


def _parse_pdb_bytes_numpy(data: bytes, ignore_endmdl: bool):
    """Vectorised fixed-column parse of the ATOM/HETATM records of a PDB file (up to the first ENDMDL).

    Returns (columns, xyz, HEADER idCode).  Same fields and error behaviour as a line-by-line parse, but
    the text is viewed as one byte matrix [records, 80] and every column is converted in one numpy call."""
    buf = np.frombuffer(data, dtype=np.uint8)
    if len(buf) and buf[-1] != 10:
        buf = np.concatenate([buf, np.array([10], np.uint8)])
    ends = np.flatnonzero(buf == 10)
    starts = np.concatenate([[0], ends[:-1] + 1]) if len(ends) else np.zeros(0, np.int64)
    lens = ends - starts
    if len(lens):
        lens = lens - (buf[np.maximum(ends - 1, 0)] == 13) * (lens > 0)  # drop a trailing \r
    # first six bytes of every line (short lines padded with spaces)
    if len(starts):
        pos6 = starts[:, None] + np.arange(6)[None, :]
        six = np.where(np.arange(6)[None, :] < lens[:, None], buf[np.minimum(pos6, len(buf) - 1)], 32).astype(np.uint8)
        tag = np.ascontiguousarray(six).view("S6").ravel()
    else:
        tag = np.zeros(0, "S6")
    is_atom = (tag == b"ATOM  ") | (tag == b"HETATM")
    if not ignore_endmdl:
        endm = np.flatnonzero(tag == b"ENDMDL")
        if len(endm):
            is_atom[endm[0]:] = False
    header_id = None
    head = np.flatnonzero(tag == b"HEADER")
    if len(head):
        h = head[0]
        if lens[h] >= 66:
            header_id = bytes(buf[starts[h] + 62: starts[h] + 66]).decode("ascii", "replace").strip() or None
    rows_idx = np.flatnonzero(is_atom)
    n = len(rows_idx)
    cols = {k: np.zeros(n, dtype=dt) for k, dt in _MOL_COLUMNS}
    xyz = np.zeros((n, 3), dtype=np.float64)
    if n == 0:
        return cols, xyz, header_id
    st, ln = starts[rows_idx], lens[rows_idx]
    pos = st[:, None] + np.arange(80)[None, :]
    mat = np.where(np.arange(80)[None, :] < ln[:, None], buf[np.minimum(pos, len(buf) - 1)], 32).astype(np.uint8)
    if (ln < 54).any():
        bad = int(np.flatnonzero(ln < 54)[0])
        raise ValueError(f"malformed PDB record: {bytes(mat[bad]).decode('ascii', 'replace').rstrip()!r}")
    # the seven numeric fields of every record in ONE pass: stacked as [7 * n, 8] (narrower fields space-padded)
    spans = ((6, 11), (22, 26), (30, 38), (38, 46), (46, 54), (54, 60), (60, 66))
    stack = np.full((7 * n, 8), 32, np.uint8)
    for f, (lo, hi) in enumerate(spans):
        stack[f * n:(f + 1) * n, : hi - lo] = mat[:, lo:hi]
    blank_ok = np.repeat(np.array([0, 0, 0, 0, 0, 1, 1], bool), n)
    integer = np.repeat(np.array([1, 1, 0, 0, 0, 0, 0], bool), n)
    mant, frac, ok = _fixed_decimals(stack, blank_ok, integer)
    if not ok.all():
        bad = int(np.flatnonzero(~ok)[0]) % n
        raise ValueError(f"malformed PDB record: {bytes(mat[bad]).decode('ascii', 'replace').rstrip()!r}")
    value = mant.astype(np.float64) / _POW10[frac]
    cols["serial"][:] = mant[:n]
    cols["residue_number"][:] = mant[n:2 * n]
    xyz[:, 0], xyz[:, 1], xyz[:, 2] = value[2 * n:3 * n], value[3 * n:4 * n], value[4 * n:5 * n]
    cols["occupancy"][:] = value[5 * n:6 * n]
    cols["temperature_factor"][:] = value[6 * n:]
    take = lambda lo, hi: np.ascontiguousarray(mat[:, lo:hi]).view(f"S{hi - lo}").ravel()  # noqa: E731
    cols["name"][:] = take(12, 16)
    cols["altloc"][:] = take(16, 17)
    cols["residue_name"][:] = take(17, 20)
    cols["chain_id"][:] = take(20, 22)
    cols["insertion_code"][:] = take(26, 27)
    cols["segment"][:] = np.char.strip(take(72, 76))
    cols["element"][:] = np.char.strip(take(76, 78))
    digit, sign = mat[:, 78], mat[:, 79]
    is_digit = (digit >= 48) & (digit <= 57)
    cols["charge"][:] = np.where(is_digit, (digit.astype(np.int16) - 48) * np.where(sign == 45, -1, 1), 0)
    return cols, xyz, header_id


class Molecule:
    """Atom table of one target structure.

    Stored column-wise (numpy) so thousands of molecules flatten into a device batch without
    touching Python objects; ``Atom`` objects are materialised lazily on iteration.
    """

    def __init__(self, atoms: Iterable[Atom] = (), id: Optional[str] = None):
        atoms = list(atoms)
        n = len(atoms)
        cols = {k: np.zeros(n, dtype=dt) for k, dt in _MOL_COLUMNS}
        xyz = np.zeros((n, 3), dtype=np.float64)
        for i, a in enumerate(atoms):
            if not isinstance(a, Atom):
                raise TypeError(f"expected Atom, found {type(a).__name__}")
            cols["serial"][i] = a.serial
            cols["name"][i] = a._raw_name
            cols["altloc"][i] = (a.altloc or " ").encode()
            cols["residue_name"][i] = raw_residue_name(a.residue_name)
            cols["chain_id"][i] = a.chain_id.rjust(2).encode()
            cols["residue_number"][i] = a.residue_number
            cols["insertion_code"][i] = (a.insertion_code or " ").encode()
            cols["occupancy"][i] = a.occupancy
            cols["temperature_factor"][i] = a.temperature_factor
            cols["segment"][i] = a.segment.encode()
            cols["element"][i] = a.element.encode()
            cols["charge"][i] = a.charge
            xyz[i] = (a.x, a.y, a.z)
        self._init_columns(cols, xyz, id)
        self._atoms = [a.copy() for a in atoms]

    def _init_columns(self, cols, xyz, id):
        self._cols = cols
        self._xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        self.id = id
        self._atoms = None
        self._enc = None  # (vocabulary version, encoded arrays)
        self._row = None  # (tm_structure_columns record, the arrays it points into)
        self._lock = threading.Lock()
        if _molecule_created is not None:
            _molecule_created(self)  # (the sweep planner keeps a weak reference: a molecule that was just made is about to be asked for)

    _ROW_COLUMNS = (("name", "S4"), ("residue_name", "S3"), ("chain_id", "S2"), ("residue_number", np.int32),
                    ("insertion_code", "S1"))

    def _pointer_row(self) -> np.ndarray:
        """This molecule as a ``tm_structure_columns`` record (7 x uint64: atom count and the addresses of its xyz /
        name / residue_name / chain_id / residue_number / insertion_code arrays), cached together with the arrays:
        ``make_batch`` hands a list of molecules to the library without copying their columns together first."""
        row = getattr(self, "_row", None)
        if row is None:
            keep = [np.ascontiguousarray(self._xyz, dtype=np.float64)]
            keep += [np.ascontiguousarray(self._cols[k], dtype=dt) for k, dt in self._ROW_COLUMNS]
            rec = np.array([len(self)] + [_address_of(a) for a in keep], dtype=np.uint64)
            row = self._row = (rec, keep)
        return row[0]

    @classmethod
    def from_columns(cls, id: Optional[str] = None, *, xyz, name, residue_name, chain_id, residue_number, **other):
        """Build from arrays: ``name`` raw 4-char PDB fields (``S4``), ``residue_name`` (``S3``),
        ``chain_id`` (``S2``, right-justified) and any other column of the table."""
        n = len(residue_number)
        cols = {k: np.zeros(n, dtype=dt) for k, dt in _MOL_COLUMNS}
        cols["altloc"][:] = b" "
        cols["insertion_code"][:] = b" "
        cols["serial"][:] = np.arange(1, n + 1)
        given = dict(name=name, residue_name=residue_name, chain_id=chain_id, residue_number=residue_number, **other)
        for k, v in given.items():
            if k not in cols:
                raise TypeError(f"unknown column {k!r}")
            cols[k][:] = v
        mol = cls.__new__(cls)
        mol._init_columns(cols, xyz, id)
        return mol

    # -- loading ------------------------------------------------------------
    @classmethod
    def loads(cls, text: str, id: Optional[str] = None, ignore_endmdl: bool = False) -> "Molecule":
        return cls.load(io.StringIO(text), id=id, ignore_endmdl=ignore_endmdl)

    @classmethod
    def load(cls, file, id: Optional[str] = None, ignore_endmdl: bool = False) -> "Molecule":
        """Parse ATOM and HETATM records of a PDB file (path or text file object).

        Reading stops at the first ENDMDL unless ``ignore_endmdl`` (jess_run.py:538-540).  Without an
        explicit ``id`` the HEADER record's idCode is used (tests/test_jess_run.py:34-35, 62).
        """
        if hasattr(file, "read"):
            data = file.read()
        else:
            with open(os.fspath(file), "rb") as handle:
                data = handle.read()
        if isinstance(data, str):
            data = data.encode("ascii", "replace")
        cols, xyz, header_id = _parse_pdb_bytes(data, ignore_endmdl)
        mol = cls.__new__(cls)
        mol._init_columns(cols, xyz, id if id is not None else header_id)
        return mol

    @classmethod
    def load_many(cls, files: Sequence, ids: Optional[Sequence[Optional[str]]] = None,
                  ignore_endmdl: bool = False) -> List["Molecule"]:
        """``load`` for many paths at once: the files are parsed on a team of native host threads
        (``tm_pdb_parse_host_many``).  Same molecules, same error for the first malformed file."""
        tables = _native.parse_pdb_host_many(_native.read_files([os.fspath(f) for f in files]), ignore_endmdl)
        out = []
        for i, (cols, xyz, header_id) in enumerate(tables):
            mol = cls.__new__(cls)
            mol._init_columns(cols, xyz, ids[i] if ids is not None and ids[i] is not None else header_id)
            out.append(mol)
        cls._share_rows(out)
        return out

    @classmethod
    def _share_rows(cls, molecules: Sequence["Molecule"]) -> None:
        """The hand-over records (``_pointer_row``) of molecules whose tables are consecutive slices of the SAME arrays -- what
        ``parse_pdb_host_many`` returns -- in one vectorised step: seven base addresses per call instead of six address
        look-ups per molecule (12 us each under the GIL when ``make_batch`` first meets the molecule, round 6).  Molecules
        that do not share their arrays keep making their record at first use."""
        if len(molecules) < 2:
            return
        first = molecules[0]
        arrays = [first._xyz] + [first._cols[k] for k, _ in cls._ROW_COLUMNS]
        bases = [a.base if a.base is not None else a for a in arrays]
        want = [np.dtype(np.float64)] + [np.dtype(dt) for _, dt in cls._ROW_COLUMNS]
        if any(b is None or not b.flags.c_contiguous or b.dtype != w for b, w in zip(bases, want)):
            return
        step = [24] + [w.itemsize for w in want[1:]]  # bytes per atom: three doubles, then one item per column
        base_addr = [_address_of(b) for b in bases]
        n = len(molecules)
        counts = np.fromiter((len(m) for m in molecules), dtype=np.uint64, count=n)
        rows = np.empty((n, 7), dtype=np.uint64)
        rows[:, 0] = counts
        # consecutive slices in list order (what load_many makes): every molecule starts where the one before it ends --
        # checked on every molecule's coordinate view (same base array) and, by address, on the last non-empty one
        starts = np.zeros(n, dtype=np.uint64)
        np.cumsum(counts[:-1], out=starts[1:])
        if any(m._xyz.base is not bases[0] for m in molecules):
            return  # (not views of the first molecule's arrays: every record is left to its first use)
        for i in range(n - 1, -1, -1):
            if counts[i]:
                if _address_of(molecules[i]._xyz) != base_addr[0] + 24 * int(starts[i]):
                    return
                break
        for j in range(6):
            rows[:, j + 1] = np.uint64(base_addr[j]) + starts * np.uint64(step[j])
        for i, m in enumerate(molecules):
            # (the views keep the shared arrays alive; an empty molecule hands over count 0 and is never dereferenced)
            m._row = (rows[i], arrays)

    # -- container protocol -------------------------------------------------
    def _atom(self, i: int) -> Atom:
        c = self._cols
        altloc = bytes(c["altloc"][i]).decode() or " "
        return Atom(
            serial=int(c["serial"][i]), name=bytes(c["name"][i]).decode().strip(),
            altloc=None if altloc == " " else altloc,
            residue_name=bytes(c["residue_name"][i]).decode().strip(),
            chain_id=bytes(c["chain_id"][i]).decode().strip(), residue_number=int(c["residue_number"][i]),
            insertion_code=bytes(c["insertion_code"][i]).decode() or " ",
            x=self._xyz[i, 0], y=self._xyz[i, 1], z=self._xyz[i, 2], occupancy=c["occupancy"][i],
            temperature_factor=c["temperature_factor"][i], segment=bytes(c["segment"][i]).decode(),
            element=bytes(c["element"][i]).decode(), charge=int(c["charge"][i]),
            _raw_name=bytes(c["name"][i]).ljust(4),
        )

    def _atom_list(self) -> List[Atom]:
        if self._atoms is None:
            self._atoms = [self._atom(i) for i in range(len(self))]
        return self._atoms

    def __len__(self):
        return len(self._xyz)

    def __bool__(self):
        return len(self._xyz) > 0

    def __iter__(self) -> Iterator[Atom]:
        return iter(self._atom_list())

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self._subset(np.arange(len(self))[i])
        return self._atom_list()[i]

    def __repr__(self):
        return f"Molecule(id={self.id!r}, atoms={len(self)})"

    def _subset(self, keep) -> "Molecule":
        mol = Molecule.__new__(Molecule)
        mol._init_columns({k: v[keep] for k, v in self._cols.items()}, self._xyz[keep], self.id)
        return mol

    def conserved(self, cutoff: float = 0.0) -> "Molecule":
        """New molecule keeping atoms whose temperature factor is >= ``cutoff``."""
        return self._subset(self._cols["temperature_factor"] >= cutoff)

    def copy(self) -> "Molecule":
        return self._subset(np.arange(len(self)))

    # -- flattening for the device -------------------------------------------
    def _encoded(self):
        """SoA view of the molecule for tm_batch_create, cached per vocabulary version."""
        with self._lock:
            if self._enc is not None and self._enc[0] == VOCAB.version:
                return self._enc[1]
            c = self._cols
            n = len(self)
            chain16 = c["chain_id"].view(np.uint8).reshape(n, 2).astype(np.int64) if n else np.zeros((0, 2), np.int64)
            icode = c["insertion_code"].view(np.uint8).astype(np.int64) if n else np.zeros(0, np.int64)
            # one integer per (chain, resSeq, iCode): equal keys <=> same residue
            key = ((chain16[:, 0] << 8 | chain16[:, 1]) << 40) | ((c["residue_number"].astype(np.int64) + (1 << 24)) << 8) | icode
            _, residue = np.unique(key, return_inverse=True)
            _, chain = np.unique(chain16[:, 0] << 8 | chain16[:, 1], return_inverse=True)
            enc = dict(xyz=self._xyz, res_type=VOCAB.encode_residues(c["residue_name"]),
                       name_id=VOCAB.encode_names(c["name"]), residue=residue.astype(np.uint32),
                       chain=(chain & 0xFF).astype(np.uint8))
            self._enc = (VOCAB.version, enc)
            return enc


# ---------------------------------------------------------------------------------- TemplateAtom


class TemplateAtom:
    """One atom line of a Jess template."""

    __slots__ = ("chain_id", "residue_number", "residue_names", "atom_names", "distance_weight", "match_mode",
                 "x", "y", "z", "_raw_names", "_raw_residues")

    def __init__(self, *, chain_id: str, residue_number: int, x: float, y: float, z: float,
                 residue_names: Sequence[str], atom_names: Sequence[str], distance_weight: float = 0.0,
                 match_mode: int = 0):
        if not atom_names:
            raise ValueError("a template atom needs at least one atom name")
        if not residue_names:
            raise ValueError("a template atom needs at least one residue name")
        for nm in atom_names:
            if not 0 < len(nm) <= 4:
                raise ValueError(f"invalid atom name {nm!r}")
        for nm in residue_names:
            if not 0 < len(nm) <= 3:
                raise ValueError(f"invalid residue name {nm!r}")
        self.chain_id = chain_id
        self.residue_number = int(residue_number)
        self.residue_names = tuple(residue_names)
        self.atom_names = tuple(atom_names)
        self.distance_weight = float(distance_weight)
        self.match_mode = int(match_mode)
        self.x, self.y, self.z = float(x), float(y), float(z)
        self._raw_names = tuple(raw_atom_name(n) for n in self.atom_names)
        self._raw_residues = tuple(raw_residue_name(n) for n in self.residue_names)

    @classmethod
    def loads(cls, text: str) -> "TemplateAtom":
        """Parse a template ATOM line (SURVEY.md 8(a6) column layout)."""
        line = text.rstrip("\r\n")
        if not line.startswith("ATOM"):
            raise ValueError(f"not a template ATOM line: {line!r}")
        if len(line) < 54:
            raise ValueError(f"template ATOM line too short: {line!r}")
        line = line.ljust(66)
        try:
            mode = int(line[6:11])
            resnum = int(line[22:26])
            x, y, z = float(line[30:38]), float(line[38:46]), float(line[46:54])
            weight_txt = line[60:66].strip()
            weight = float(weight_txt) if weight_txt else 0.0
        except ValueError as exc:
            raise ValueError(f"malformed template ATOM line: {line!r}") from exc
        name = line[12:16]
        if not name.strip():
            raise ValueError(f"template ATOM line without an atom name: {line!r}")
        residues = [line[17:20].strip()]
        for letter in line[54:60].strip():
            if letter not in ONE_LETTER:
                raise ValueError(f"unknown one-letter residue code {letter!r} in {line!r}")
            if ONE_LETTER[letter] not in residues:
                residues.append(ONE_LETTER[letter])
        atom = cls(chain_id=line[20:22].strip(), residue_number=resnum, x=x, y=y, z=z, residue_names=residues,
                   atom_names=[name.strip()], distance_weight=weight, match_mode=mode)
        atom._raw_names = (name.encode("ascii"),)
        return atom

    def _key(self):
        return (self.chain_id, self.residue_number, self.residue_names, self.atom_names, self.distance_weight,
                self.match_mode, self.x, self.y, self.z)

    def __eq__(self, other):
        return self._key() == other._key() if isinstance(other, TemplateAtom) else NotImplemented

    def __hash__(self):
        return hash(self._key())

    def __repr__(self):
        return (f"TemplateAtom(chain_id={self.chain_id!r}, residue_number={self.residue_number}, "
                f"residue_names={list(self.residue_names)!r}, atom_names={list(self.atom_names)!r}, "
                f"match_mode={self.match_mode}, x={self.x}, y={self.y}, z={self.z})")

    def copy(self) -> "TemplateAtom":
        new = TemplateAtom.__new__(TemplateAtom)
        for f in TemplateAtom.__slots__:
            setattr(new, f, getattr(self, f))
        return new


# ---------------------------------------------------------------------------------- Template


class Template:
    """A Jess template: an ordered list of ``TemplateAtom``.  Meant to be subclassed
    (enzymm/template.py:455 adds residues, cluster, EC/CATH ... on top).

    The atoms of a constructed template must not be changed afterwards (PyJess's ``TemplateAtom`` is immutable; this
    mirror's is a plain object for convenience): what the device needs of them is taken when the template is made."""

    def __init__(self, atoms: Iterable[TemplateAtom] = (), id: Optional[str] = None):
        self._atoms = tuple(atoms)
        for a in self._atoms:
            if not isinstance(a, TemplateAtom):
                raise TypeError(f"expected TemplateAtom, found {type(a).__name__}")
        self.id = id
        # What the device library needs of this template, made here, once: PyJess builds the C template when a
        # ``Template`` is constructed and ``Jess(templates)`` only collects them; round 5 does the same -- ``Jess`` on a list
        # of 6780 templates used to walk their 90 543 atoms in Python (0.18 s before the first query of a process).
        self._rows = _template_rows(self._atoms)

    @classmethod
    def loads(cls, text: str, id: Optional[str] = None) -> "Template":
        return cls.load(io.StringIO(text), id=id)

    @classmethod
    def load(cls, file, id: Optional[str] = None) -> "Template":
        if hasattr(file, "read"):
            lines = file.read().splitlines()
        else:
            with open(os.fspath(file)) as handle:
                lines = handle.read().splitlines()
        atoms = [TemplateAtom.loads(l) for l in lines if l.startswith("ATOM")]
        if id is None:
            for l in lines:
                tok = l.split()
                if len(tok) >= 3 and tok[0] == "REMARK" and tok[1] in ("PDB_ID", "ID"):
                    id = tok[2]
                    break
        return cls(atoms, id=id)

    def __len__(self):
        return len(self._atoms)

    def __iter__(self):
        return iter(self._atoms)

    def __getitem__(self, i):
        return self._atoms[i]

    def __eq__(self, other):
        if isinstance(other, Template):
            return self.id == other.id and self._atoms == other._atoms
        return NotImplemented

    def __hash__(self):
        return hash((self.id, self._atoms))

    def copy(self) -> "Template":
        return Template(self._atoms, id=self.id)

    # What a template caches for the device is process state: ``_rows[6]`` numbers the atoms' keys in THIS process's
    # ``_atom_kind_list``, ``_vocab_version`` says the names were registered with THIS process's vocabulary.  Neither may
    # travel in a pickle (spawned workers, a template cache on disk): the rows are made again where the template arrives,
    # as PyJess rebuilds its C template when a ``Template`` is unpickled.
    _PROCESS_STATE = ("_rows", "_orient_rows", "_vocab_version")

    def __getstate__(self):
        state = dict(self.__dict__)
        for name in Template._PROCESS_STATE:
            state.pop(name, None)
        return state

    def __setstate__(self, state):
        self.__dict__.update(state)
        for name in Template._PROCESS_STATE:  # (a state written by an older build that kept them)
            self.__dict__.pop(name, None)
        self._rows = _template_rows(self._atoms)

    @property
    def dimension(self) -> int:
        """Number of template residues (distinct chain + residue number)."""
        d = self.__dict__.get("_dimension") if hasattr(self, "__dict__") else None
        if d is None:
            d = len({(a.chain_id, a.residue_number) for a in self._atoms})
            try:
                self._dimension = d  # the atoms of a template never change
            except AttributeError:
                pass
        return d


# ---------------------------------------------------------------------------------- packing


_atom_kinds: dict = {}          # (match_mode, raw names, raw residues) -> number, in order of first appearance
_atom_kind_list: list = []      # ... and back
_atom_kinds_lock = threading.Lock()
_PROCESS_TOKEN = object()       # rows carry it: kind numbers made in another process (a copied __dict__) are not trusted


def _kind_numbers(keys) -> np.ndarray:
    """Process-wide numbers of the atoms' (match_mode, raw names, raw residues) keys: a library's masks are then two
    table look-ups over the concatenated numbers instead of a dictionary access per atom."""
    out = np.empty(len(keys), dtype=np.uint32)
    get = _atom_kinds.get
    for i, key in enumerate(keys):
        n = get(key)
        if n is None:
            with _atom_kinds_lock:
                n = _atom_kinds.get(key)
                if n is None:
                    n = len(_atom_kind_list)
                    _atom_kind_list.append(key)
                    _atom_kinds[key] = n
        out[i] = n
    return out


def _template_rows(atoms) -> tuple:
    """(xyz [k, 3] f8, weight [k] f8, any-residue [k] u1, group [k] u1, chain [k] u1, keys, kind numbers [k] u4): the
    per-atom rows of one template as the device library takes them -- group / chain = ordinal of (chain, resSeq) / chain
    id by first appearance, keys = (match_mode, raw names, raw residues) per atom, what the vocabulary turns into masks."""
    k = len(atoms)
    xyz = np.array([(a.x, a.y, a.z) for a in atoms], dtype=np.float64).reshape(k, 3)
    weight = np.array([a.distance_weight for a in atoms], dtype=np.float64)
    groups, chains = {}, {}
    group = np.array([groups.setdefault((a.chain_id, a.residue_number), len(groups)) for a in atoms], dtype=np.uint8)
    chain = np.array([chains.setdefault(a.chain_id, len(chains)) for a in atoms], dtype=np.uint8)
    keys = tuple((a.match_mode, a._raw_names, a._raw_residues) for a in atoms)
    any_res = np.array([1 if key[0] >= 100 else 0 for key in keys], dtype=np.uint8)
    return xyz, weight, any_res, group, chain, keys, _kind_numbers(keys), _PROCESS_TOKEN


def _rows_of(t):
    """The template's rows if they were made in this process for the atoms it has, else None."""
    r = getattr(t, "_rows", None)
    if r is None or len(r) != 8 or r[7] is not _PROCESS_TOKEN or len(r[5]) != len(t._atoms):
        return None
    return r


class _PackedLibrary:
    """Host tables + device handle of a list of templates (what ``Jess(templates)`` builds)."""

    def __init__(self, templates: Sequence[Template], device: int):
        if hasattr(templates, "device_tables"):  # a library_pack.PackedTemplates: flat arrays, no per-atom objects
            tables = templates.device_tables(VOCAB)
            self.vocab_version = VOCAB.version
            self.offsets = np.ascontiguousarray(tables[0], dtype=np.uint32)
            self.tables = tuple(tables[:9])
            self.native = _native.Library(device, *tables[:8], orientation=tables[8])
            self.device = device
            _settle_library_objects(len(self.offsets) - 1)
            return
        rows = []
        for t in templates:
            if len(t._atoms) == 0:
                raise ValueError("cannot search an empty template")
            r = _rows_of(t)
            if r is None:  # (an object made without Template.__init__, or rows that came out of another process)
                r = _template_rows(t._atoms)
                try:
                    t._rows = r
                except AttributeError:
                    pass
            rows.append(r)
        self.offsets = np.zeros(len(rows) + 1, dtype=np.uint32)
        np.cumsum([len(r[5]) for r in rows], out=self.offsets[1:])
        cat = lambda c, dt, shape: np.concatenate([r[c] for r in rows]) if rows else np.zeros(shape, dtype=dt)
        xyz, weight = cat(0, np.float64, (0, 3)), cat(1, np.float64, 0)
        any_res, group, chain = cat(2, np.uint8, 0), cat(3, np.uint8, 0), cat(4, np.uint8, 0)
        kinds = cat(6, np.uint32, 0)
        distinct = np.unique(kinds)
        for n in distinct.tolist():  # every name first, the masks after that (a mask names every id of its element)
            VOCAB.register_template_atom(*_atom_kind_list[n])
        self.vocab_version = VOCAB.version
        res_table = np.zeros(len(_atom_kind_list), dtype=np.uint32)
        name_table = np.zeros(len(_atom_kind_list), dtype=np.uint64)
        for n in distinct.tolist():
            key = _atom_kind_list[n]
            res_table[n] = VOCAB.res_mask(key[0], key[2])
            name_table[n] = VOCAB.name_mask(key[0], key[1])
        res_mask, name_mask = res_table[kinds], name_table[kinds]
        orientation = self._orientation_table(templates)
        # (the host tables are kept: a library over several template lists is put together from theirs, `concat`)
        self.tables = (self.offsets, xyz, weight, res_mask, name_mask, any_res, group, chain, orientation)
        self.native = _native.Library(device, self.offsets, xyz, weight, res_mask, name_mask, any_res, group, chain,
                                      orientation=orientation)
        self.device = device
        _settle_library_objects(len(templates))

    @classmethod
    def concat(cls, parts: Sequence["_PackedLibrary"], device: int) -> "_PackedLibrary":
        """One device library holding the templates of several packed lists, one list after the other: the per-atom
        tables are concatenated (no walk over Template objects).  The parts must have been packed under the current
        vocabulary and agree on whether they carry residue-orientation tables."""
        self = cls.__new__(cls)
        self.vocab_version = VOCAB.version
        if any(p.vocab_version != VOCAB.version for p in parts):
            raise ValueError("parts packed under another state of the vocabulary")
        with_orientation = [p.tables[8] is not None for p in parts]
        if any(with_orientation) and not all(with_orientation):
            raise ValueError("only some of the parts carry orientation tables")
        offs, shift = [np.zeros(1, dtype=np.uint32)], 0
        for p in parts:
            o = np.asarray(p.tables[0], dtype=np.uint32)
            offs.append(o[1:] + np.uint32(shift))
            shift += int(o[-1])
        self.offsets = np.concatenate(offs)
        cols = [np.concatenate([np.asarray(p.tables[c]) for p in parts]) for c in range(1, 8)]
        orientation = None
        if all(with_orientation) and parts:
            ro, rshift = [np.zeros(1, dtype=np.uint32)], 0
            for p in parts:
                o = np.asarray(p.tables[8][0], dtype=np.uint32)
                ro.append(o[1:] + np.uint32(rshift))
                rshift += int(o[-1])
            orientation = (np.concatenate(ro),) + tuple(np.concatenate([np.asarray(p.tables[8][c]) for p in parts]) for c in (1, 2, 3))
        self.tables = (self.offsets, *cols, orientation)
        self.native = _native.Library(device, self.offsets, *cols, orientation=orientation)
        self.device = device
        return self


    @staticmethod
    def _orientation_table(templates):
        """Residue-orientation definitions of enzymm-style templates (objects with ``.residues`` whose
        members expose ``orientation_vector`` / ``orientation_vector_indices``, template.py:409-422) for the
        fused hit epilogue; None when any template lacks them.  A template's rows are made once and kept on it
        (``_orient_rows``: its residues never change)."""
        first, second, vec, counts = [], [], [], []
        for t in templates:
            rows = getattr(t, "_orient_rows", None)
            if rows is None:
                rows = _PackedLibrary._orientation_rows(t)
                try:
                    t._orient_rows = rows
                except AttributeError:
                    pass
            if rows is False:
                return None
            first.append(rows[0]); second.append(rows[1]); vec.append(rows[2]); counts.append(len(rows[0]))
        offsets = np.zeros(len(counts) + 1, dtype=np.uint32)
        np.cumsum(counts, out=offsets[1:])
        if not counts:
            return offsets, np.zeros(0, np.uint8), np.zeros(0, np.uint8), np.zeros((0, 3), np.float64)
        return offsets, np.concatenate(first), np.concatenate(second), np.concatenate(vec).reshape(-1, 3)

    @staticmethod
    def _orientation_rows(t):
        """(first u1 [r], second u1 [r], vectors f8 [r, 3]) of one template, or False when it has no usable residues."""
        residues = getattr(t, "residues", None)
        if residues is None or 3 * len(residues) != len(t._atoms):
            return False
        flat = [a for r in residues for a in r.atoms]
        if any(x is not y for x, y in zip(flat, t._atoms)):
            return False
        first, second, vec = [], [], []
        for r in residues:
            i, j = r.orientation_vector_indices
            v = r.orientation_vector
            first.append(i); second.append(j); vec.append((v.x, v.y, v.z))
        return (np.array(first, dtype=np.uint8), np.array(second, dtype=np.uint8), np.array(vec, dtype=np.float64).reshape(-1, 3))


_settled = False


def _settle_library_objects(n_templates: int) -> None:
    """A template library is a quarter of a million small Python objects that live as long as the process.  Every full
    collection of the cyclic garbage collector walks them -- 250-300 ms with the GIL held, measured in the middle of a
    50 ms sweep of the reference's call pattern (and of Matcher.run_files, which pauses the collector for that reason).
    Once a large library has been packed for the device, the objects alive at that moment are moved to the collector's
    permanent generation (``gc.freeze``): they are still freed when their reference counts drop to zero, they are just
    no longer walked.  ``ENZYMM_AMD_GC_FREEZE=0`` leaves the collector alone."""
    global _settled
    if _settled or n_templates < 512 or os.environ.get("ENZYMM_AMD_GC_FREEZE", "1") == "0":
        return
    import gc
    import logging
    _settled = True
    gc.collect()
    gc.freeze()
    # (a log record, not a warning: this is the intended default of every run with the full library, and the reference's
    # CLI shows warnings to its users -- ADVICE r05)
    logging.getLogger("enzymm_amd").info(
        "a library of %d templates was packed; the objects alive now were moved to the garbage collector's permanent "
        "generation (gc.freeze) so that full collections stop walking them (ENZYMM_AMD_GC_FREEZE=0 leaves the "
        "collector alone)", n_templates)


_pack_lock = threading.Lock()
_pack_cache: "collections.OrderedDict[tuple, tuple]" = collections.OrderedDict()
_PACK_CACHE_SIZE = 32


def _packed(templates: Sequence[Template], device: int) -> _PackedLibrary:
    """enzymm rebuilds ``Jess(templates)`` for every (molecule, size) call (jess_run.py:800);
    the device copy of a template list is therefore cached by identity of its members."""
    key = (device, "pack", id(templates)) if hasattr(templates, "device_tables") \
        else (device, tuple(id(t) for t in templates))
    with _pack_lock:
        hit = _pack_cache.get(key)
        if hit is not None and hit[1].vocab_version == VOCAB.version:
            _pack_cache.move_to_end(key)
            return hit[1]
        packed = _PackedLibrary(templates, device)
        keep_alive = templates if hasattr(templates, "device_tables") else tuple(templates)
        _pack_cache[key] = (keep_alive, packed)  # keeps the templates alive: ids stay unique
        while len(_pack_cache) > _PACK_CACHE_SIZE:
            _pack_cache.popitem(last=False)
        return packed


def make_batch(molecules: Sequence[Molecule], device: int) -> "_native.Batch":
    """Flatten molecules into one device-resident batch.  The library reads every molecule's raw columns where they
    lie, encodes them with the vocabulary and packs them on host threads (``tm_batch_create_from_structures``)."""
    tables = VOCAB.native_tables()
    rows = np.empty((len(molecules), 7), dtype=np.uint64)
    for i, m in enumerate(molecules):
        rows[i] = m._pointer_row()
    batch = _native.Batch.from_structures(device, rows, tables)
    batch.vocab_version = tables[-1]  # ids are only meaningful against this state of the vocabulary
    # (a molecule beyond the atom limit or with more residues than the kernels number stays in the batch as an empty
    # structure; only a molecule with more atoms than the residue limit can be either)
    batch.refused_list = batch.refused() if any(len(m) > _native.TM_MAX_STRUCTURE_RESIDUES for m in molecules) else []
    warn_refused([(molecules[i].id, n) for i, n in batch.refused_list])
    return batch


def warn_refused(refused) -> None:
    """Structures beyond the engine's limits -- more than TM_MAX_STRUCTURE_ATOMS atoms or TM_MAX_STRUCTURE_RESIDUES
    residues -- are kept in their batch as empty structures: they get no hits, the rest of the batch is searched as usual
    (``tm_batch_refused``).  ``refused``: ``[(id or path, atoms)]``."""
    if refused:
        import warnings
        listed = ", ".join(f"{ident} ({atoms} atoms)" for ident, atoms in refused[:8]) + (" ..." if len(refused) > 8 else "")
        warnings.warn(f"{len(refused)} structure(s) with more than {_native.TM_MAX_STRUCTURE_ATOMS} atoms or "
                      f"{_native.TM_MAX_STRUCTURE_RESIDUES} residues are not searched: {listed}")


def warn_unsearched(unsearched) -> None:
    """Wide structures (more than 65 535 atoms) with more than 65 535 atoms admissible to ONE template-atom kind cannot
    be searched with that library (``tm_result_unsearched``): they get no hits, the rest of their batch is searched."""
    if unsearched:
        import warnings
        listed = ", ".join(f"{ident} ({atoms} atoms)" for ident, atoms in unsearched[:8]) + (" ..." if len(unsearched) > 8 else "")
        warnings.warn(f"{len(unsearched)} structure(s) with more than 65535 atoms admissible to one template atom kind are not "
                      f"searched with these templates: {listed}")


class _SingleBatches:
    """Device copies of molecules that are queried one at a time, kept between queries (most recently used last) up to
    a budget of atoms: enzymm's call pattern asks for the same molecule once per template size.  An entry is lent to
    one query at a time (a tm_batch carries the candidate index of its last query); a second concurrent query of the
    same molecule builds a batch of its own."""

    BUDGET_ATOMS = 48_000_000  # about 2.5 GB of HBM

    def __init__(self):
        self._lock = threading.Lock()
        self._entries: "collections.OrderedDict[tuple, list]" = collections.OrderedDict()
        self._atoms = 0

    def borrow(self, molecule, device):
        import weakref
        key = (id(molecule), device)
        with self._lock:
            e = self._entries.get(key)
            if e is not None and (e[0]() is not molecule or e[1].vocab_version != VOCAB.version):
                self._drop(key)  # another object at that address now, or names registered since
                e = None
            if e is not None:
                if e[2]:
                    return None, None  # on loan
                e[2] = True
                self._entries.move_to_end(key)
                return e[1], lambda e=e: self._give_back(e)
        batch = make_batch([molecule], device)
        with self._lock:
            if key in self._entries:  # another thread was faster: this one stays private
                return batch, None
            try:
                ref = weakref.ref(molecule)
            except TypeError:
                return batch, None
            e = self._entries[key] = [ref, batch, True, len(molecule), False]
            self._atoms += len(molecule)
            while self._atoms > self.BUDGET_ATOMS and len(self._entries) > 1:
                oldest = next(iter(self._entries))
                if self._entries[oldest][2]:
                    break
                self._drop(oldest)
        return batch, lambda e=e: self._give_back(e)

    def _give_back(self, e):
        """The borrower of entry ``e`` (that very list, not whatever sits under its key by now) is done with it."""
        with self._lock:
            e[2] = False
            if e[4]:  # dropped while on loan: closed now
                e[1].close()

    def _drop(self, key):
        e = self._entries.pop(key)
        self._atoms -= e[3]
        if e[2]:
            e[4] = True  # on loan: its borrower closes it when it hands it back
        else:
            e[1].close()


_single_batches = _SingleBatches()


def _single_batch(molecule, device):
    """(batch, release) for a one-molecule query: a kept device copy when there is one to lend (release hands it
    back), else a fresh batch the caller owns (release None)."""
    batch, release = _single_batches.borrow(molecule, device)
    if batch is None:
        return make_batch([molecule], device), None
    return batch, release


# ---------------------------------------------------------------------------------- Hit / Query / Jess


class Hit:
    """Best superposition of one template on one molecule."""

    __slots__ = ("template", "_molecule", "rmsd", "_rot", "_trans", "_atom_index", "n_candidates", "truncated",
                 "orientation", "predicted")

    def __init__(self, template, molecule, rmsd, rot, trans, atom_index, n_candidates=0, truncated=False,
                 orientation=math.nan, predicted=None):
        self.template = template
        self._molecule = molecule
        self.rmsd = float(rmsd)
        self._rot = np.asarray(rot, dtype=np.float64).reshape(3, 3)
        self._trans = np.asarray(trans, dtype=np.float64).reshape(3)
        self._atom_index = tuple(int(i) for i in atom_index)
        self.n_candidates = int(n_candidates)
        self.truncated = bool(truncated)
        #: mean residue-orientation angle computed in the hit epilogue on the GPU (NaN if the template
        #: carries no orientation definitions); ``predicted`` is the fused logistic filter's verdict or None
        self.orientation = float(orientation)
        self.predicted = predicted

    @property
    def determinant(self) -> float:
        return float(np.linalg.det(self._rot))

    @property
    def log_evalue(self) -> float:
        """Jess's log E-value model lives in the absent pyjess/Jess C source and is pinned by only two
        known answers (tests/test_jess_run.py:78, 134); it is reported as NaN rather than invented."""
        return math.nan

    @property
    def evalue(self) -> float:
        return math.nan

    def _transform(self, atom: Atom) -> Atom:
        r, t = self._rot, self._trans
        x, y, z = atom.x, atom.y, atom.z
        return atom._moved(r[0, 0] * x + r[0, 1] * y + r[0, 2] * z + t[0],
                           r[1, 0] * x + r[1, 1] * y + r[1, 2] * z + t[1],
                           r[2, 0] * x + r[2, 1] * y + r[2, 2] * z + t[2])

    def atoms(self, transform: bool = False) -> List[Atom]:
        """Matched target atoms in template-atom order; ``transform`` maps them into the template frame."""
        mol = self._molecule
        # only the matched atoms are materialised (a full atom list would cost far more than the query)
        if mol._atoms is None and hasattr(mol, "_atoms_at"):  # a lazily parsed molecule: only the matched lines
            picked = mol._atoms_at(self._atom_index)
        else:
            picked = [mol._atoms[i] if mol._atoms is not None else mol._atom(i) for i in self._atom_index]
        return [self._transform(a) for a in picked] if transform else [a.copy() for a in picked]

    def molecule(self, transform: bool = False) -> Molecule:
        if not transform:
            return self._molecule
        mol = self._molecule.copy()
        mol._xyz = self._molecule._xyz @ self._rot.T + self._trans
        return mol

    def __repr__(self):
        return f"Hit(template={getattr(self.template, 'id', None)!r}, rmsd={self.rmsd:.4f})"


class _Ready:
    """Result of a query that needed no launch."""

    unsearched = ()
    refused = ()

    def __init__(self, value):
        self._value = value

    def result(self):
        return self._value


class _PendingQuery:
    """A query in flight on the GPU (``Jess.query_batch(..., _async=True)``): the launch has been issued, ``result()``
    waits for it and builds what ``query_batch`` returns.  If a result table overflowed, the query is repeated with the
    sizes the first run reported."""

    def __init__(self, jess, molecules, sel, args, models, owned_batch, count_candidates, return_raw, stream=0,
                 release=None):
        self._jess, self._molecules, self._sel, self._args, self._models = jess, molecules, sel, args, models
        self._owned_batch, self._count, self._raw, self._stream = owned_batch, count_candidates, return_raw, stream
        self._release = release  # hands a cached single-molecule batch back (see _single_batch)
        self.unsearched = []
        self.refused = list(getattr(args[1], "refused_list", ()))  # [(structure, atoms)] the batch was made without
        self._res = _native.query(*args, models=models, retry=False, stream=stream)

    def result(self):
        try:
            try:
                table = self._res.hits()
            except _native.NativeError as exc:
                if exc.code != _native.TM_E_OVERFLOW:
                    raise
                cands, pairs = self._res.needed()
                self._res.close()
                self._res = _native.query(*self._args, models=self._models, retry=True, stream=self._stream,
                                          candidate_capacity=max(int(1.25 * cands) + 1024, 65536),
                                          hit_capacity=max(int(1.25 * pairs) + 1024, 4096))
                table = self._res.hits()
            cands = self._res.candidates() if self._count else None
            # structures this library cannot search (a kind list beyond 16-bit positions): no hits, reported, the rest searched
            self.unsearched = self._res.unsearched()
            if self.unsearched:
                warn_unsearched([(getattr(self._molecules[k], "id", None) or k, len(self._molecules[k])) for k in self.unsearched])
        finally:
            self._res.close()
            if self._owned_batch is not None:
                self._owned_batch.close()
            if self._release is not None:
                self._release()
                self._release = None
        self.table = table  # the raw rows, for callers that post-process in bulk (ingest.PdbBatch.annotate_hits)
        if self._raw:
            return (table, cands) if self._count else table
        return self._jess._hits_from_table(table, self._sel, self._molecules)


class Query(collections.abc.Iterator):
    """Iterator over the hits of one ``Jess.query`` call, in template order."""

    def __init__(self, jess, molecule, rmsd_threshold, distance_cutoff, max_dynamic_distance, max_candidates,
                 ignore_chain, best_match, hits):
        self.jess = jess
        self.molecule = molecule
        self.rmsd_threshold = rmsd_threshold
        self.distance_cutoff = distance_cutoff
        self.max_dynamic_distance = max_dynamic_distance
        self.max_candidates = max_candidates
        self.ignore_chain = ignore_chain
        self.best_match = best_match
        self._hits = iter(hits)

    def __iter__(self):
        return self

    def __next__(self) -> Hit:
        return next(self._hits)


def register_vocabulary(templates) -> None:
    """Give every atom/residue name the templates refer to its id now, so that batches encoded after
    this call stay valid for queries with these templates."""
    todo = [t for t in templates if getattr(t, "_vocab_version", None) is None]
    if not todo:
        return
    keys = set()
    for t in todo:  # (the distinct (match mode, names, residues) of their atoms: a thousand for the whole library)
        rows = _rows_of(t)
        if rows is not None:
            keys.update(rows[5])
        else:
            keys.update((a.match_mode, a._raw_names, a._raw_residues) for a in t._atoms)
    for key in keys:
        VOCAB.register_template_atom(*key)
    for t in todo:
        try:
            t._vocab_version = VOCAB.version
        except AttributeError:
            pass


_jess_cache: "dict[tuple, tuple]" = {}  # (id(template list), device) -> (tuple of its templates, _PackedLibrary or None, the list, a copy of the list)
_jess_cache_lock = threading.Lock()
_thread_streams = threading.local()
_OWN_STREAM = -1  # query_batch(stream=...): "the calling thread's own stream" (resolved only when something is launched)


def _thread_stream(device: int) -> int:
    """This thread's own HIP stream on ``device`` (created on first use, destroyed with the thread)."""
    streams = getattr(_thread_streams, "by_device", None)
    if streams is None:
        streams = _thread_streams.by_device = {}
    st = streams.get(device)
    if st is None:
        st = streams[device] = _native.Stream(device)
        _all_thread_streams.append(st)
    return st.handle


_all_thread_streams: list = []


def _shutdown():
    """Release what this module keeps alive between queries (device copies of single molecules, per-thread streams,
    device libraries) while the HIP runtime is still there: left to interpreter finalisation, their destructors run in
    arbitrary order, some after the runtime has gone."""
    with _single_batches._lock:
        for e in _single_batches._entries.values():
            try:
                if e[2]:
                    e[4] = True  # a query is still using it: closed when it is handed back
                else:
                    e[1].close()
            except Exception:
                pass
        _single_batches._entries.clear()
        _single_batches._atoms = 0
    for st in _all_thread_streams:
        try:
            st.close()
        except Exception:
            pass
    _all_thread_streams.clear()
    with _pack_lock:
        for _, packed in _pack_cache.values():
            try:
                packed.native.close()
            except Exception:
                pass
        _pack_cache.clear()
    _jess_cache.clear()
    _sweeps.clear()


import atexit  # noqa: E402

atexit.register(_shutdown)


class _QueryCombiner:
    """Single-molecule queries with identical templates and thresholds, issued from several threads at once, merged
    into batches.  No timer: the first call launches at once; calls arriving while it runs queue up, and when it is done
    one of the waiting threads takes everything queued as ONE query_batch call and hands the others their hits."""

    class _Request:
        __slots__ = ("molecule", "event", "hits", "error", "promoted", "rows", "leading")

        def __init__(self, molecule):
            self.molecule, self.event = molecule, threading.Event()
            self.hits, self.error, self.promoted, self.rows, self.leading = None, None, False, None, False

    MAX_BATCH = 64
    MAX_LEADERS = int(os.environ.get("ENZYMM_AMD_LEADERS", "1"))  # launches of one key in flight at a time
    PEER_WINDOW_S = 0.05   # a thread that called this recently is expected to call again (a pool mapping over molecules)
    FILL_WAIT_S = 1e-6 * float(os.environ.get("ENZYMM_AMD_FILL_WAIT_US", "400"))  # how long a leader waits for those threads to queue up

    def __init__(self, templates):
        self._templates = templates  # kept alive: the combiner is found by id(templates), which must not be reused
        self._lock = threading.Lock()
        self._filled = threading.Condition(self._lock)
        self._queue: list = []
        self._leaders = 0
        self._peers: dict = {}  # thread ident -> when it last called

    def _pass_on(self):
        """(under the lock) A leader is done: whoever waits first in the queue takes its place, else the place is free."""
        for nxt in self._queue:
            if not nxt.promoted and not nxt.leading:  # (a leader's own request waits in the queue, too)
                nxt.promoted = True
                nxt.event.set()
                return
        self._leaders -= 1

    def run(self, jess, molecule, thresholds, options, execute=None):
        """``execute(batch)`` serves a list of requests (sets ``hits`` on each); the default searches their molecules as
        one ``jess.query_batch`` call."""
        req = self._Request(molecule)
        with self._lock:
            self._peers[threading.get_ident()] = time.monotonic()
            self._queue.append(req)
            lead = req.leading = self._leaders < self.MAX_LEADERS
            if lead:
                self._leaders += 1
            else:
                self._filled.notify_all()
        while True:
            if not lead:
                req.event.wait()
                with self._lock:
                    served = req.hits is not None or req.error is not None
                    req.event.clear()
                    if req.promoted:  # a thread that finished a batch handed its place as a leader to this one
                        req.promoted = False
                        if served:    # ... which the other leader has served meanwhile: the place goes on
                            self._pass_on()
                            break
                        lead = req.leading = True
                    elif served:
                        break
                    else:
                        continue
            with self._lock:
                # The threads of a pool come back for their next molecule within a few hundred microseconds of each other:
                # a leader that launched at once would search ONE molecule while the others queue up behind it.  It waits --
                # bounded -- until its share of the threads that have been calling lately is queued (two leaders: while one
                # batch is on the GPU the other is being put together and uploaded).
                now = time.monotonic()
                if len(self._peers) > 1:
                    for ident in [i for i, t in self._peers.items() if now - t > self.PEER_WINDOW_S]:
                        del self._peers[ident]
                    target = min(self.MAX_BATCH, -(-len(self._peers) // self.MAX_LEADERS))
                    deadline = now + self.FILL_WAIT_S
                    while len(self._queue) < target:
                        left = deadline - time.monotonic()
                        if left <= 0 or not self._filled.wait(left):
                            break
                batch, self._queue = self._queue[:self.MAX_BATCH], self._queue[self.MAX_BATCH:]
            if batch:
                try:
                    if execute is not None:
                        execute(batch)
                    else:
                        per_molecule = jess.query_batch([r.molecule for r in batch], *thresholds, stream=_OWN_STREAM, **options)
                        for r, hits in zip(batch, per_molecule):
                            r.hits = hits
                except BaseException as exc:  # every caller of the batch sees the failure
                    for r in batch:
                        r.error = exc
                for r in batch:
                    if r is not req:
                        r.event.set()
            with self._lock:
                req.leading = False
                self._pass_on()
                lead = False
                if req.hits is not None or req.error is not None:
                    break
            # (the request went into the other leader's batch: wait for that one)
        if req.error is not None:
            raise req.error
        return req.hits


_combiners: "collections.OrderedDict[tuple, _QueryCombiner]" = collections.OrderedDict()
_combiners_lock = threading.Lock()


def _combiner(key, templates) -> _QueryCombiner:
    with _combiners_lock:
        c = _combiners.get(key)
        if c is None:
            c = _combiners[key] = _QueryCombiner(templates)
            while len(_combiners) > 64:
                old_key, old = next(iter(_combiners.items()))
                if old._leaders or old._queue:
                    break
                _combiners.pop(old_key)
        else:
            _combiners.move_to_end(key)
        return c


def _search_raw(jess, molecules, thresholds, options, trace=None):
    """The raw hit table of ``molecules`` against ``jess``'s templates: the one place where the sweep planner meets the
    engine (tests put a recorder here)."""
    t0 = time.perf_counter()
    batch = make_batch(molecules, jess._device) if len(molecules) > 1 else None  # (one molecule: its kept device copy)
    t1 = time.perf_counter()
    try:
        pending = jess.query_batch(molecules, *thresholds, return_raw=True, stream=_OWN_STREAM, batch=batch, _async=True, **options)
        t2 = time.perf_counter()
        table = pending.result()
    finally:
        if batch is not None:
            batch.close()
    if trace is not None:
        trace[:] = [(t1 - t0) * 1e3, (t2 - t1) * 1e3, (time.perf_counter() - t2) * 1e3]
    return table


class _Sweeps:
    """enzymm's ``Matcher.run`` asks ``Jess(templates of size k).query(molecule)`` for every molecule, one size bucket after
    the other, from a thread pool (/root/reference/enzymm/jess_run.py:896-988 around :785-843): 6 x N small queries, each
    a launch chain of its own however the concurrent ones are merged.  A query is a pure function of (templates,
    thresholds, options, molecule) and all four are immutable, so the answer may be computed BEFORE it is asked for.
    This planner sits between ``Jess.query`` and the launch and recognises the sweep:

    * molecules ahead: once a (templates, thresholds, options) combination -- a *key* -- has been asked for two
      molecules, a launch for that key also searches every recently queried molecule the key has not been asked for
      yet (the molecules of the previous bucket, in the reference's loop);
    * buckets ahead: a launch also searches, for its molecules, the other keys seen recently with the same options
      that those molecules have not been asked with -- through ONE device library put together from the keys' packed
      tables (``_PackedLibrary.concat``), so a molecule's candidate index and pair tables are built once for all
      buckets instead of once per bucket.

    Results that were not asked for wait in a table keyed by (key, molecule) and are handed out -- as the same ``Hit``
    lists a direct query builds -- when the call arrives; they expire after ``WINDOW_S``.  What is computed ahead is
    bounded per launch (``MAX_MOLECULES``, ``MAX_ATOMS``); a lone query stays a lone query.  ``ENZYMM_AMD_SWEEP=0``
    switches the planner off (every call is then searched when it is made, merged only with concurrent calls)."""

    WINDOW_S = 120.0
    MAX_MOLECULES = 1024
    MAX_ATOMS = 8_000_000
    MAX_RESULTS = 1 << 17
    MAX_UNIONS = 4

    class _Key:
        __slots__ = ("key", "jess", "templates", "thresholds", "options", "device", "last_seen", "asked", "planned", "sel", "requests", "_pack")

        def pack(self):
            """The packed tables of the key's templates (looked up once: the key of that cache is a tuple of 6780 ids)."""
            p = self._pack
            if p is None or p.vocab_version != VOCAB.version:
                p = self._pack = _packed(self.templates, self.device)
            return p

        def hits(self, rows, molecule):
            if len(rows) == 0:
                return []
            return self.jess._hits_from_table(rows, self.sel, [molecule])[0]

    def __init__(self):
        self.enabled = os.environ.get("ENZYMM_AMD_SWEEP", "1") != "0"
        self._lock = threading.Lock()
        self._keys: "collections.OrderedDict[tuple, _Sweeps._Key]" = collections.OrderedDict()
        self._molecules: "collections.OrderedDict[int, list]" = collections.OrderedDict()  # id -> [weakref, device, last seen]
        self._results: "collections.OrderedDict[tuple, tuple]" = collections.OrderedDict()  # (key, id) -> (weakref, rows, when)
        self._unions: "collections.OrderedDict[tuple, tuple]" = collections.OrderedDict()
        self._inflight: dict = {}  # (key, id(molecule)) -> Event set when the launch computing it ahead has stored its results
        # molecules that were constructed lately (Molecule.load, .conserved(), ...), oldest first: enzymm loads its molecules
        # and then asks for them in that order (jess_run.py:509-560, 896-988), so these are what a sweep will ask for next
        self._loaded: "collections.OrderedDict[int, tuple]" = collections.OrderedDict()  # id -> (weak reference, serial number)
        self._made = 0
        self.launches = []  # (molecules, keys) of the last launches: what tests and traces look at

    # -- bookkeeping (under the lock) -----------------------------------------------------------------------------
    def _touch_key(self, jess, key, thresholds, options, now):
        info = self._keys.get(key)
        if info is None:
            info = self._keys[key] = self._Key()
            info.key, info.jess, info.templates = key, jess, jess._templates
            info.thresholds, info.options, info.device = thresholds, options, jess._device
            info.asked, info.planned, info.requests, info._pack = {}, {}, 0, None
            info.sel = np.arange(len(jess._templates), dtype=np.uint32)
            while len(self._keys) > 64:
                # what was computed ahead for an evicted key goes with it: the key holds id(template tuple), and once the
                # tuple is gone a new one may get its address
                old_key, _ = self._keys.popitem(last=False)
                for pair in [p for p in self._results if p[0] == old_key]:
                    del self._results[pair]
                for ukey in [u for u in self._unions if old_key in u]:
                    del self._unions[ukey]
        info.last_seen = now
        return info

    def _touch_molecule(self, molecule, device, now):
        import weakref
        mid = id(molecule)
        ent = self._molecules.get(mid)
        if ent is not None:
            if ent[0]() is molecule:
                ent[2] = now
                return
            for k in self._keys.values():  # another object at that address now: what was noted for the old one is void
                k.asked.pop(mid, None)
                k.planned.pop(mid, None)
                self._results.pop((k.key, mid), None)
        try:
            self._molecules[mid] = [weakref.ref(molecule), device, now]
        except TypeError:
            return
        while len(self._molecules) > 4 * self.MAX_MOLECULES:
            self._molecules.popitem(last=False)

    def _expire(self, now):
        while self._results:
            k, ent = next(iter(self._results.items()))
            if now - ent[2] <= self.WINDOW_S and len(self._results) <= self.MAX_RESULTS:
                break
            self._results.popitem(last=False)

    def _take(self, key, molecule):
        ent = self._results.pop((key, id(molecule)), None)
        if ent is None or ent[0]() is not molecule:
            return None
        return ent[1]

    MAX_LOADED = 8192

    def loaded(self, molecule):
        """A molecule was constructed (a weak reference is all that is kept)."""
        import weakref
        try:
            ref = weakref.ref(molecule)
        except TypeError:
            return
        with self._lock:
            mid = id(molecule)
            known = self._loaded.pop(mid, None) is not None  # (a new object at an address seen before goes to the young end)
            self._made += 1
            self._loaded[mid] = (ref, self._made)
            while len(self._loaded) > self.MAX_LOADED:
                self._loaded.popitem(last=False)
            old = self._molecules.get(mid)
            if old is not None and old[0]() is not molecule:
                del self._molecules[mid]
                known = True
            if known:  # what was noted for the old object at that address is void
                for k in self._keys.values():
                    if mid in k.asked or mid in k.planned:
                        k.asked.pop(mid, None)
                        k.planned.pop(mid, None)
                        self._results.pop((k.key, mid), None)

    # -- the two entry points ---------------------------------------------------------------------------------------
    def lookup(self, jess, key, thresholds, options, molecule):
        """The hits of a query whose answer was computed ahead, or None."""
        now = time.monotonic()
        with self._lock:
            rows = self._take(key, molecule)
            if rows is None:
                return None
            info = self._touch_key(jess, key, thresholds, options, now)
            info.asked[id(molecule)] = now
            info.requests += 1
            self._touch_molecule(molecule, info.device, now)
        return info.hits(rows, molecule)

    def execute(self, jess, key, thresholds, options, requests):
        """Serves the requests (objects with ``molecule`` and ``hits``) of one key: from the table of results computed
        ahead where it can -- waiting for those a launch of another thread is computing right now --, by ONE launch
        otherwise, which also computes ahead what the sweep will ask next."""
        pending, first = list(requests), True
        while pending:
            now = time.monotonic()
            todo, waiting = [], []
            with self._lock:
                info = self._touch_key(jess, key, thresholds, options, now)
                if first:
                    self._expire(now)
                for r in pending:
                    mid = id(r.molecule)
                    if first:
                        self._touch_molecule(r.molecule, info.device, now)
                    rows = self._take(key, r.molecule)
                    if rows is not None:
                        r.hits = info.hits(rows, r.molecule)
                    elif (key, mid) in self._inflight:
                        waiting.append((r, self._inflight[(key, mid)]))
                    else:
                        todo.append(r)
                    info.asked[mid] = now
                if first:
                    info.requests += len(pending)
                    if len(info.asked) + len(info.planned) > 1 << 16:
                        info.asked = {m: t for m, t in info.asked.items() if now - t <= self.WINDOW_S}
                        info.planned = {m: t for m, t in info.planned.items() if now - t <= self.WINDOW_S}
                extra, parts, flight = [], [info], None
                if todo and self.enabled:
                    extra, parts, flight = self._plan(info, todo, now)
            if todo:
                molecules = [r.molecule for r in todo] + extra
                try:
                    try:
                        self._launch(jess, info, thresholds, todo, molecules, parts)
                    except (_native.NativeError, ValueError):
                        if len(parts) == 1:
                            raise
                        self._launch(jess, info, thresholds, todo, molecules, [info])  # (a library over all the keys the engine refuses)
                finally:
                    if flight is not None:
                        self._landed(flight)
            for _, event in waiting:
                event.wait(120.0)
            pending, first = [r for r, _ in waiting], False

    def _landed(self, flight):
        """The launch that was computing the pairs of ``flight`` ahead has stored them (or failed): whoever waits may look."""
        with self._lock:
            for pair in flight[1]:
                self._inflight.pop(pair, None)
        flight[0].set()

    def _launch(self, jess, info, thresholds, todo, molecules, parts):
        """One search of ``molecules`` with the keys ``parts`` (one library over all of them when there are several); the
        first ``len(todo)`` molecules are the ones asked for with ``info``'s key, everything else is kept for later."""
        t_launch = time.perf_counter()
        trace = [0.0, 0.0, 0.0]  # ms: device copy of the molecules, launch, wait + hit table
        table = offsets = None
        options = parts[0].options
        if len(parts) > 1:
            ujess, offsets, thr = self._union(parts)
            table = _search_raw(ujess, molecules, thr, options, trace)
        else:
            table = _search_raw(parts[0].jess if jess is None else jess, molecules, parts[0].thresholds if thresholds is None else thresholds,
                                options, trace)
        self.launches.append((len(molecules), len(todo), tuple(k.key for k in parts), time.perf_counter() - t_launch, tuple(trace)))
        del self.launches[:-64]
        # rows of (molecule, key), as tables of that key's own query of that one molecule
        nk, nm = len(parts), len(molecules)
        tsel = table["template_sel"].astype(np.int64)
        kidx = np.searchsorted(offsets[1:], tsel, side="right") if offsets is not None else np.zeros(len(table), dtype=np.int64)
        group = table["structure"].astype(np.int64) * nk + kidx
        if len(group) > 1 and np.any(group[1:] < group[:-1]):  # (the engine sorts by structure, template)
            order = np.argsort(group, kind="stable")
            table, group, tsel, kidx = table[order], group[order], tsel[order], kidx[order]
        table["template_sel"] = tsel - (offsets[kidx] if offsets is not None else 0)
        table["structure"] = 0
        bounds = np.searchsorted(group, np.arange(nm * nk + 1)).tolist()
        own = parts.index(info) if info is not None else -1
        import weakref
        now = time.monotonic()
        with self._lock:
            for mi, m in enumerate(molecules):
                mid = id(m)
                for ki, kinfo in enumerate(parts):
                    rows = table[bounds[mi * nk + ki]:bounds[mi * nk + ki + 1]]
                    if ki == own and mi < len(todo):
                        todo[mi].rows = rows
                    elif (kinfo.key, mid) not in self._results:  # (kept until asked for, or until it expires)
                        self._results[(kinfo.key, mid)] = (weakref.ref(m), rows, now)
        for r in todo:
            r.hits = info.hits(r.rows, r.molecule)

    def drain(self, timeout=60.0):
        """Waits until nothing computed ahead is in flight any more (tests)."""
        deadline = time.monotonic() + timeout
        while time.monotonic() < deadline:
            with self._lock:
                if not self._inflight:
                    return True
            time.sleep(0.0005)
        return False

    def _plan(self, info, todo, now):
        """(molecules to search ahead with info's key, keys to search in the launch: info and those worth searching ahead)"""
        extra = []
        if info.requests >= 2:
            have = {id(r.molecule) for r in todo}
            n_atoms = sum(len(r.molecule) for r in todo)
            # candidates: (1) the molecules constructed AFTER the ones asked for now, in the order they were made -- enzymm
            # loads its molecules and then asks for them in that order, so those are what the sweep asks next; molecules made
            # before them are not touched (already handled, or none of this sweep's business) --, (2) the molecules other keys
            # were asked for lately (the buckets after the first of a first sweep)
            seen = set()
            first_made = min((self._loaded[mid][1] for mid in have if mid in self._loaded), default=None)
            younger = []
            if first_made is not None:
                for mid in reversed(self._loaded):
                    ref, serial = self._loaded[mid]
                    if serial <= first_made:
                        break
                    younger.append((mid, ref))
                younger.reverse()
            recent = [(mid, ent[0]) for mid, ent in self._molecules.items() if ent[1] == info.device and now - ent[2] <= self.WINDOW_S] \
                if (self._molecules.keys() - info.asked.keys() - info.planned.keys() - have) else []
            for mid, ref in younger + recent:
                if len(todo) + len(extra) >= self.MAX_MOLECULES or n_atoms >= self.MAX_ATOMS:
                    break
                if mid in have or mid in seen or mid in info.asked or mid in info.planned or (info.key, mid) in self._results:
                    continue
                m = ref()
                if m is None or not len(m):
                    continue
                self._touch_molecule(m, info.device, now)  # (voids whatever an earlier object at that address left behind)
                extra.append(m)
                seen.add(mid)
                info.planned[mid] = now  # (a launch of another leader must not compute it ahead as well)
                n_atoms += len(m)
        wanted = [id(r.molecule) for r in todo] + [id(m) for m in extra]
        parts = []
        for k in self._keys.values():
            if k is info:
                parts.append(k)
            elif (k.options == info.options and k.device == info.device and k.requests >= 2 and now - k.last_seen <= self.WINDOW_S
                  and any(mid not in k.asked and mid not in k.planned and (k.key, mid) not in self._results for mid in wanted)):
                parts.append(k)
        if len(parts) > 1:  # (orientation tables: all or none, see _PackedLibrary.concat)
            try:
                packs = [k.pack() for k in parts]
                if len({p.tables[8] is not None for p in packs}) != 1:
                    parts = [info]
            except Exception:
                parts = [info]
        # what this launch computes ahead is IN FLIGHT until its results are stored: a request for it waits instead of
        # launching the same search again
        flight = (threading.Event(), [(info.key, id(m)) for m in extra])
        for k in parts:
            if k is not info:
                for mid in wanted:
                    if mid not in k.asked and mid not in k.planned and (k.key, mid) not in self._results:
                        k.planned[mid] = now
                        flight[1].append((k.key, mid))
        for pair in flight[1]:
            self._inflight[pair] = flight[0]
        return extra, parts, flight

    def _union(self, parts):
        ukey = tuple(k.key for k in parts)
        with self._lock:
            hit = self._unions.get(ukey)
            if hit is not None and hit[0]._packed_ref.vocab_version == VOCAB.version:
                self._unions.move_to_end(ukey)
                return hit
        device = parts[0].device
        packs = [k.pack() for k in parts]
        union = _PackedLibrary.concat(packs, device)
        ujess = Jess.__new__(Jess)
        ujess._device, ujess._cache_key = device, None
        ujess._templates = tuple(t for k in parts for t in k.templates)
        ujess._packed_ref = union
        offsets = np.concatenate([[0], np.cumsum([len(k.templates) for k in parts])]).astype(np.int64)
        thr = tuple(np.concatenate([np.full(len(k.templates), k.thresholds[i], dtype=np.float64) for k in parts]) for i in range(3))
        with self._lock:
            self._unions[ukey] = (ujess, offsets, thr)
            while len(self._unions) > self.MAX_UNIONS:
                self._unions.popitem(last=False)  # (its device library goes when the last query using it is done)
        return ujess, offsets, thr

    def clear(self):
        self.drain(5.0)
        with self._lock:
            self._keys.clear(); self._molecules.clear(); self._results.clear(); self._unions.clear(); self._inflight.clear()
            self._loaded.clear()
            del self.launches[:]


_sweeps = _Sweeps()
_molecule_created = _sweeps.loaded


class Jess:
    query_batch_accepts_async = True  # Matcher pipelines batches through query_batch(_async=True)

    """A set of templates to search structures with (device-resident once constructed)."""

    def __init__(self, templates: Iterable[Template] = (), device: Optional[int] = None):
        self._device = default_device() if device is None else int(device)
        if hasattr(templates, "device_tables"):
            # library_pack.PackedTemplates: a lazy sequence; Template objects appear only for hits
            self._templates = templates
            _packed(templates, self._device)  # registers the library's vocabulary
            return
        # enzymm builds Jess(templates) anew for every (molecule, size) call with the SAME list object
        # (jess_run.py:800, templates_by_effective_size[size]): a list seen before, holding the same objects, needs
        # neither a new tuple of its members nor the type checks nor a new device library.  The entry keeps the list
        # object itself (while it is cached no other list can take its id, and `is` says it is that very list) and a copy
        # of it: comparing the two lists -- at C speed, element comparison stops at identity -- notices a list that was
        # changed in place.
        self._cache_key = (id(templates), self._device) if isinstance(templates, (list, tuple)) else None
        known = _jess_cache.get(self._cache_key) if self._cache_key is not None else None
        if known is not None and known[2] is templates and known[3] == templates:
            self._templates = known[0]
            if known[1] is not None and known[1].vocab_version == VOCAB.version:
                self._packed_ref = known[1]
            return
        self._templates = tuple(templates)
        for t in self._templates:
            if not isinstance(t, Template):
                raise TypeError(f"expected Template, found {type(t).__name__}")
        register_vocabulary(self._templates)
        if self._cache_key is not None and self._templates:
            with _jess_cache_lock:  # (Jess objects are built from pool threads)
                known = _jess_cache.get(self._cache_key)
                if known is not None and known[2] is templates and known[3] == templates:
                    self._templates = known[0]  # another thread was faster: one tuple per list (queries are keyed by it)
                else:
                    _jess_cache[self._cache_key] = (self._templates, None, templates, type(templates)(templates))
                    while len(_jess_cache) > _PACK_CACHE_SIZE:
                        _jess_cache.pop(next(iter(_jess_cache)), None)

    def __len__(self):
        return len(self._templates)

    def __iter__(self):
        return iter(self._templates)

    def __getitem__(self, i):
        return self._templates[i]

    # -- batched entry point (what enzymm_amd.jess_run.Matcher uses) ---------
    def query_batch(self, molecules: Sequence[Molecule], rmsd_threshold, distance_cutoff, max_dynamic_distance, *,
                    max_candidates: Optional[int] = None, ignore_chain: bool = False, best_match: bool = True,
                    template_index=None, batch=None, count_candidates: bool = False, return_raw: bool = False,
                    models=None, _async: bool = False, stream: int = 0):
        """Search every molecule with every (selected) template in one launch.

        The three thresholds are scalars or arrays with one entry per selected template, so all of
        enzymm's size buckets (jess_run.py:564-571) go through a single call.  Returns
        ``[[Hit, ...] per molecule]`` in template order, or the raw structured hit table.
        """
        molecules = list(molecules)
        for m in molecules:
            if not isinstance(m, Molecule):
                raise TypeError(f"expected Molecule, found {type(m).__name__}")
        sel = np.arange(len(self._templates), dtype=np.uint32) if template_index is None \
            else np.ascontiguousarray(template_index, dtype=np.uint32)
        if not len(self._templates) or not len(sel) or not molecules:
            empty = np.zeros(0, dtype=_native.HIT_DTYPE) if return_raw else [[] for _ in molecules]
            if return_raw and count_candidates:
                empty = (empty, np.zeros((len(molecules), len(sel)), dtype=np.uint32))
            return _Ready(empty) if _async else empty
        # the device copy is looked up by the identity of the templates; remembered on this object, so that a
        # long-lived Jess does not rebuild the 6780-entry key for every batch
        packed = getattr(self, "_packed_ref", None)
        if packed is None or packed.vocab_version != VOCAB.version:
            packed = self._packed_ref = _packed(self._templates, self._device)
            key = getattr(self, "_cache_key", None)
            with _jess_cache_lock:
                known = _jess_cache.get(key) if key is not None else None
                if known is not None and known[0] is self._templates:
                    _jess_cache[key] = (self._templates, packed, known[2], known[3])
        own_batch = batch is None
        release = None
        if own_batch and len(molecules) == 1:
            # the reference queries one molecule six times, once per size bucket: its device copy is kept
            batch, release = _single_batch(molecules[0], self._device)
            own_batch = release is None
        elif own_batch:
            batch = make_batch(molecules, self._device)
        else:
            batch = getattr(batch, "native", batch)  # an ingest.PdbBatch or a raw _native.Batch
            if getattr(batch, "vocab_version", VOCAB.version) != VOCAB.version:
                raise ValueError("this batch was encoded before new atom/residue names were registered by a "
                                 "template; create the Jess (or its templates) first, then the batch")
            if batch.n_structures != len(molecules):
                raise ValueError("batch and molecule list differ in length")
        if stream == _OWN_STREAM:
            stream = _thread_stream(self._device)
        flags = _native.TM_FLAG_BEST_MATCH if best_match else 0
        if ignore_chain:
            flags |= _native.TM_FLAG_IGNORE_CHAIN
        if count_candidates:
            flags |= _native.TM_FLAG_COUNT_CANDIDATES
        cap = _UNLIMITED if max_candidates is None else max(0, min(int(max_candidates), _UNLIMITED))
        args = (packed.native, batch, sel, rmsd_threshold, distance_cutoff, max_dynamic_distance, cap, flags)
        try:
            pending = _PendingQuery(self, molecules, sel, args, models, batch if own_batch else None, count_candidates,
                                    return_raw, stream, release)
        except BaseException:
            if release is not None:
                release()
            raise
        return pending if _async else pending.result()

    def _hits_from_table(self, table: np.ndarray, sel: np.ndarray, molecules: Sequence[Molecule]) -> List[List[Hit]]:
        """``[[Hit, ...] per molecule]`` from a hit table (sorted by structure, template).  Columns are pulled out
        of the table once; a Hit costs about a microsecond (tens of thousands of structures per second come with
        thousands of hits per batch)."""
        out: List[List[Hit]] = [[] for _ in molecules]
        n = len(table)
        if n == 0:
            return out
        structure = table["structure"].tolist()
        tpl_index = sel[table["template_sel"]].tolist()
        n_atoms = table["n_atoms"].tolist()
        rmsd = table["rmsd"].tolist()
        flags = table["flags"].tolist()
        orientation = table["orientation"].tolist()
        n_candidates = table["n_candidates"].tolist()
        atoms = table["atoms"].tolist()
        rot = np.ascontiguousarray(table["rot"]).reshape(n, 3, 3)
        trans = np.ascontiguousarray(table["trans"])
        templates = self._templates
        TRUNC, PRED, CORRECT = _native.TM_HIT_TRUNCATED, _native.TM_HIT_PREDICTED, _native.TM_HIT_CORRECT
        new = Hit.__new__
        for i in range(n):
            h = new(Hit)
            s = structure[i]
            f = flags[i]
            h.template = templates[tpl_index[i]]
            h._molecule = molecules[s]
            h.rmsd = rmsd[i]
            h._rot = rot[i]
            h._trans = trans[i]
            h._atom_index = tuple(atoms[i][:n_atoms[i]])
            h.n_candidates = n_candidates[i]
            h.truncated = bool(f & TRUNC)
            h.orientation = orientation[i]
            h.predicted = bool(f & CORRECT) if f & PRED else None
            out[s].append(h)
        return out

    # -- the pyjess call enzymm makes ----------------------------------------
    def query(self, molecule: Molecule, rmsd_threshold: float, distance_cutoff: float,
              max_dynamic_distance: float, *, max_candidates: Optional[int] = None, ignore_chain: bool = False,
              best_match: bool = False) -> Query:
        # enzymm maps this call over molecules from a thread pool (jess_run.py:919-921, 977-981).  Calls that arrive
        # while another call with the same templates and thresholds is on the GPU are searched TOGETHER, as one batch, by
        # whichever of the waiting threads goes next (_QueryCombiner): sixteen threads then cost a launch chain per
        # sixteen molecules, not per molecule.  A lone call goes straight through (on the calling thread's own stream).
        # The reference repeats the sweep for every size bucket: _Sweeps computes ahead what it can (the molecules of the
        # previous bucket, the buckets still to come) and hands those answers out here without a launch.
        key = (id(self._templates), self._device, float(rmsd_threshold), float(distance_cutoff), float(max_dynamic_distance),
               max_candidates, bool(ignore_chain), bool(best_match))
        thresholds = (rmsd_threshold, distance_cutoff, max_dynamic_distance)
        options = dict(max_candidates=max_candidates, ignore_chain=ignore_chain, best_match=best_match)
        hits = _sweeps.lookup(self, key, thresholds, options, molecule) if _sweeps.enabled else None
        if hits is None:
            # (the molecule's column record, built once per molecule, by the thread that asks -- not by the leader of the
            # launch, whose work is on the critical path of everyone waiting for that launch)
            molecule._pointer_row()
            hits = _combiner(key, self._templates).run(
                self, molecule, thresholds, options,
                execute=lambda batch: _sweeps.execute(self, key, thresholds, options, batch))
        return Query(self, molecule, rmsd_threshold, distance_cutoff, max_dynamic_distance, max_candidates,
                     ignore_chain, best_match, hits)
