"""Match-mode semantics -> integer vocabularies and bit masks.

The HIP library never sees residue names, atom names or Jess match modes.
This module is the single place where they are interpreted:

* every target atom gets a residue-type id (0..30, 31 = other) and an
  atom-name id (0..62, 63 = other);
* every template atom gets ``res_mask`` (bit r set: residue-type id r is
  admissible), ``name_mask`` (bit a set: atom-name id a is admissible) and an
  ``any_residue`` flag.

Rules (SURVEY.md 8c-2 "hypothesis E"; the behaviour of ``pyjess``'s
``TemplateAtom``/``Jess.query`` that enzymm relies on at
/root/reference/enzymm/template.py:679 and jess_run.py:803-811).  Names are
compared on the raw 4-character PDB field, so a calcium ion ``CA  `` is not a
C-alpha `` CA ``:

====================  =====================================================
``match_mode % 100``  admissible target atom names
====================  =====================================================
0                     the template atom's name
1, 3                  any name with the same element columns (first two
                      characters of the field).  Mode 1 is NOT pinned by any
                      reference test and is treated like 3.
8                     the name, or its histidine ring-flip partner
                      (ND1<->CD2, NE2<->CE1)
====================  =====================================================

``match_mode >= 100`` disables the residue-name test; otherwise the target
residue name must be the template atom's 3-letter name or the expansion of one
of its one-letter alternatives.  Other modes raise ``ValueError``.
"""
from __future__ import annotations

import threading

import numpy as np

RES_OTHER = 31
NAME_OTHER = 63

STANDARD_RESIDUES = (
    "ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE",
    "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL",
)

ONE_LETTER = dict(zip("ARNDCQEGHILKMFPSTWYV", STANDARD_RESIDUES))
# letters beyond the 20 that occur in the shipped library (meaning not pinned by any reference test)
ONE_LETTER.update({"X": "XXX", "U": "SEC", "O": "PYL"})

_STANDARD_ATOMS = (
    " N  ", " CA ", " C  ", " O  ", " CB ", " CG ", " CG1", " CG2", " CD ", " CD1", " CD2", " CE ", " CE1",
    " CE2", " CE3", " CZ ", " CZ2", " CZ3", " CH2", " ND1", " ND2", " NE ", " NE1", " NE2", " NH1", " NH2",
    " NZ ", " OD1", " OD2", " OE1", " OE2", " OG ", " OG1", " OH ", " SD ", " SG ", " OXT",
)

_HIS_FLIP = {b" ND1": b" CD2", b" CD2": b" ND1", b" NE2": b" CE1", b" CE1": b" NE2"}


def raw_atom_name(name: str) -> bytes:
    """4-character PDB field of a stripped name ('CA' -> b' CA '); 4-character names are kept."""
    name = str(name)
    if len(name) >= 4:
        return name[:4].encode("ascii")
    return (" " + name.ljust(3)).encode("ascii")


def raw_residue_name(name: str) -> bytes:
    name = str(name)
    return name.rjust(3)[:3].encode("ascii") if len(name) < 3 else name[:3].encode("ascii")


class Vocabulary:
    """Append-only id registries shared by every template and molecule in the process."""

    def __init__(self):
        self._lock = threading.RLock()
        self.version = 0
        self.res_ids: dict[bytes, int] = {raw_residue_name(r): i for i, r in enumerate(STANDARD_RESIDUES)}
        self.name_ids: dict[bytes, int] = {}      # raw 4-char name -> id
        self.element_ids: dict[bytes, int] = {}   # raw 2-char element columns -> id of "other name of that element"
        for nm in _STANDARD_ATOMS:
            self._add_name(nm.encode())

    # -- registration -----------------------------------------------------
    def _next_name_id(self) -> int:
        n = len(self.name_ids) + len(self.element_ids)
        if n >= NAME_OTHER:
            raise ValueError("more than 63 distinct atom names/elements across the loaded templates")
        return n

    def _add_name(self, raw: bytes) -> int:
        with self._lock:
            i = self.name_ids.get(raw)
            if i is None:
                i = self.name_ids[raw] = self._next_name_id()
                self.version += 1
            return i

    def _add_element(self, el: bytes) -> int:
        with self._lock:
            i = self.element_ids.get(el)
            if i is None:
                i = self.element_ids[el] = self._next_name_id()
                self.version += 1
            return i

    def _add_residue(self, raw: bytes) -> int:
        with self._lock:
            i = self.res_ids.get(raw)
            if i is None:
                if len(self.res_ids) >= RES_OTHER:
                    raise ValueError("more than 31 distinct residue names across the loaded templates")
                i = self.res_ids[raw] = len(self.res_ids)
                self.version += 1
            return i

    def register_template_atom(self, match_mode: int, raw_names, raw_residues):
        """Make sure every name this template atom can refer to has an id."""
        rule = match_mode % 100
        if rule not in (0, 1, 3, 8):
            raise ValueError(f"unsupported Jess match mode {match_mode}")
        with self._lock:
            for raw in raw_names:
                self._add_name(raw)
                if rule in (1, 3):
                    self._add_element(raw[:2])
                elif rule == 8 and raw in _HIS_FLIP:
                    self._add_name(_HIS_FLIP[raw])
            if match_mode < 100:
                for r in raw_residues:
                    self._add_residue(r)

    # -- masks (call after every template of the library was registered) ---
    def name_mask(self, match_mode: int, raw_names) -> int:
        rule = match_mode % 100
        mask = 0
        for raw in raw_names:
            if rule == 0:
                mask |= 1 << self.name_ids[raw]
            elif rule in (1, 3):
                el = raw[:2]
                mask |= 1 << self.element_ids[el]
                for other, i in self.name_ids.items():
                    if other[:2] == el:
                        mask |= 1 << i
            elif rule == 8:
                mask |= 1 << self.name_ids[raw]
                if raw in _HIS_FLIP:
                    mask |= 1 << self.name_ids[_HIS_FLIP[raw]]
            else:
                raise ValueError(f"unsupported Jess match mode {match_mode}")
        return mask

    def res_mask(self, match_mode: int, raw_residues) -> int:
        if match_mode >= 100:
            return 0
        mask = 0
        for r in raw_residues:
            mask |= 1 << self.res_ids[r]
        return mask

    # -- target-atom encoding ----------------------------------------------
    def encode_names(self, raw4: np.ndarray) -> np.ndarray:
        """uint8 atom-name ids of an ``S4`` array."""
        out = np.full(raw4.shape, NAME_OTHER, dtype=np.uint8)
        uniq, inv = np.unique(raw4, return_inverse=True)
        lut = np.empty(len(uniq), dtype=np.uint8)
        for i, raw in enumerate(uniq):
            raw = bytes(raw).ljust(4)
            j = self.name_ids.get(raw)
            if j is None:
                j = self.element_ids.get(raw[:2], NAME_OTHER)
            lut[i] = j
        out[...] = lut[inv].reshape(raw4.shape)
        return out

    def encode_residues(self, raw3: np.ndarray) -> np.ndarray:
        uniq, inv = np.unique(raw3, return_inverse=True)
        lut = np.array([self.res_ids.get(bytes(r).ljust(3), RES_OTHER) for r in uniq], dtype=np.uint8)
        return lut[inv].reshape(raw3.shape)


    def native_tables(self):
        """(names S4, name ids, elements S2, element ids, residues S3, residue ids) for tm_vocab."""
        with self._lock:
            names = np.array(list(self.name_ids), dtype="S4")
            name_ids = np.array(list(self.name_ids.values()), dtype=np.uint8)
            els = np.array(list(self.element_ids), dtype="S2")
            el_ids = np.array(list(self.element_ids.values()), dtype=np.uint8)
            res = np.array(list(self.res_ids), dtype="S3")
            res_ids = np.array(list(self.res_ids.values()), dtype=np.uint8)
            return names, name_ids, els, el_ids, res, res_ids, self.version


VOCAB = Vocabulary()
