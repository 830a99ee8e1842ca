"""Template-side host model: the part of ``enzymm/template.py`` that sits on the matching path.

Mirrors the reference's public names and behaviour for

* ``Vec3``                       /root/reference/enzymm/template.py:43-181
* ``Residue`` (3 template atoms + orientation vector)          :184-422
* ``Cluster``                                                    :435-452
* ``Template(pyjess.Template)``: REMARK metadata, ``effective_size``,
  ``multimeric``, ``relative_order``, ``load``/``loads``         :455-1021
* ``load_templates``                                             :1505-1575

so that the bucket key (``effective_size``, :786-808), the orientation definition the
hit filter needs (:213-304), the EC/CATH enrichment of the TSV columns (:839-885) and the exception
types the reference's tests expect (tests/test_template.py:453-497, 697-707) are the same.  The
``AnnotatedTemplate`` layer (catalytic-residue homolog counts, :1024-1474) needs a data blob the
reference itself does not ship (.MISSING_LARGE_BLOBS) and never touches geometry: out of scope
(SURVEY.md section 2, rows 8 and 12).
"""
from __future__ import annotations

import functools
import glob
import gzip
import io
import json
import math
import os
import re
import warnings
from dataclasses import dataclass
from functools import cached_property
from pathlib import Path
from typing import Iterable, Iterator, List, Optional, Sequence, Tuple, Union

from . import pyjess

__all__ = ["Vec3", "Residue", "Cluster", "Template", "load_templates", "load_packed_library", "check_template"]

_Number = (int, float)


@dataclass(frozen=True)
class Vec3:
    """Immutable 3-vector; refuses NaN components (a failed superposition must not pass silently)."""

    x: float
    y: float
    z: float

    def __post_init__(self):
        if any(math.isnan(c) for c in (self.x, self.y, self.z)):
            raise ValueError("Cannot create a Vec3 with NaN values. Likely the superposition failed.")

    @classmethod
    def from_xyz(cls, item) -> "Vec3":
        return cls(item.x, item.y, item.z)

    @property
    def norm(self) -> float:
        return math.sqrt(self.x ** 2 + self.y ** 2 + self.z ** 2)

    def normalize(self) -> "Vec3":
        n = self.norm
        return Vec3(self.x, self.y, self.z) if n == 0 else self / n

    def _lift(self, other):
        if isinstance(other, Vec3):
            return other.x, other.y, other.z
        if isinstance(other, _Number):
            return other, other, other
        raise TypeError(f"Expected int, float or Vec3, got {type(other).__name__}")

    def __matmul__(self, other: "Vec3") -> float:
        if not isinstance(other, Vec3):
            raise TypeError(f"Expected Vec3, got {type(other).__name__}")
        return self.x * other.x + self.y * other.y + self.z * other.z

    def __add__(self, other) -> "Vec3":
        a, b, c = self._lift(other)
        return Vec3(self.x + a, self.y + b, self.z + c)

    def __sub__(self, other) -> "Vec3":
        a, b, c = self._lift(other)
        return Vec3(self.x - a, self.y - b, self.z - c)

    def __truediv__(self, other) -> "Vec3":
        a, b, c = self._lift(other)
        return Vec3(self.x / a, self.y / b, self.z / c)

    def angle_to(self, other: "Vec3") -> float:
        """Angle in radians between the two directions."""
        dot = self.normalize() @ other.normalize()
        if -1 <= dot <= 1:
            return math.acos(dot)
        if math.isclose(dot, 1, rel_tol=1e-5):
            return 0
        if math.isclose(dot, -1, rel_tol=1e-5):
            return math.pi
        raise ValueError(f"ArcCos is not defined outside [-1,1]: {self} vs {other}")


# Orientation of a template residue (template.py:233-256): either the vector between two named
# atoms, or from a centre atom to the midpoint of the other two ("mid").
_ORIENTATION = {
    "GLY": ("C", "O"), "ALA": ("CA", "CB"), "VAL": ("CA", "CB"), "LEU": ("CA", "CB"), "ILE": ("CA", "CB"),
    "MET": ("CG", "SD"), "PHE": ("CZ", "mid"), "TYR": ("CZ", "OH"), "TRP": ("CZ2", "NE1"), "CYS": ("CB", "SG"),
    "PRO": ("C", "O"), "SER": ("CB", "OG"), "THR": ("CB", "OG1"), "ASN": ("CG", "OD1"), "GLN": ("CD", "OE1"),
    "LYS": ("CE", "NZ"), "ARG": ("CZ", "mid"), "HIS": ("CG", "ND1"), "ASP": ("CG", "mid"), "GLU": ("CD", "mid"),
    "PTM": ("CA", "CB"), "ANY": ("C", "O"),
}
_MID = 9  # sentinel second index meaning "midpoint of the other two atoms" (jess_run.py:434-436)

_THREE_TO_ONE = {
    "ALA": "A", "CYS": "C", "ASP": "D", "GLU": "E", "PHE": "F", "GLY": "G", "HIS": "H", "ILE": "I", "LYS": "K",
    "LEU": "L", "MET": "M", "ASN": "N", "PRO": "P", "GLN": "Q", "ARG": "R", "SER": "S", "THR": "T", "VAL": "V",
    "TRP": "W", "TYR": "Y", "XXX": "X",
}


class Residue:
    """Three template atoms of one template residue plus its orientation vector."""

    __slots__ = ("_atoms", "_vec", "_indices")

    def __init__(self, atoms: Tuple[pyjess.TemplateAtom, pyjess.TemplateAtom, pyjess.TemplateAtom]):
        atoms = tuple(atoms)
        vec, indices = self.calc_residue_orientation(atoms)
        object.__setattr__(self, "_atoms", atoms)
        object.__setattr__(self, "_vec", vec)
        object.__setattr__(self, "_indices", indices)

    def __setattr__(self, key, value):
        raise AttributeError("Residue is immutable")

    def __reduce__(self):  # (immutable + slots: unpickled through the constructor, which recomputes the orientation)
        return (type(self), (self._atoms,))

    def __eq__(self, other):
        return isinstance(other, Residue) and self._atoms == other._atoms

    def __hash__(self):
        return hash(self._atoms)

    def __repr__(self):
        return f"Residue({self.residue_name} {self.chain_id}{self.residue_number})"

    @staticmethod
    def calc_residue_orientation(atoms) -> Tuple[Vec3, Tuple[int, int]]:
        kind = atoms[0].residue_names[0]
        if kind not in _ORIENTATION:
            raise KeyError(f"Residue orientation is not defined for the residue type {kind}")
        first, second = _ORIENTATION[kind]

        def locate(name, what):
            for i, a in enumerate(atoms):
                if a.atom_names[0] == name:
                    return i
            raise ValueError(f"Failed to find {what} atom for amino-acid {kind!r}")

        if second == "mid":
            mid_i = locate(first, "middle")
            sides = [Vec3.from_xyz(a) for i, a in enumerate(atoms) if i != mid_i]
            if len(sides) != 2:
                raise ValueError(f"Failed to find side atoms for amino-acid {kind!r}")
            return (sides[0] + sides[1]) / 2 - Vec3.from_xyz(atoms[mid_i]), (mid_i, _MID)
        i, j = locate(first, "first"), locate(second, "second")
        return Vec3.from_xyz(atoms[j]) - Vec3.from_xyz(atoms[i]), (i, j)

    @classmethod
    def construct_residues_from_atoms(cls, atoms: Iterable[pyjess.TemplateAtom]) -> List["Residue"]:
        atoms = list(atoms)
        out = []
        for i in range(0, len(atoms), 3):
            trip = tuple(atoms[i:i + 3])
            if len(trip) != 3:
                raise ValueError(f"Failed to construct residues. Got only {len(trip)} ATOM lines")
            keys = {(a.residue_names[0], a.chain_id, a.residue_number) for a in trip}
            if len(keys) != 1:
                raise ValueError(f"Failed to construct residues. Atoms of different chains, residue types or "
                                 f"residue numbers {keys} found in Atom triplet")
            out.append(cls(trip))
        return out

    atoms = property(lambda self: self._atoms)
    residue_name = property(lambda self: self._atoms[0].residue_names[0])
    match_mode = property(lambda self: self._atoms[0].match_mode)
    residue_number = property(lambda self: self._atoms[0].residue_number)
    chain_id = property(lambda self: self._atoms[0].chain_id)
    orientation_vector = property(lambda self: self._vec)
    orientation_vector_indices = property(lambda self: self._indices)

    @property
    def allowed_residues(self) -> str:
        return "".join(set(_THREE_TO_ONE[n] for n in self._atoms[0].residue_names))

    @property
    def backbone(self) -> bool:
        names = self._atoms[0].residue_names
        return "ANY" in names or "XXX" in names


@dataclass(frozen=True)
class Cluster:
    id: int
    member: int
    size: int

    def __post_init__(self):
        if self.member > self.size:
            raise ValueError("Cluster member cannot be greater than cluster size")


_UNIPROT = re.compile(r"[OPQ][0-9][A-Z0-9]{3}[0-9]|[A-NR-Z][0-9]([A-Z][A-Z0-9]{2}[0-9]){1,2}")
_EC = re.compile(r"[1-7](\.(\-|\d{1,})){3}")
_EC_PRELIM = re.compile(r"[1-7](\.(\-|\d{1,}|n\d{1,})){3}")
_CATH = re.compile(r"[1-46](\.(\-|\d{1,})){3}")
_EMPTY = {"NONE", "?", "NAN", "NA"}


def _remark_handlers():
    def pdb_id(tok, md, warn):
        if len(tok[2]) != 4:
            raise ValueError(f"Found {tok[2]} which does not have the 4 characters of a PDB_ID.")
        md["pdb_id"] = tok[2].lower()

    def uniprot(tok, md, warn):
        m = _UNIPROT.search(tok[2])
        if not m:
            raise ValueError(f"Did not find a valid UniProt ID, found {tok[2]}")
        md["uniprot_id"] = m.group()

    def integer(field, what):
        def handler(tok, md, warn):
            try:
                md[field] = int(tok[2])
            except ValueError as exc:
                raise ValueError(f"Did not find {what}, found {tok[2]}") from exc
        return handler

    def cluster(tok, md, warn):
        try:
            md["cluster"] = Cluster(*map(int, tok[2].split("_")))
        except (ValueError, TypeError) as exc:
            raise ValueError(f"Did not find a Cluster specification <id>_<member>_<size>, found {tok[2]}") from exc

    def resolution(tok, md, warn):
        try:
            md["resolution"] = float(tok[2])
        except ValueError as exc:
            raise ValueError(f"Ill-formatted pdb resolution: {tok[2]}") from exc

    def joined(field):
        return lambda tok, md, warn: md.__setitem__(field, " ".join(tok[2:]))

    def single(field):
        return lambda tok, md, warn: md.__setitem__(field, tok[2])

    def ec(tok, md, warn):
        found = [m.group() for m in _EC.finditer(tok[2])]
        prelim = [m.group() for m in _EC_PRELIM.finditer(tok[2])]
        if not found and not prelim:
            raise ValueError(f"Did not find a valid EC number, found {tok[2]}")
        if not found and warn:
            warnings.warn(f"Rare EC number(s) {prelim} presumed to be noncatalytic detected!")
        for e in found or prelim:
            if e not in md["ec"]:
                md["ec"].append(e)

    def cath(tok, md, warn):
        found = [m.group() for m in _CATH.finditer(tok[2])]
        if not found:
            raise ValueError(f"Did not find a valid CATH number, found {tok[2]}")
        for c in found:
            if c not in md["cath"]:
                md["cath"].append(c)

    return {
        "ID": single("template_id_string"), "PDB_ID": pdb_id, "UNIPROT_ID": uniprot,
        "MCSA_ID": integer("mcsa_id", "a M-CSA ID"), "CLUSTER": cluster, "ORGANISM_NAME": joined("organism"),
        "ORGANISM_ID": single("organism_id"), "RESOLUTION": resolution,
        "EXPERIMENTAL_METHOD": joined("experimental_method"), "EC": ec, "CATH": cath,
        "ENZYME": joined("enzyme_discription"),
        "REPRESENTING": integer("represented_sites", "the number of represented sites"),
    }


_HANDLERS = _remark_handlers()


@functools.lru_cache(maxsize=1)
def _annotation_tables():
    data = Path(__file__).resolve().parent / "data"
    return {"cath": json.loads((data / "MCSA_CATH_mapping.json").read_text()),
            "ec": json.loads((data / "MCSA_EC_mapping.json").read_text()),
            "sifts": json.loads((data / "pdb_sifts.json").read_text())}


class Template(pyjess.Template):
    """A template with its residues and M-CSA bookkeeping (same constructor and attributes as the
    reference's ``enzymm.template.Template``)."""

    def __init__(self, residues: Sequence[Residue], pdb_id: Optional[str] = None, mcsa_id: Optional[int] = None, *,
                 id: Optional[str] = None, template_id_string: Optional[str] = None,
                 cluster: Optional[Cluster] = None, uniprot_id: Optional[str] = None,
                 organism: Optional[str] = None, organism_id: Optional[str] = None,
                 resolution: Optional[float] = None, experimental_method: Optional[str] = None,
                 enzyme_discription: Optional[str] = None, represented_sites: Optional[int] = None,
                 ec: Iterable[str] = (), cath: Iterable[str] = ()):
        residues = tuple(residues)
        if not residues:
            raise ValueError("Tried creating a `Template` from an empty list of residues!")
        if any(type(r) is not type(residues[0]) for r in residues):
            raise ValueError("Tried creating a `Template` from different types of `Residue` objects.")
        super().__init__([a for r in residues for a in r.atoms], id=id)
        self.residues = residues
        self.pdb_id, self.mcsa_id = pdb_id, mcsa_id
        self.template_id_string, self.cluster = template_id_string, cluster
        self.uniprot_id, self.organism, self.organism_id = uniprot_id, organism, organism_id
        self.resolution, self.experimental_method = resolution, experimental_method
        self.enzyme_discription, self.represented_sites = enzyme_discription, represented_sites
        self.ec = tuple(sorted({*ec, *self._annotations("ec")}))
        self.cath = tuple(sorted({*cath, *self._annotations("cath")}))
        # the residue-orientation rows of the device library's fused hit epilogue, made with the template like its atom rows
        # (pyjess.Template.__init__): Jess(templates) then only concatenates
        self._orient_rows = pyjess._PackedLibrary._orientation_rows(self)

    def _annotations(self, what: str) -> List[str]:
        """EC / CATH numbers known for this template's M-CSA entry and PDB chains
        (reference: template.py:839-885, tables loaded at :1459-1468)."""
        tables = _annotation_tables()
        found: List[str] = []
        if what == "ec":
            if self.mcsa_id is not None:
                found.append(tables["ec"][str(self.mcsa_id)])
        elif self.mcsa_id:
            found.extend(tables["cath"][str(self.mcsa_id)])
        if self.pdb_id is not None if what == "ec" else bool(self.pdb_id):
            for chain in {f"{self.pdb_id}{r.chain_id}" for r in self.residues}:
                entry = tables["sifts"].get(chain)
                if entry is not None:
                    found.extend(v for v in entry.get(what) if v != "?")
        return found

    # -- identity ------------------------------------------------------------
    def _state(self):
        return (self.residues, self.id, self.pdb_id, self.mcsa_id, self.template_id_string, self.cluster,
                self.effective_size, self.uniprot_id, self.organism, self.organism_id, self.resolution,
                self.experimental_method, self.enzyme_discription, self.represented_sites, self.ec, self.cath)

    def __eq__(self, other):
        return self._state() == other._state() if isinstance(other, Template) else NotImplemented

    def __ne__(self, other):
        return self._state() != other._state() if isinstance(other, Template) else NotImplemented

    def __hash__(self):
        return hash((type(self), self._state()))

    def copy(self) -> "Template":
        return type(self)(self.residues, pdb_id=self.pdb_id, mcsa_id=self.mcsa_id, id=self.id,
                          template_id_string=self.template_id_string, cluster=self.cluster,
                          uniprot_id=self.uniprot_id, organism=self.organism, organism_id=self.organism_id,
                          resolution=self.resolution, experimental_method=self.experimental_method,
                          enzyme_discription=self.enzyme_discription, represented_sites=self.represented_sites,
                          ec=self.ec, cath=self.cath)

    __copy__ = copy

    # -- parsing -------------------------------------------------------------
    @classmethod
    def loads(cls, text: str, id: Optional[str] = None, warn: bool = False) -> "Template":
        return cls.load(io.StringIO(text), id=id, warn=warn)

    @classmethod
    def load(cls, file, id: Optional[str] = None, warn: bool = False) -> "Template":
        """Parse one template file: REMARK metadata + ATOM lines in Jess template format."""
        if hasattr(file, "read") or not isinstance(file, (str, os.PathLike)):
            lines = list(file)
        else:
            with open(os.fspath(file)) as handle:
                lines = list(handle)
        md = {"ec": [], "cath": []}
        atoms, seen = [], set()
        for line in lines:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "REMARK":
                handler = _HANDLERS.get(tok[1]) if len(tok) > 1 else None
                if handler is None:
                    continue
                if len(tok) < 3:
                    raise IndexError(f"Expected some annotation after the REMARK {tok[1]} flag")
                if tok[2].upper() in _EMPTY:
                    continue
                handler(tok, md, warn)
            elif tok[0] == "ATOM":
                if line in seen:
                    raise ValueError("Duplicate Atom lines passed!")
                seen.add(line)
                atoms.append(pyjess.TemplateAtom.loads(line))
            elif tok[0] == "HETATM":
                raise ValueError("Supplied template with HETATM record. HETATMs cannot be searched")
        return cls(Residue.construct_residues_from_atoms(atoms), id=id, **md)

    # -- derived properties ----------------------------------------------------
    @cached_property
    def effective_size(self) -> int:
        """Residues that are residue-type specific (match_mode < 100) and not backbone wildcards."""
        return sum(1 for r in self.residues if r.match_mode < 100 and not r.backbone)

    @cached_property
    def multimeric(self) -> bool:
        return len({r.chain_id for r in self.residues}) > 1

    @cached_property
    def relative_order(self) -> List[int]:
        if self.multimeric:
            return [0]
        numbers = [r.residue_number for r in self.residues]
        rank = {v: i + 1 for i, v in enumerate(sorted(set(numbers)))}
        return [rank[n] for n in numbers]


def check_template(template: Template, warn: bool = True) -> bool:
    """Annotation sanity check (template.py:1579-1618); warnings only, never geometry."""
    if not warn:
        return True
    ok = True
    if not template.ec:
        warnings.warn("Could not find EC number annotations")
        ok = False
    if not template.cath:
        warnings.warn("Could not find CATH annotations")
        ok = False
    if template.pdb_id:
        # the template's UniProt accession against PDB-SIFTS, for every PDB chain the template takes residues from
        # (template.py:1595-1615): a mismatch excludes the template from a Matcher built with warn=True
        sifts = _annotation_tables()["sifts"]
        for chain in {f"{template.pdb_id}{r.chain_id}" for r in template.residues}:
            entry = sifts.get(chain)
            if entry:
                if template.uniprot_id != entry["uniprot_id"]:
                    ok = False
                    warnings.warn(f"Different UniProt Accessions {template.uniprot_id} and {entry['uniprot_id']} found")
    return ok


def load_packed_library(path: Union[str, os.PathLike, None] = None, warn: bool = False) -> List[Template]:
    """The shipped jess_templates_20230210 library (7607 templates) from the packed data file."""
    if path is None:
        path = Path(__file__).resolve().parent / "data" / "jess_templates_20230210.txt.gz"
    out, name, buf = [], None, []

    def flush():
        if name is not None:
            out.append(Template.load(buf, warn=warn))

    with gzip.open(path, "rt") as handle:
        for line in handle:
            if line.startswith("@@ "):
                flush()
                name, buf = line[3:].strip(), []
            else:
                buf.append(line)
    flush()
    return out


def load_templates(template_dir: Union[Path, None], warn: bool = False, verbose: bool = False,
                   cpus: int = 0, with_annotations: bool = False) -> Iterator[Template]:
    """Load every ``*.pdb`` under ``template_dir`` (recursively); ``None`` = the shipped library."""
    if template_dir is None:
        yield from load_packed_library(warn=warn)
        return
    template_dir = Path(template_dir)
    if not template_dir.is_dir():
        raise NotADirectoryError(f"The path {template_dir} doesnt exist or is not a directory!")
    paths = [Path(p) for p in glob.glob(f"{template_dir}/**/*.pdb", recursive=True)]
    if not paths:
        raise FileNotFoundError(f"No template files with the .pdb extension found in {template_dir.resolve()}")
    for p in paths:
        try:
            yield Template.load(p, warn=warn)
        except KeyError as exc:
            raise ValueError(f"Passed Template file {p.resolve()} contained issues with some residues.") from exc
