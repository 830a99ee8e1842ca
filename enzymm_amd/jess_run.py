"""Driver of the matching path: the batch counterpart of ``enzymm/jess_run.py``.

Same public names, arguments and results as the reference:

* ``Matcher``  -- groups templates by ``effective_size``, walks the sizes in descending order
  with per-size (rmsd, distance, max_dynamic_distance) (/root/reference/enzymm/jess_run.py:559-736,
  896-988), filters with the logistic models (:867-894);
* ``Match``    -- wraps a ``Hit``: residue triplets, orientation angle, logistic filter,
  completeness, TSV / PDB writers (:59-496);
* ``load_molecules`` (:523-556), including the reference's quirk that ``conserved()`` is called
  and its result discarded (:541-545).

What is different is *how* ``Matcher.run`` reaches the search: the reference builds
``pyjess.Jess(templates)`` and queries once per (molecule, size) on a thread pool
(:785-843, 919-921, 977-981); here every molecule and every size bucket go to the GPU in ONE
``Jess.query_batch`` launch and the per-size sequential logic (skip_smaller_hits, filtering,
completeness, insertion order) is replayed on the returned hit table.
"""
from __future__ import annotations

import collections
import csv
import io
import itertools
import json
import math
import os
import sys
import warnings
from dataclasses import dataclass, field
from functools import cached_property
from pathlib import Path
from typing import IO, ClassVar, Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np

from . import pyjess
from .template import Template, Vec3, check_template

__all__ = ["LogisticRegressionModel", "Match", "Matcher", "load_molecules"]

PROTEINOGENIC_AMINO_ACIDS = [
    "ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE",
    "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL",
]
SPECIAL_AMINO_ACIDS = [
    "ASX", "GLX", "SEC", "PYL", "UNK", "MSE", "SEP", "TPO", "PTR", "HYP",
    "CME", "CSO", "CSD", "PCA", "MLY", "DAL", "DAR", "DSG", "ORN", "PTM",
]
_COUNTED_RESIDUES = frozenset(PROTEINOGENIC_AMINO_ACIDS + SPECIAL_AMINO_ACIDS)

_TSV_HEADER = [
    "query_id", "pairwise_distance", "match_index", "template_pdb_id", "template_pdb_chains",
    "template_cluster_id", "template_cluster_member", "template_cluster_size", "template_effective_size",
    "template_dimension", "template_mcsa_id", "template_uniprot_id", "template_ec", "template_cath",
    "template_multimeric", "query_multimeric", "query_atom_count", "query_residue_count", "rmsd", "log_evalue",
    "orientation", "preserved_order", "completeness", "predicted_correct", "matched_residues",
    "number_of_mutated_residues", "number_of_side_chain_residues_(template,reference)",
    "number_of_metal_ligands_(template,reference)", "number_of_ptm_residues_(template, reference)",
    "total_reference_residues",
]
_ONE_CHAR_ELEMENTS = frozenset("HBCNOFPSKVYIWU")


def ranked_argsort(values: Sequence[int]) -> List[int]:
    rank = {v: i + 1 for i, v in enumerate(sorted(set(values)))}
    return [rank[v] for v in values]


@dataclass(frozen=True)
class LogisticRegressionModel:
    """f(x) = 1 / (1 + exp(-(intercept + coef . x))) with a decision threshold."""

    coefficents: List[float]
    intercept: float
    threshold: float


def _load_models() -> Dict[str, Dict[str, List[LogisticRegressionModel]]]:
    path = Path(__file__).resolve().parent / "data" / "logistic_regression_models.json"
    raw = json.loads(path.read_text())
    return {
        size: {dist: [LogisticRegressionModel(m["coef"], m["intercept"], m["threshold"]) for m in entry["model_list"]]
               for dist, entry in by_dist["pairwise_distance"].items()}
        for size, by_dist in raw["match_size"].items()
    }


@dataclass
class Match:
    """An annotated hit (reference: jess_run.py:59-496)."""

    hit: pyjess.Hit
    complete: bool = field(default=False)
    pairwise_distance: float = field(default=0)
    index: int = field(default=0)
    _logistic_regression_models: ClassVar[Dict[str, Dict[str, List[LogisticRegressionModel]]]] = _load_models()

    # ---- geometry --------------------------------------------------------
    @cached_property
    def atom_triplets(self) -> List[Tuple[pyjess.Atom, pyjess.Atom, pyjess.Atom]]:
        """Matched atoms in the template frame, three per template residue."""
        atoms = self.hit.atoms(transform=True)
        out = []
        for i in range(0, len(atoms), 3):
            trip = tuple(atoms[i:i + 3])
            if len(trip) != 3:
                raise ValueError(f"Failed to construct residues. Got only {len(trip)} ATOM lines")
            owners = {(a.residue_name, a.chain_id, a.residue_number) for a in trip}
            if len(owners) != 1:
                raise ValueError(f"Mixed up atom triplets {owners}. The atoms come from different residues!")
            out.append(trip)
        return out

    @property
    def matched_residues(self) -> List[Tuple[str, str, str]]:
        return [(t[0].residue_name, t[0].chain_id, str(t[0].residue_number)) for t in self.atom_triplets]

    @property
    def truncated(self) -> bool:
        """True when the search of this (template, molecule) pair stopped at ``max_candidates``: the hit is the best
        of the assignments visited, and which ones are visited first is this engine's own order (not Jess's), so the
        reported hit of such a pair may differ from PyJess's.  Not a column of the reference's TSV."""
        return bool(getattr(self.hit, "truncated", False))

    @property
    def multimeric(self) -> bool:
        atoms = self.hit.atoms()
        return any(a.chain_id != atoms[0].chain_id for a in atoms)

    @property
    def preserved_resid_order(self) -> bool:
        if self.hit.template.multimeric or self.multimeric:
            return False
        return ranked_argsort([t[0].residue_number for t in self.atom_triplets]) == self.hit.template.relative_order

    @cached_property
    def match_vector_list(self) -> List[Vec3]:
        """Orientation vector of every matched residue, in the template frame (template.py:213-304)."""
        out = []
        for trip, residue in zip(self.atom_triplets, self.hit.template.residues):
            i, j = residue.orientation_vector_indices
            if j == 9:  # centre atom -> midpoint of the other two
                others = [Vec3.from_xyz(a) for n, a in enumerate(trip) if n != i]
                out.append((others[0] + others[1]) / 2 - Vec3.from_xyz(trip[i]))
            else:
                out.append(Vec3.from_xyz(trip[j]) - Vec3.from_xyz(trip[i]))
        return out

    @property
    def template_vector_list(self) -> List[Vec3]:
        return [r.orientation_vector for r in self.hit.template.residues]

    @property
    def orientation(self) -> float:
        """Mean angle (radians) between template and matched residue orientation vectors."""
        tv, mv = self.template_vector_list, self.match_vector_list
        if len(tv) != len(mv):
            raise ValueError("Vector lists for Template and matching Query structure had different lengths.")
        return sum(a.angle_to(b) for a, b in zip(tv, mv)) / len(tv)

    @property
    def query_atom_count(self) -> int:
        return len(self.hit.molecule())

    @property
    def query_residue_count(self) -> int:
        mol = self.hit.molecule()
        counted = getattr(mol, "_residue_count", None)  # counted on the GPU during ingestion (ingest.PdbBatch)
        if counted is not None:
            return counted
        names = np.char.strip(mol._cols["residue_name"].astype("U3"))
        keep = np.isin(names, list(_COUNTED_RESIDUES))
        return len(set(mol._cols["residue_number"][keep].tolist()))

    # ---- filter -----------------------------------------------------------
    @property
    def predicted_correct(self) -> bool:
        """Majority vote of the logistic models for (effective size, pairwise distance);
        sizes without models (>= 5, < 3) are accepted (jess_run.py:298-346)."""
        size = str(self.hit.template.effective_size)
        if size not in self._logistic_regression_models:
            return True
        try:
            models = self._logistic_regression_models[size][str(self.pairwise_distance)]
            votes = 0
            for m in models:
                z = m.intercept + m.coefficents[0] * self.hit.rmsd + m.coefficents[1] * self.orientation
                votes += (1 / (1 + math.e ** -z)) >= m.threshold
            return bool(votes >= round(len(models) / 2, 0))
        except KeyError as exc:
            raise KeyError("Missing appropriate model parameters to predict correctness. No models for the "
                           f"pairwise distance {self.pairwise_distance} were provided") from exc
        except IndexError as exc:
            raise IndexError("Missing coefficients for both RMSD and Residue Orientation. "
                             "Expecting models with 2 coeficients.") from exc

    def get_identifying_attributes(self) -> Tuple[int, int, int]:
        t = self.hit.template
        return (t.mcsa_id, t.cluster.id, t.dimension)

    # ---- writers ----------------------------------------------------------
    def dumps(self, header: bool = False) -> str:
        buf = io.StringIO()
        self.dump(buf, header=header)
        return buf.getvalue()

    def dump(self, file: IO[str], header: bool = False, predict_correctness: bool = True):
        """One TSV row (jess_run.py:185-284)."""
        w = csv.writer(file, dialect="excel-tab", delimiter="\t", lineterminator="\n")
        if header:
            w.writerow(_TSV_HEADER)
        t = self.hit.template
        c = t.cluster
        row = [
            str(self.hit.molecule().id), str(self.pairwise_distance), str(self.index),
            str(t.pdb_id if t.pdb_id else ""), ",".join(set(r.chain_id for r in t.residues)),
            str(c.id if c else ""), str(c.member if c else ""), str(c.size if c else ""),
            str(t.effective_size), str(t.dimension), str(t.mcsa_id if t.mcsa_id else ""),
            str(t.uniprot_id if t.uniprot_id else ""), ",".join(t.ec if t.ec is not None else ""),
            ",".join(t.cath if t.cath else ""), str(t.multimeric), str(self.multimeric),
            str(self.query_atom_count), str(self.query_residue_count), str(round(self.hit.rmsd, 5)),
            str(round(self.hit.log_evalue, 5)), str(round(self.orientation, 5)), str(self.preserved_resid_order),
            str(self.complete), str(self.predicted_correct) if predict_correctness else "",
            ",".join("_".join(r) for r in self.matched_residues),
        ]
        if hasattr(t, "number_of_mutated_residues"):
            row += [str(t.number_of_mutated_residues), ",".join(map(str, t.number_of_side_chain_residues)),
                    ",".join(map(str, t.number_of_metal_ligands)), ",".join(map(str, t.number_of_ptm_residues)),
                    str(t.total_reference_residues)]
        else:
            row += [""] * 6
        w.writerow(row)

    @staticmethod
    def _atom_line(a: pyjess.Atom) -> str:
        altloc = a.altloc if a.altloc is not None else ""
        name = f" {a.name:<3s}" if a.element in _ONE_CHAR_ELEMENTS else f"{a.name:<4s}"
        return (f"ATOM  {a.serial:>5} {name}{altloc:<1}{a.residue_name:<3}{a.chain_id:>2}{a.residue_number:>4}"
                f"{a.insertion_code:1s}   {a.x:>8.3f}{a.y:>8.3f}{a.z:>8.3f}{a.occupancy:>6.2f}"
                f"{a.temperature_factor:>6.2f}      {a.segment:<4s}{a.element:>2s} \n")

    def dump2pdb(self, file: IO[str], include_query: bool = False, transform: bool = False):
        """Matched atoms (optionally the whole query first) as PDB text (jess_run.py:96-183)."""
        mol = self.hit.molecule()
        if include_query:
            file.write(f"HEADER MOLECULE_ID {mol.id}\n")
            for a in self.hit.molecule(transform=transform):
                file.write(self._atom_line(a))
            file.write("END\n\n")
        t = self.hit.template
        file.write(f"HEADER {self.predicted_correct} MATCH {mol.id} {self.index}\n")
        file.write(f"REMARK TEMPLATE_PDB {t.pdb_id}_{','.join(set(r.chain_id for r in t.residues))}\n")
        if t.cluster:
            file.write(f"REMARK TEMPLATE CLUSTER {t.cluster.id}_{t.cluster.member}_{t.cluster.size}\n")
        if t.represented_sites:
            file.write(f"REMARK TEMPLATE RESIDUES {t.template_id_string}\n")
        file.write(f"REMARK MOLECULE_ID {mol.id}\n")
        file.write(f"REMARK MATCH INDEX {self.index}\n")
        file.write("REMARK TEMPLATE COORDINATE FRAME\n" if transform else "REMARK QUERY COORDINATE FRAME\n")
        for a in self.hit.atoms(transform=transform):
            file.write(self._atom_line(a))
        file.write("END\n\n")


def load_molecules(molecule_paths: List[Path], conservation_cutoff: float = 0, warn: bool = False
                   ) -> List[pyjess.Molecule]:
    """Load query structures; duplicate file stems get ``_2``, ``_3`` ... suffixes.  The files are parsed together
    by the native reader on several host threads (``pyjess.Molecule.load_many``); order, ids, warnings and the
    error for a malformed file are those of a file-by-file loop."""
    seen: collections.Counter = collections.Counter()
    paths, ids = [], []
    for path in molecule_paths:
        text = os.fspath(path)
        name = os.path.basename(text)
        if name:  # (pathlib's rule for .stem on the final component, without building a Path: 5 us per file)
            dot = name.rfind(".")
            stem = name[:dot] if 0 < dot < len(name) - 1 else name
        else:
            stem = Path(path).stem
        seen[stem] += 1
        paths.append(text)
        ids.append(stem if seen[stem] == 1 else f"{stem}_{seen[stem]}")
    molecules = []
    for path, mol in zip(paths, pyjess.Molecule.load_many(paths, ids)):
        if conservation_cutoff:
            # the reference computes the filtered molecule and drops it (jess_run.py:541-545); kept as is
            mol.conserved(conservation_cutoff)
        if mol:
            molecules.append(mol)
        elif warn:
            warnings.warn(f"received an empty molecule from {path}")
    if not molecules and warn:
        warnings.warn("received no molecules from input")
    return molecules


class _BatchHits:
    """The hit table of one searched batch plus what the per-size replay derived from it (completeness, pairwise
    distance); builds ``Match`` objects for a range of kept rows on demand."""

    def __init__(self, templates, molecules, table, rows, complete, pairwise):
        self._templates, self._molecules, self._table = templates, molecules, table
        self._rows, self._complete, self._pairwise = rows, complete, pairwise
        self._cols = None

    def matches(self, lo: int, hi: int) -> List["Match"]:
        c = self._cols
        if c is None:  # columns of the KEPT rows as Python lists, once per batch
            t = self._table[self._rows]
            n = len(t)
            c = self._cols = dict(
                structure=t["structure"].tolist(), template=t["template_sel"].tolist(), n_atoms=t["n_atoms"].tolist(),
                rmsd=t["rmsd"].tolist(), flags=t["flags"].tolist(), orientation=t["orientation"].tolist(),
                n_candidates=t["n_candidates"].tolist(), atoms=t["atoms"].tolist(),
                rot=np.ascontiguousarray(t["rot"]).reshape(n, 3, 3), trans=np.ascontiguousarray(t["trans"]),
                complete=self._complete[self._rows].tolist(), pairwise=self._pairwise[self._rows].tolist())
            self._table = None
        from . import _native
        TRUNC, PRED, CORRECT = _native.TM_HIT_TRUNCATED, _native.TM_HIT_PREDICTED, _native.TM_HIT_CORRECT
        new, Hit = pyjess.Hit.__new__, pyjess.Hit
        out = []
        for i in range(lo, hi):
            h = new(Hit)
            f = c["flags"][i]
            h.template = self._templates[c["template"][i]]
            h._molecule = self._molecules[c["structure"][i]]
            h.rmsd = c["rmsd"][i]
            h._rot = c["rot"][i]
            h._trans = c["trans"][i]
            h._atom_index = tuple(c["atoms"][i][:c["n_atoms"][i]])
            h.n_candidates = c["n_candidates"][i]
            h.truncated = bool(f & TRUNC)
            h.orientation = c["orientation"][i]
            h.predicted = bool(f & CORRECT) if f & PRED else None
            out.append(Match(h, c["complete"][i], c["pairwise"][i]))
        return out


class _LazyMatchList(list):
    """The matches of one molecule.  A plain list to its users; its ``Match`` objects are built from the batch's hit
    table the first time anything but ``len()`` looks at them (at tens of thousands of structures per second,
    building every object eagerly is what the host would spend its time on)."""

    def __init__(self, source: _BatchHits, lo: int, hi: int, first_size: int = 0):
        super().__init__()
        self._source, self._lo, self._hi = source, lo, hi
        #: size bucket of the first match (where the molecule entered the result): Matcher._merged orders by it
        self.first_size = first_size

    def _fill(self):
        src = self._source
        if src is not None:
            self._source = None
            list.extend(self, src.matches(self._lo, self._hi))

    def __len__(self):
        return self._hi - self._lo if self._source is not None else list.__len__(self)

    def __bool__(self):
        return len(self) > 0

    def __reduce__(self):
        self._fill()
        return (list, (list(self),))


def _filling(name):
    method = getattr(list, name)

    def wrapper(self, *args, **kw):
        self._fill()
        return method(self, *args, **kw)
    wrapper.__name__ = name
    return wrapper


for _name in ("__iter__", "__getitem__", "__setitem__", "__delitem__", "__contains__", "__eq__", "__ne__", "__repr__",
              "__reversed__", "__add__", "__iadd__", "__mul__", "__lt__", "__le__", "__gt__", "__ge__", "append",
              "extend", "insert", "pop", "remove", "index", "count", "sort", "reverse", "copy", "clear"):
    setattr(_LazyMatchList, _name, _filling(_name))
del _name


def _default_cpus() -> int:
    return len(os.sched_getaffinity(0)) if sys.platform == "linux" else (os.cpu_count() or 1)


class _GcPaused:
    """The cyclic garbage collector is a process-wide switch; pipelines that want it to rest share this counter: the
    first one in turns it off (if it was on), the last one out turns it back on."""

    def __init__(self):
        import threading
        self._lock = threading.Lock()
        self._depth = 0
        self._was_enabled = False

    def enter(self):
        import gc
        with self._lock:
            if self._depth == 0:
                self._was_enabled = gc.isenabled()
                gc.disable()
            self._depth += 1

    def leave(self):
        import gc
        with self._lock:
            self._depth -= 1
            if self._depth == 0 and self._was_enabled:
                gc.enable()


_gc_paused = _GcPaused()


class Matcher:
    """Match query molecules against a list of templates (reference: jess_run.py:559-1000)."""

    _DEFAULT_JESS_PARAMS = {
        3: {"rmsd": 2, "distance": 0.9, "max_dynamic_distance": 0.9},
        4: {"rmsd": 2, "distance": 1.7, "max_dynamic_distance": 1.7},
        5: {"rmsd": 2, "distance": 2.0, "max_dynamic_distance": 2.0},
        6: {"rmsd": 2, "distance": 2.0, "max_dynamic_distance": 2.0},
        7: {"rmsd": 2, "distance": 2.0, "max_dynamic_distance": 2.0},
        8: {"rmsd": 2, "distance": 2.0, "max_dynamic_distance": 2.0},
    }
    MAX_CANDIDATES = 10000  # jess_run.py:792

    def __init__(self, templates: List[Template], jess_params: Optional[Dict[int, Dict[str, float]]] = None,
                 conservation_cutoff: int = 0, warn: bool = False, verbose: bool = False,
                 skip_smaller_hits: bool = False, match_small_templates: bool = False,
                 cpus: Optional[int] = None, filter_matches: bool = True, console=None,
                 jess_factory=pyjess.Jess):
        self.templates = templates
        self.cpus = _default_cpus() if cpus is None else cpus
        self.conservation_cutoff = conservation_cutoff
        self.warn, self.verbose = warn, verbose
        #: (molecule, atoms) of the molecules this matcher could not search: more atoms than the engine indexes
        #: (65 535, include/enzymm_amd.h); they have no entry in the results of run / run_files
        self.refused: List = []
        self.skip_smaller_hits = skip_smaller_hits
        self.match_small_templates = match_small_templates
        self.filter_matches = filter_matches
        self.jess_params = self._DEFAULT_JESS_PARAMS if jess_params is None else jess_params
        self.console = console
        self._jess_factory = jess_factory

        if len(set(templates)) < len(templates):
            raise ValueError("Duplicate templates were found.")
        if self.cpus <= 0:
            self.cpus = max(1, _default_cpus() + self.cpus)
        self.verbose_print(f"PyJess Version: {pyjess.__version__}")
        pyjess.register_vocabulary(templates)  # device batches created from now on match these templates

        self.templates_by_effective_size: Dict[int, List[Template]] = collections.defaultdict(list)
        for t in templates:
            if check_template(t, warn=self.warn):
                self.templates_by_effective_size[t.effective_size].append(t)
        self.template_effective_sizes = sorted(self.templates_by_effective_size, reverse=True)

        small = [t for s in self.template_effective_sizes if s < 3 for t in self.templates_by_effective_size[s]]
        if self.warn and small:
            tail = ("For small templates Jess parameters for templates of 3 residues will be used."
                    if self.match_small_templates else "These will be excluded since these templates are too general.")
            warnings.warn(f"{len(small)} Templates with an effective size smaller than 3 defined sidechain "
                          f"residues were supplied.\n{tail}")

    def verbose_print(self, *args):
        if self.verbose:
            print(*args)

    def _get_jess_parameters(self, template_size: int) -> Tuple[float, float, float]:
        p = self.jess_params[min(max(template_size, 3), 8)]
        return p["rmsd"], p["distance"], p["max_dynamic_distance"]

    @staticmethod
    def _check_completeness(matches: List[Match]) -> List[Match]:
        """A match is complete when every member 1..size of its template cluster hit the molecule."""
        grouped = collections.defaultdict(list)
        for m in matches:
            t = m.hit.template
            if t.mcsa_id is not None and t.cluster is not None:
                grouped[m.get_identifying_attributes()].append(m)
            else:
                m.complete = True
        for members in grouped.values():
            want = list(range(1, members[0].hit.template.cluster.size + 1))
            if sorted(m.hit.template.cluster.member for m in members) == want:
                for m in members:
                    m.complete = True
        return matches

    @staticmethod
    def _run_jess(molecule: pyjess.Molecule, templates: List[Template], rmsd_threshold: float = 2.0,
                  distance_cutoff: float = 1.5, max_dynamic_distance: float = 1.5, max_candidates: int = 10000,
                  jess_factory=pyjess.Jess) -> List[Match]:
        """Single (molecule, template list) query with the reference's exact call (jess_run.py:785-843)."""
        query = jess_factory(templates).query(
            molecule=molecule, rmsd_threshold=rmsd_threshold, distance_cutoff=distance_cutoff,
            max_dynamic_distance=max_dynamic_distance, max_candidates=max_candidates, best_match=True,
            ignore_chain=True)
        return [Match(hit=h, pairwise_distance=distance_cutoff) for h in query]

    #: ``run`` searches a longer list in batches of this many molecules (or RUN_BATCH_ATOMS atoms), the next batch
    #: flattened and uploaded on a worker thread while the previous one is searched
    RUN_BATCH = 1000
    RUN_BATCH_ATOMS = 8_000_000

    def run(self, molecules: List[pyjess.Molecule], batch_size: Optional[int] = None
            ) -> Dict[pyjess.Molecule, List[Match]]:
        """Search all molecules; larger templates first, sizes < 3 skipped unless match_small_templates.

        ``molecules`` may also be an ``enzymm_amd.ingest.PdbBatch`` (PDB files parsed on the GPU); the
        result is then keyed by its lazy ``Molecule`` objects.  A list longer than ``batch_size`` (default
        ``RUN_BATCH``) goes through the GPU in batches, pipelined like ``run_files``; the result -- its insertion
        order included -- is that of one pass over the whole list."""
        self.refused = []  # (of THIS run: structures beyond the engine's atom limit, see _submit)
        if hasattr(molecules, "native") and hasattr(molecules, "molecules"):
            return self._collect(self._submit(molecules))
        molecules = list(molecules)
        limit = int(batch_size) if batch_size else self.RUN_BATCH
        if len(molecules) <= limit and sum(len(m) for m in molecules) <= self.RUN_BATCH_ATOMS:
            return self._collect(self._submit(molecules))
        from . import ingest
        return self._run_pipelined(ingest.iter_molecule_batches(molecules, limit, self.RUN_BATCH_ATOMS))

    def _submit(self, molecules, stream: int = 0):
        """Launch the search of one batch (every molecule x every searched template, ONE launch) and return at
        once; ``_collect`` waits for it and builds the matches."""
        device_batch = None
        if hasattr(molecules, "native") and hasattr(molecules, "molecules"):  # an ingest.PdbBatch
            device_batch, molecules = molecules, molecules.molecules
            for k, n in getattr(device_batch, "refused", ()):
                self.refused.append((molecules[k], n))
        plan = self._plan()
        sizes, searched, rmsd, dist, dyn, size_of, pairwise, models, jess = plan
        pending = None
        if searched and molecules:
            # the orientation angle and the logistic filter are evaluated in the hit epilogue on the GPU
            # (Hit.orientation, Hit.predicted)
            kw = dict(max_candidates=self.MAX_CANDIDATES, best_match=True, ignore_chain=True, models=models,
                      batch=device_batch)
            if getattr(jess, "query_batch_accepts_async", False):
                pending = jess.query_batch(molecules, rmsd, dist, dyn, _async=True, stream=stream, return_raw=True, **kw)
            else:  # a stand-in engine (tests): synchronous
                pending = pyjess._Ready(jess.query_batch(molecules, rmsd, dist, dyn, **kw))
            if device_batch is None:
                # beyond the engine's atom or residue limit: the molecule was kept in the batch as an empty structure -- no
                # matches -- and is listed here (pyjess.make_batch warns)
                self.refused.extend((molecules[k], n) for k, n in getattr(pending, "refused", ()))
        return plan, device_batch, molecules, pending

    def _collect(self, submitted) -> Dict[pyjess.Molecule, List[Match]]:
        plan, device_batch, molecules, pending = submitted
        sizes, searched, rmsd, dist, dyn, size_of, pairwise, models, jess = plan
        result = pending.result() if pending is not None else [[] for _ in molecules]
        for k in getattr(pending, "unsearched", ()):  # (a kind list beyond the engine's 16-bit positions: no hits, listed)
            self.refused.append((molecules[k], len(molecules[k])))
        table = result if isinstance(result, np.ndarray) else None
        if table is not None:
            if device_batch is not None and hasattr(device_batch, "annotate_hits"):
                device_batch.annotate_hits(table)
            processed = self._collect_table(plan, molecules, table)
            if processed is not None:
                return processed
            result = jess._hits_from_table(table, np.arange(len(searched)), molecules)  # the object-by-object replay
        per_molecule: List[List[pyjess.Hit]] = result
        if table is None and device_batch is not None and hasattr(device_batch, "annotate"):
            device_batch.annotate(mi for mi, hits in enumerate(per_molecule) if hits)

        # Replay of the reference's per-size loop (jess_run.py:930-988) on the hit table.  Only molecules with hits
        # are visited; for each, its hits are bucketed by size once.  The result's insertion order is the
        # reference's: sizes descending, molecules in input order within a size.
        if self.warn:
            cut_short = sum(1 for hits in per_molecule for h in hits if h.truncated)
            if cut_short:
                warnings.warn(f"{cut_short} (template, molecule) searches stopped at max_candidates="
                              f"{self.MAX_CANDIDATES}; their best hit is chosen among the assignments visited "
                              "(Match.truncated), whose order differs from Jess's")
        filtering, skipping = self.filter_matches, self.skip_smaller_hits
        check = self._check_completeness
        buckets: Dict[int, List[Tuple[int, List[Match]]]] = {s: [] for s in sizes}
        for mi, hits in enumerate(per_molecule):
            if not hits:
                continue
            if len(hits) == 1:
                s = size_of[id(hits[0].template)]
                buckets[s].append((mi, [Match(hits[0], False, pairwise[s])]))
                continue
            groups: Dict[int, List[Match]] = {}
            for h in hits:
                s = size_of[id(h.template)]
                groups.setdefault(s, []).append(Match(h, False, pairwise[s]))
            for s, ms in groups.items():
                buckets[s].append((mi, ms))

        processed: Dict[pyjess.Molecule, List[Match]] = collections.defaultdict(list)
        for s in sizes:
            self.verbose_print(f"Now matching query structure(s) to template of size {s}")
            total = 0
            for mi, matches in buckets[s]:
                mol = molecules[mi]
                if skipping and mol in processed:
                    continue
                check(matches)
                if filtering:
                    # the epilogue's verdict when it had a model set for this bucket; otherwise the reference's
                    # Python rule (which also raises KeyError for a distance without models)
                    kept = [m for m in matches
                            if (m.hit.predicted if m.hit.predicted is not None else m.predicted_correct)]
                    if not kept:
                        continue
                    matches = kept
                have = processed.get(mol)
                if have is None:
                    processed[mol] = matches
                else:
                    have.extend(matches)
                total += len(matches)
            self.verbose_print(f"{total} matches found!")
        return processed

    def _template_arrays(self, plan):
        """Per searched template, as arrays: size bucket, pairwise distance of the bucket, completeness group
        (templates sharing bucket, M-CSA id, cluster id and dimension; -1 when a template has no cluster), cluster
        member and size, and whether its bucket has logistic models.  Cached with the plan."""
        cached = getattr(self, "_arrays_cache", None)
        if cached is not None and cached[0] is plan:
            return cached[1]
        sizes, searched, rmsd, dist, dyn, size_of, pairwise, models, jess = plan
        n = len(searched)
        size = np.array([size_of[id(t)] for t in searched], dtype=np.int64)
        groups: Dict[tuple, int] = {}
        group = np.full(n, -1, dtype=np.int64)
        member = np.zeros(n, dtype=np.int64)
        csize = np.zeros(n, dtype=np.int64)
        for i, t in enumerate(searched):
            if t.mcsa_id is not None and t.cluster is not None:
                group[i] = groups.setdefault((int(size[i]), t.mcsa_id, t.cluster.id, t.dimension), len(groups))
                member[i], csize[i] = t.cluster.member, t.cluster.size
        arrays = dict(size=size, pairwise=np.array([pairwise[int(b)] for b in size], dtype=np.float64), group=group,
                      member=member, csize=csize, n_groups=max(1, len(groups)),
                      has_models=(models[0] >= 0) if models is not None else np.zeros(n, dtype=bool))
        self._arrays_cache = (plan, arrays)
        return arrays

    def _collect_table(self, plan, molecules, table) -> Optional[Dict[pyjess.Molecule, List[Match]]]:
        """The per-size replay of the reference (jess_run.py:930-988) done on whole columns of the hit table:
        completeness, filter verdicts, skip_smaller_hits and the insertion order of the result, without touching a
        Python object per hit.  Returns None when a verdict needs the reference's Python rule (no fused models for
        a bucket that has them): the caller then replays object by object."""
        sizes, searched, rmsd, dist, dyn, size_of, pairwise, models, jess = plan
        from . import _native
        A = self._template_arrays(plan)
        flags = table["flags"]
        ts = table["template_sel"].astype(np.int64)
        st = table["structure"].astype(np.int64)
        predicted = (flags & _native.TM_HIT_PREDICTED) != 0
        if self.filter_matches and bool((A["has_models"][ts] & ~predicted).any() if models is not None
                                        else np.isin(A["size"][ts], [int(k) for k in Match._logistic_regression_models]).any()):
            return None
        if self.warn:
            cut_short = int(((flags & _native.TM_HIT_TRUNCATED) != 0).sum())
            if cut_short:
                warnings.warn(f"{cut_short} (template, molecule) searches stopped at max_candidates="
                              f"{self.MAX_CANDIDATES}; their best hit is chosen among the assignments visited "
                              "(Match.truncated), whose order differs from Jess's")
        size = A["size"][ts]
        # completeness: every member 1..size of a template's cluster hit the molecule (within one size bucket)
        group = A["group"][ts]
        complete = group < 0
        idx = np.nonzero(~complete)[0]
        if len(idx):
            key = st[idx] * A["n_groups"] + group[idx]
            uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
            member, want = A["member"][ts[idx]], A["csize"][ts[idx]][first]
            count = np.bincount(inv, minlength=len(uniq))
            distinct = np.bincount(np.unique(inv * (int(member.max()) + 2) + member) // (int(member.max()) + 2),
                                   minlength=len(uniq))
            lowest = np.full(len(uniq), np.iinfo(np.int64).max)
            highest = np.zeros(len(uniq), dtype=np.int64)
            np.minimum.at(lowest, inv, member)
            np.maximum.at(highest, inv, member)
            ok = (count == want) & (distinct == want) & (lowest == 1) & (highest == want)
            complete[idx] = ok[inv]
        keep = np.ones(len(table), dtype=bool)
        if self.filter_matches:
            keep = np.where(predicted, (flags & _native.TM_HIT_CORRECT) != 0, True)
        if self.skip_smaller_hits and keep.any():
            top = np.zeros(len(molecules), dtype=np.int64)
            np.maximum.at(top, st[keep], size[keep])
            keep &= size == top[st]
        rows = np.nonzero(keep)[0]  # (structure, template) order = per molecule: sizes descending, template order
        processed: Dict[pyjess.Molecule, List[Match]] = collections.defaultdict(list)
        if len(rows) == 0:
            return processed
        mol_of_row = st[rows]
        mols, lo = np.unique(mol_of_row, return_index=True)
        hi = np.append(lo[1:], len(rows))
        first_size = size[rows[lo]]  # the bucket in which the molecule got its first kept match
        source = _BatchHits(searched, molecules, table, rows, complete, A["pairwise"][ts])
        for j in np.lexsort((mols, -first_size)).tolist():
            processed[molecules[int(mols[j])]] = _LazyMatchList(source, int(lo[j]), int(hi[j]), int(first_size[j]))
        if self.verbose:
            for s_ in sizes:
                self.verbose_print(f"Now matching query structure(s) to template of size {s_}")
                self.verbose_print(f"{int((size[rows] == s_).sum())} matches found!")
        return processed

    def _plan(self):
        """The flattened search plan (templates by descending size with their bucket's thresholds, the lookup
        tables the per-size replay needs, the logistic model tables).  Depends only on the matcher's settings;
        cached until one of them changes."""
        key = (self.match_small_templates, tuple(self.template_effective_sizes),
               tuple(sorted((k, tuple(sorted(v.items()))) for k, v in self.jess_params.items())))
        cached = getattr(self, "_plan_cache", None)
        if cached is not None and cached[0] == key:
            return cached[1]
        sizes = [s for s in self.template_effective_sizes if s >= 3 or self.match_small_templates]
        searched: List[Template] = []
        rmsd, dist, dyn, bucket_of = [], [], [], []
        for s in sizes:
            r, d, m = self._get_jess_parameters(s)
            for t in self.templates_by_effective_size[s]:
                searched.append(t)
                rmsd.append(r); dist.append(d); dyn.append(m); bucket_of.append(s)
        size_of = {id(t): b for t, b in zip(searched, bucket_of)}           # template -> its size bucket
        pairwise = {s: self._get_jess_parameters(s)[1] for s in sizes}      # bucket -> Match.pairwise_distance
        plan = (sizes, searched, np.array(rmsd, float), np.array(dist, float), np.array(dyn, float), size_of,
                pairwise, self._model_tables(searched, dist), self._jess_factory(searched) if searched else None)
        self._plan_cache = (key, plan)
        return plan

    @staticmethod
    def _model_tables(templates: Sequence[Template], distances: Sequence[float]):
        """Flatten Match._logistic_regression_models for tm_query: one model set per (effective size,
        pairwise distance) that has models; templates of sizes without models get -1 (always correct), and
        so do (size, distance) combinations the JSON lacks -- for those Hit.predicted stays None and the Python
        rule decides (it raises KeyError, jess_run.py:339-342)."""
        table = Match._logistic_regression_models
        sets, offsets, votes, params, index = {}, [0], [], [], []
        lacking = False
        for t, d in zip(templates, distances):
            by_dist = table.get(str(t.effective_size))
            if by_dist is None:
                index.append(-1)
                continue
            models = by_dist.get(str(d))
            if models is None:
                lacking = True
                index.append(-1)
                continue
            key = (str(t.effective_size), str(d))
            if key not in sets:
                sets[key] = len(votes)
                for m in models:
                    params.append((m.coefficents[0], m.coefficents[1], m.intercept, m.threshold))
                offsets.append(len(params))
                votes.append(int(round(len(models) / 2, 0)))
            index.append(sets[key])
        if lacking or not votes:
            return None  # keep the pure-Python filter so the reference's KeyError behaviour is untouched
        return (np.array(index, np.int32), np.array(offsets, np.uint32), np.array(votes, np.uint32),
                np.array(params, np.float64).reshape(-1, 4))

    def run_files(self, paths, batch_size: int = 1000, **ingest_options) -> Dict[pyjess.Molecule, List[Match]]:
        """``run`` over PDB files without host-side parsing, as a three-stage pipeline: worker threads read,
        upload and parse batches i + 1, i + 2 on the GPU (``enzymm_amd.ingest.iter_batches``) while batch i is
        searched on a stream of its own and the matches of batch i - 1 are built here.  ``ingest_options``:
        ``conservation_cutoff`` (default 0: like ``load_molecules``, which computes the filtered molecule and
        drops it), ``ignore_endmdl``, and ``workers`` / ``prefetch`` of ``iter_batches`` (3 / 3).  The result is
        that of ``run(load_molecules(paths))``, insertion order included, keyed by lazy molecules."""
        from . import ingest
        self.refused = []
        return self._run_pipelined(ingest.iter_batches(paths, batch_size=batch_size, **ingest_options))

    def _run_pipelined(self, batches) -> Dict[pyjess.Molecule, List[Match]]:
        """Search device-resident batches (objects with ``native``, ``molecules`` and ``close()``) as they come out
        of ``batches``: batch i is searched on one of two streams while the matches of batch i - 1 are built here
        and whoever feeds the iterator prepares the next ones."""
        from . import _native
        parts: List[Dict[pyjess.Molecule, List[Match]]] = []
        streams = [_native.Stream(pyjess.default_device()) for _ in range(2)]
        in_flight = None  # (batch, submitted) of the previous batch
        # The pipeline builds no reference cycles, but its allocations do trigger full collections, and one of those
        # walks the quarter of a million objects of the template library with the GIL held: 90-300 ms in which none
        # of the three stages moves (thirty batches' worth of search).  The cyclic collector rests until we return
        # (_gc_paused counts the pipelines that are running: two Matchers on two threads do not switch it back on
        # under each other).
        _gc_paused.enter()
        try:
            for n, batch in enumerate(batches):
                try:
                    submitted = self._submit(batch, stream=streams[n % 2].handle)
                except BaseException:
                    batch.close()
                    raise
                previous, in_flight = in_flight, (batch, submitted)
                if previous is not None:
                    try:
                        parts.append(self._collect(previous[1]))
                    finally:
                        previous[0].close()
            if in_flight is not None:
                previous, in_flight = in_flight, None
                try:
                    parts.append(self._collect(previous[1]))
                finally:
                    previous[0].close()
        finally:
            if in_flight is not None:  # an exception above: wait for the launched search, then release
                try:
                    in_flight[1][3].result()
                except Exception:
                    pass
                in_flight[0].close()
            for st in streams:
                st.close()
            _gc_paused.leave()
        return self._merged(parts)

    def _merged(self, parts: List[Dict[pyjess.Molecule, List[Match]]]) -> Dict[pyjess.Molecule, List[Match]]:
        """One result from the results of consecutive batches, in the order one pass over all molecules gives
        (jess_run.py:930-988: a molecule enters the result in the first -- largest -- size bucket that keeps a match
        of it; within a bucket, molecules in input order)."""
        if len(parts) == 1:
            return parts[0]
        size_of = self._plan()[5]
        entries = []
        for part in parts:
            for mol, matches in part.items():
                first = getattr(matches, "first_size", None)
                if first is None:
                    first = size_of[id(matches[0].hit.template)] if len(matches) else 0
                entries.append((-first, mol, matches))
        entries.sort(key=lambda e: e[0])  # stable: batches, and molecules within a batch, stay in input order
        out: Dict[pyjess.Molecule, List[Match]] = collections.defaultdict(list)
        for _, mol, matches in entries:
            out[mol] = matches
        return out

    def run_single(self, molecule: pyjess.Molecule) -> List[Match]:
        return self.run([molecule])[molecule]
