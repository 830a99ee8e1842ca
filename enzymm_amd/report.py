"""Result writers: the TSV table and the per-query PDB files of enzymm's CLI, driven by the hit table.

Mirrors what ``enzymm/_cli.py`` does after ``Matcher.run`` (/root/reference/enzymm/_cli.py:266-316):
one TSV line per match with a 1-based ``match_index`` per query and the header on the first line only,
and one ``<query id>_matches.pdb`` per query.  Row and record formatting is ``Match.dump`` /
``Match.dump2pdb`` (jess_run.py:96-284), whose output is pinned byte for byte by the reference's
fixtures (tests/test_data/results.tsv, 1AMY_matches_*.pdb).
"""
from __future__ import annotations

from pathlib import Path
from typing import IO, Dict, List, Optional, Union

from . import __version__
from .jess_run import Match
from .pyjess import Molecule


def write_tsv(processed: Dict[Molecule, List[Match]], out: Union[str, Path, IO[str]], *,
              predict_correctness: bool = True, version: Optional[str] = None) -> int:
    """Write every match of ``Matcher.run``'s result; returns the number of rows."""
    own = not hasattr(out, "write")
    if own:
        path = Path(out)
        path.parent.mkdir(parents=True, exist_ok=True)
        handle = open(path, "w", newline="", encoding="utf-8")
    else:
        handle = out
    rows = 0
    try:
        handle.write(f"# Version {__version__ if version is None else version}\n")
        for index, (molecule, matches) in enumerate(processed.items()):
            for jndex, match in enumerate(matches):
                match.index = jndex + 1  # 1-indexed matches per query (_cli.py:275)
                # the reference writes the header when index + jndex == 0, i.e. before the very first row
                match.dump(handle, header=(index + jndex == 0), predict_correctness=predict_correctness)
                rows += 1
    finally:
        if own:
            handle.close()
    return rows


def write_pdbs(processed: Dict[Molecule, List[Match]], outdir: Union[str, Path], *, include_query: bool = False,
               transform: bool = False) -> List[Path]:
    """One ``<molecule id>_matches.pdb`` per query (_cli.py:297-316); the directory must not exist yet."""
    outdir = Path(outdir)
    outdir.mkdir(parents=True, exist_ok=False)
    written = []
    for molecule, matches in processed.items():
        path = outdir / f"{molecule.id}_matches.pdb"
        with open(path, "w", encoding="utf-8") as handle:
            for i, match in enumerate(matches):
                match.dump2pdb(handle, include_query=include_query and i == 0, transform=transform)
        written.append(path)
    return written
