"""Shared test helpers: the same inputs loaded for the product (enzymm_amd.pyjess) and for the oracle."""
from __future__ import annotations

import functools
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"
LIBRARY = ROOT / "enzymm_amd" / "data" / "jess_templates_20230210.txt.gz"

# Matcher._DEFAULT_JESS_PARAMS, /root/reference/enzymm/jess_run.py:564-571
DEFAULT_DISTANCE = {3: 0.9, 4: 1.7, 5: 2.0, 6: 2.0, 7: 2.0, 8: 2.0}


@functools.lru_cache(maxsize=None)
def library_texts():
    from oracle import parse
    return parse.read_packed_library(LIBRARY)


@functools.lru_cache(maxsize=None)
def oracle_library_templates():
    from oracle import parse
    return [parse.parse_template(text) for _, text in library_texts()]


@functools.lru_cache(maxsize=None)
def product_library_templates():
    from enzymm_amd import pyjess
    return [pyjess.Template.loads(text, id=name) for name, text in library_texts()]


def default_selection():
    """Indices + per-template distance cutoffs of the templates enzymm searches by default
    (effective size >= 3, jess_run.py:932-934), in library order."""
    eff = np.array([t.effective_size for t in oracle_library_templates()])
    idx = np.nonzero(eff >= 3)[0]
    cut = np.array([DEFAULT_DISTANCE[min(int(e), 8)] for e in eff[idx]])
    return idx, cut


def golden_text(rel):
    return (GOLDEN / rel).read_text()
