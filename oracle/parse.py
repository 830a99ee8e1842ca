"""Minimal, independent parsers feeding the oracle (TEST INFRASTRUCTURE ONLY).

They restate what enzymm obtains from ``pyjess.Molecule.load``
(/root/reference/enzymm/jess_run.py:538-540: "by default it will stop at
ENDMDL") and ``pyjess.TemplateAtom.loads`` (/root/reference/enzymm/template.py:679)
as plain numpy tables.  Column layout of template ATOM lines follows
SURVEY.md 8(a6): cols 7-11 match_mode, 13-16 atom name, 18-20 residue name,
21-22 chain, 23-26 resSeq, 31-54 xyz, 55-60 one-letter residue list, 61-66
distance weight.
"""
from __future__ import annotations

import gzip
import numpy as np

ONE_TO_THREE = {
    "A": "ALA", "R": "ARG", "N": "ASN", "D": "ASP", "C": "CYS", "Q": "GLN", "E": "GLU", "G": "GLY",
    "H": "HIS", "I": "ILE", "L": "LEU", "K": "LYS", "M": "MET", "F": "PHE", "P": "PRO", "S": "SER",
    "T": "THR", "W": "TRP", "Y": "TYR", "V": "VAL",
    # [UNPINNED] non-standard letters that occur in the shipped library
    "X": "XXX", "U": "SEC", "O": "PYL",
}
MAXRESN = 8  # 3-letter name + up to 6 one-letter alternatives (+1 spare)


class OracleMolecule:
    def __init__(self, xyz, name, res, chain, resseq, icode, bfac, serial):
        self.xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        self.name = np.ascontiguousarray(name, dtype="S4")
        self.res = np.ascontiguousarray(res, dtype="S3")
        self.chain = np.ascontiguousarray(chain, dtype="S2")
        self.resseq = np.ascontiguousarray(resseq, dtype=np.int32)
        self.icode = np.ascontiguousarray(icode, dtype="S1")
        self.bfac = np.ascontiguousarray(bfac, dtype=np.float64)
        self.serial = np.ascontiguousarray(serial, dtype=np.int32)

    def __len__(self):
        return len(self.resseq)

    def conserved(self, cutoff):
        """atoms with temperature factor >= cutoff (jess_run.py:541-545 comment)."""
        keep = self.bfac >= cutoff
        return OracleMolecule(self.xyz[keep], self.name[keep], self.res[keep], self.chain[keep],
                              self.resseq[keep], self.icode[keep], self.bfac[keep], self.serial[keep])


def parse_molecule(text: str) -> OracleMolecule:
    xyz, name, res, chain, resseq, icode, bfac, serial = [], [], [], [], [], [], [], []
    for line in text.splitlines():
        if line.startswith("ENDMDL"):
            break
        if not (line.startswith("ATOM  ") or line.startswith("HETATM")):
            continue
        line = line.ljust(80)
        serial.append(int(line[6:11]))
        name.append(line[12:16])
        res.append(line[17:20])
        chain.append(line[20:22])
        resseq.append(int(line[22:26]))
        icode.append(line[26])
        xyz.append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
        b = line[60:66].strip()
        bfac.append(float(b) if b else 0.0)
    return OracleMolecule(np.array(xyz, dtype=np.float64).reshape(-1, 3), name, res, chain, resseq, icode, bfac, serial)


class OracleTemplate:
    def __init__(self, atoms, remarks):
        k = len(atoms)
        self.k = k
        self.remarks = remarks
        self.xyz = np.zeros((k, 3), dtype=np.float64)
        self.mode = np.zeros(k, dtype=np.int32)
        self.name = np.zeros(k, dtype="S4")
        self.resnames = np.zeros((k, MAXRESN), dtype="S3")
        self.nres = np.zeros(k, dtype=np.int32)
        self.chain = np.zeros(k, dtype="S2")
        self.resseq = np.zeros(k, dtype=np.int32)
        self.weight = np.zeros(k, dtype=np.float64)
        for i, line in enumerate(atoms):
            line = line.rstrip("\n").ljust(66)
            self.mode[i] = int(line[6:11])
            self.name[i] = line[12:16]
            names = [line[17:20]] + [ONE_TO_THREE[c] for c in line[54:60].strip()]
            uniq = []
            for nm in names:
                if nm not in uniq:
                    uniq.append(nm)
            self.nres[i] = len(uniq)
            for r, nm in enumerate(uniq):
                self.resnames[i, r] = nm
            self.chain[i] = line[20:22]
            self.resseq[i] = int(line[22:26])
            self.xyz[i] = (float(line[30:38]), float(line[38:46]), float(line[46:54]))
            w = line[60:66].strip()
            self.weight[i] = float(w) if w else 0.0

    @property
    def effective_size(self):
        """template.py:786-808: residues with match_mode < 100 and no ANY/XXX name."""
        n = 0
        for i in range(0, self.k, 3):
            names = {x.decode() for x in self.resnames[i, : self.nres[i]]}
            if self.mode[i] < 100 and not ({"ANY", "XXX"} & names):
                n += 1
        return n

    def remark(self, key):
        for toks in self.remarks:
            if len(toks) > 2 and toks[1] == key:
                return toks[2:]
        return None


def parse_template(text: str) -> OracleTemplate:
    atoms, remarks = [], []
    for line in text.splitlines():
        if line.startswith("ATOM"):
            atoms.append(line)
        elif line.startswith("REMARK"):
            remarks.append(line.split())
    return OracleTemplate(atoms, remarks)


def read_packed_library(path):
    """[(relative_path, text)] from the '@@ path' packed file written by tests/golden/make_fixtures.py."""
    out, name, buf = [], None, []
    with gzip.open(path, "rt") as f:
        for line in f:
            if line.startswith("@@ "):
                if name is not None:
                    out.append((name, "".join(buf)))
                name, buf = line[3:].strip(), []
            else:
                buf.append(line)
    if name is not None:
        out.append((name, "".join(buf)))
    return out
