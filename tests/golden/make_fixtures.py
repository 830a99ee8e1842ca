#!/usr/bin/env python3
"""Regenerate the committed fixtures under tests/golden/ and enzymm_amd/data/.

Run in the build container only (needs /root/reference).  Everything written
here is DATA: query structures, template files and expected outputs that the
reference's own tests hold (tests/test_data/*, the template files named in
tests/test_jess_run.py:26,48,186-223 and tests/test_template.py:186-196), plus
the literal numbers those tests assert (golden.json).  No reference source
code is copied.

The shipped M-CSA template library (enzymm/jess_templates_20230210, 7607
files) is packed into ONE gzip'd text file so the GPU box -- which only
receives /root/repo -- can read it:  each template is stored as

    @@ <relative path>
    <REMARK / ATOM / END lines verbatim>
"""
import gzip
import json
import shutil
import sys
from pathlib import Path

REF = Path("/root/reference")
HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
LIB = REF / "enzymm" / "jess_templates_20230210"

TEST_DATA = [
    "1AMY.pdb",
    "AF-P0DUB6-F1-model_v4.pdb",
    "results.tsv",
    "1AMY_matches_no_query.pdb",
    "1AMY_matches_template.pdb",
    "1AMY_matches_query_included.pdb",
]

# template files/directories the reference tests name explicitly
TEMPLATE_PATHS = [
    "5_residues/results/csa3d_0285",
    "4_residues/results/csa3d_0285",
    "3_residues/results/csa3d_0285",
    "3_residues/results/csa3d_0344",
    "5_residues/results/csa3d_0045/csa3d_0045.cluster_1_1_1.2cxg_A227-A229-A257-A327-A328.template.pdb",
    "3_residues/results/csa3d_0415/csa3d_0415.cluster_1_1_2.1be0_A124-A175-A125-A289-A260.template.pdb",
    "3_residues/results/csa3d_0421/csa3d_0421.cluster_1_1_3.1bf2_A229-A232-A230-A259-A375-A435-A510-A128.template.pdb",
    "3_residues/results/csa3d_0896/csa3d_0896.cluster_2_1_2.2qy1_A135-A179-A137-A230-A175.template.pdb",
    "6_residues/results/csa3d_0001/csa3d_0001.cluster_1_1_1.1b74_A147-AA180-AA70-AA178-AA8-AA7.template.pdb",
    "4_residues/results/csa3d_0011/csa3d_0011.cluster_1_1_3.1qum_D145-D109-D37-D72-D69-D229-D182-D231-D261-D216-D179.template.pdb",
]

# Literal expectations asserted by the reference's tests (file:line cited).
GOLDEN = {
    "_source": "transcribed from /root/reference/tests/test_jess_run.py and tests/test_template.py",
    "G1": {
        "cite": "tests/test_jess_run.py:24-44,61-121",
        "template": "5_residues/results/csa3d_0285/csa3d_0285.cluster_1_1_1.1uh3_A396-A262-A356-A471-A472.template.pdb",
        "molecule": "1AMY.pdb",
        "rmsd_threshold": 2, "distance_cutoff": 1.5, "max_dynamic_distance": 1.5,
        "max_candidates": 10000, "best_match": True,
        "rmsd": 0.32093143, "log_evalue": -3.08424478, "orientation": 0.15327054322,
        "query_atom_count": 3339, "query_residue_count": 403,
        "matched_residues": [["GLU", "A", "204"], ["ASP", "A", "87"], ["ASP", "A", "179"],
                             ["HIS", "A", "288"], ["ASP", "A", "289"]],
        "match_vector_list": [
            [0.2290067979141952, -0.3853409610281773, 0.377114677867322],
            [0.4249816660862038, -0.21966898402981627, -0.3540863184957992],
            [0.45459385444007694, -0.34869961601989985, 0.10687378206512577],
            [-0.8733960645698886, 0.2563504028143271, -0.9840695023070225],
            [-0.510183600042339, -0.1958417994791759, 0.18963368325429997],
        ],
        "preserved_resid_order": True,
        "multimeric": False,
    },
    "G2": {
        "cite": "tests/test_jess_run.py:46-59,122-145",
        "template": "3_residues/results/csa3d_0415/csa3d_0415.cluster_1_1_2.1be0_A124-A175-A125-A289-A260.template.pdb",
        "molecule": "1AMY.pdb",
        "rmsd_threshold": 2, "distance_cutoff": 1, "max_dynamic_distance": 1,
        "max_candidates": 10000, "best_match": True,
        "rmsd": 1.7353479120, "log_evalue": 1.8517964201, "orientation": 1.6503123465442575,
        "matched_residues": [["TRP", "A", "38"], ["HIS", "A", "288"], ["ASP", "A", "289"]],
        "preserved_resid_order": False,
        "multimeric": False,
    },
    "A1": {
        "cite": "tests/test_jess_run.py:186-193,243-267,336-357",
        "templates": TEMPLATE_PATHS[0:0] + [
            "5_residues/results/csa3d_0285/csa3d_0285.cluster_1_1_1.1uh3_A396-A262-A356-A471-A472.template.pdb",
            "5_residues/results/csa3d_0045/csa3d_0045.cluster_1_1_1.2cxg_A227-A229-A257-A327-A328.template.pdb",
            "3_residues/results/csa3d_0285/csa3d_0285.cluster_1_1_2.1uh3_A396-A262-A356-A471-A472.template.pdb",
            "3_residues/results/csa3d_0285/csa3d_0285.cluster_1_2_2.1uh3_A396-A262-A356-A471-A472.template.pdb",
            "3_residues/results/csa3d_0421/csa3d_0421.cluster_1_1_3.1bf2_A229-A232-A230-A259-A375-A435-A510-A128.template.pdb",
            "3_residues/results/csa3d_0896/csa3d_0896.cluster_2_1_2.2qy1_A135-A179-A137-A230-A175.template.pdb",
        ],
        "molecule": "1AMY.pdb",
        "jess_params": {"3": [2, 1.2, 1.2], "4": [2, 1.7, 1.7], "5": [2, 2.0, 2.0],
                        "6": [2, 2.0, 2.0], "7": [2, 2.0, 2.0], "8": [2, 2.0, 2.0]},
        "unfiltered_pdb_ids": ["1bf2", "1uh3", "1uh3", "1uh3", "2cxg", "2qy1"],
        "filtered_pdb_ids": ["1bf2", "1uh3", "1uh3", "1uh3", "2cxg"],
    },
    "A2": {
        "cite": "tests/test_jess_run.py:204-234,301-329",
        "template_dirs": ["5_residues/results/csa3d_0285", "4_residues/results/csa3d_0285"],
        "molecules": ["1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb"],
        "kept_matches": {"1AMY": 2, "AF-P0DUB6-F1-model_v4": 2},
        "kept_matches_skip_smaller_hits": {"1AMY": 1, "AF-P0DUB6-F1-model_v4.conserved(80)": 1},
        "query_residue_count": {"AF-P0DUB6-F1-model_v4": 511, "AF-P0DUB6-F1-model_v4.conserved(80)": 494},
        "template_effective_sizes": [5, 4],
    },
    "A3": {
        "cite": "tests/test_jess_run.py:218-241,331-334",
        "template_dirs": ["5_residues/results/csa3d_0285", "4_residues/results/csa3d_0285",
                          "3_residues/results/csa3d_0344"],
        "molecule": "1AMY.pdb", "match_small_templates": True, "kept_matches": 3,
    },
    "A5": {
        "cite": "tests/test_template.py:335-395; tests/test_template.py:34-36",
        "library_template_count": 7607,
        "1b74": {"path": TEMPLATE_PATHS[8], "effective_size": 6, "dimension": 6, "multimeric": True,
                 "relative_order": [0], "pdb_id": "1b74", "mcsa_id": 1, "cluster": [1, 1, 1],
                 "uniprot_id": "P56868", "organism": "Aquifex pyrophilus", "organism_id": "2714",
                 "resolution": 2.3, "represented_sites": 2,
                 "template_id_string": "1b74_A147-AA180-AA70-AA178-AA8-AA7"},
        "1qum": {"path": TEMPLATE_PATHS[9], "effective_size": 4, "dimension": 4, "multimeric": False,
                 "relative_order": [2, 1, 4, 3], "pdb_id": "1qum", "mcsa_id": 11, "cluster": [1, 1, 3],
                 "uniprot_id": None, "organism": "Escherichia coli", "organism_id": "562",
                 "resolution": 1.55, "represented_sites": 1},
    },
    "survey_probe_library_effective_sizes": {
        "cite": "SURVEY.md 3.4 [PROBE], not a reference assertion",
        "0": 32, "1": 176, "2": 614, "3": 2703, "4": 1840, "5": 1047, "6": 656, "7": 359, "8": 180,
    },
}


def main():
    if not REF.is_dir():
        sys.exit("needs /root/reference (build container only)")
    # 1. query structures + expected outputs
    for name in TEST_DATA:
        shutil.copyfile(REF / "tests" / "test_data" / name, HERE / name)
    bad = HERE / "bad_templates"
    bad.mkdir(exist_ok=True)
    for p in sorted((REF / "tests" / "test_data" / "bad_templates").glob("*.pdb")):
        shutil.copyfile(p, bad / p.name)
    # 2. the template files the tests name
    tdir = HERE / "templates"
    for rel in TEMPLATE_PATHS:
        src = LIB / rel
        files = sorted(src.glob("*.pdb")) if src.is_dir() else [src]
        for f in files:
            dst = tdir / f.relative_to(LIB)
            dst.parent.mkdir(parents=True, exist_ok=True)
            shutil.copyfile(f, dst)
    # 3. literal expectations
    (HERE / "golden.json").write_text(json.dumps(GOLDEN, indent=1) + "\n")
    # 4. packed library
    files = sorted(LIB.glob("**/*.pdb"))
    assert len(files) == 7607, len(files)
    out = REPO / "enzymm_amd" / "data" / "jess_templates_20230210.txt.gz"
    with gzip.GzipFile(out, "wb", compresslevel=9, mtime=0) as gz:
        for f in files:
            gz.write(f"@@ {f.relative_to(LIB).as_posix()}\n".encode())
            gz.write(f.read_bytes())
            if not f.read_bytes().endswith(b"\n"):
                gz.write(b"\n")
    print("packed", len(files), "templates ->", out, out.stat().st_size, "bytes")
    # 5. residue pool for the synthetic-structure generator (enzymm_amd/synthetic.py): the protein
    #    ATOM records of the two test structures as plain arrays
    import numpy as np
    cols = {k: [] for k in ("xyz", "name", "res", "resid", "source")}
    for src, fname in enumerate(("1AMY.pdb", "AF-P0DUB6-F1-model_v4.pdb")):
        last, rid = None, -1
        for line in (REF / "tests" / "test_data" / fname).read_text().splitlines():
            if line.startswith("ENDMDL"):
                break
            if not line.startswith("ATOM  "):
                continue
            key = (line[20:22], line[22:27])
            if key != last:
                rid, last = rid + 1, key
            cols["xyz"].append((float(line[30:38]), float(line[38:46]), float(line[46:54])))
            cols["name"].append(line[12:16])
            cols["res"].append(line[17:20])
            cols["resid"].append(rid)
            cols["source"].append(src)
    np.savez_compressed(REPO / "enzymm_amd" / "data" / "residue_pool.npz",
                        xyz=np.array(cols["xyz"], dtype=np.float64), name=np.array(cols["name"], dtype="S4"),
                        res=np.array(cols["res"], dtype="S3"), resid=np.array(cols["resid"], dtype=np.int32),
                        source=np.array(cols["source"], dtype=np.int8))
    # logistic models are data the filter (jess_run.py:499-520) needs
    # ... and the M-CSA / SIFTS id -> EC / CATH tables Template.ec / Template.cath are enriched from
    # (template.py:839-885, loaded at :1459-1468)
    for name in ("logistic_regression_models.json", "MCSA_CATH_mapping.json", "MCSA_EC_mapping.json",
                 "pdb_sifts.json"):
        shutil.copyfile(REF / "enzymm" / "data" / name, REPO / "enzymm_amd" / "data" / name)


if __name__ == "__main__":
    main()
