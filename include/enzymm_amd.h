/*
 * enzymm_amd.h -- C ABI of libenzymm_amd.so, the MI355X (gfx950) template matcher
 * that replaces the PyJess call inside enzymm/jess_run.py.
 *
 * Plain pointers and sizes only; no C++/torch types cross this boundary.  All
 * functions return TM_OK (0) or a negative TM_E_* code; none throws.
 *
 * What each entry point replaces in the reference (/root/reference):
 *
 *   tm_library_create   pyjess.Jess(templates)               enzymm/jess_run.py:800-802
 *                       (+ pyjess.Template/TemplateAtom      enzymm/template.py:535, 679)
 *   tm_batch_create     the atom table pyjess.Molecule.load  enzymm/jess_run.py:538-540
 *                       yields, flattened for many molecules
 *   tm_query            Jess.query(molecule, rmsd_threshold, enzymm/jess_run.py:803-811,
 *                       distance_cutoff, max_dynamic_        tests/test_jess_run.py:37-39
 *                       distance, max_candidates,
 *                       best_match, ignore_chain)
 *   tm_result_*         iteration over the pyjess.Hit        enzymm/jess_run.py:839-843
 *                       objects (rmsd, atoms, transform)     enzymm/jess_run.py:348-373
 *
 * The library is semantics-free with respect to Jess "match modes": the host
 * (enzymm_amd/pyjess.py) turns each template atom's match_mode / atom names /
 * residue names into two bit masks over small integer vocabularies, and encodes
 * every target atom with the ids of its residue name and atom name.
 *
 * Threading: handles are not internally locked; use one tm_batch per thread.
 * A tm_library is immutable after creation and may be shared.
 */
#ifndef ENZYMM_AMD_H
#define ENZYMM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TM_ABI_VERSION 1

/* limits of this build */
#define TM_MAX_TEMPLATE_ATOMS 24   /* 8 residues x 3 atoms: enzymm's largest size bucket (jess_run.py:564-571) */
#define TM_MAX_TEMPLATE_GROUPS 8   /* template residues */
#define TM_MAX_STRUCTURE_ATOMS 16777215   /* 2^24 - 1: beyond it a structure is refused (kept in its batch as an empty one) */
#define TM_NARROW_STRUCTURE_ATOMS 65535   /* up to here the 16-bit atom indices of the tuned kernels; a larger ("wide")
                                             structure is searched by the template-at-a-time launch with 24-bit indices */
#define TM_MAX_STRUCTURE_RESIDUES 65533  /* residue numbers 0xfffd..0xffff are reserved by the search kernel */
#define TM_RES_OTHER 31            /* residue-type id of "not in the vocabulary" */
#define TM_NAME_OTHER 63           /* atom-name id of "not in the vocabulary" */

enum {
    TM_OK = 0,
    TM_E_INVALID = -1,     /* bad argument */
    TM_E_UNSUPPORTED = -2, /* over a TM_MAX_* limit or an unimplemented flag combination */
    TM_E_DEVICE = -3,      /* HIP runtime error (tm_last_error() has the text) */
    TM_E_NOMEM = -4,
    TM_E_OVERFLOW = -5     /* result tables too small: retry with the sizes tm_result_needed() reports */
};

enum {
    TM_FLAG_IGNORE_CHAIN = 1u,     /* Jess.query(ignore_chain=True), jess_run.py:810 */
    TM_FLAG_BEST_MATCH = 2u,       /* Jess.query(best_match=True), jess_run.py:809; off: every hit within rmsd_threshold */
    TM_FLAG_COUNT_CANDIDATES = 4u, /* also record the candidate count of every (structure, template) pair */
    TM_FLAG_REBUILD_INDEX = 8u     /* recompute the per-structure candidate index even if cached */
};

/* bits of tm_hit.flags */
enum {
    TM_HIT_TRUNCATED = 1u,  /* more than max_candidates complete assignments exist; search was cut short */
    TM_HIT_PREDICTED = 2u,  /* logistic filter evaluated for this hit (orientation table + model set present) */
    TM_HIT_CORRECT = 4u     /* ... and it predicts the hit as correct */
};

typedef struct tm_library tm_library;
typedef struct tm_batch tm_batch;
typedef struct tm_result tm_result;

/* Templates, concatenated; every per-atom array is in TEMPLATE (file) order. */
typedef struct tm_library_desc {
    uint32_t n_templates;
    const uint32_t *atom_offsets; /* [n_templates + 1] */
    const double *xyz;            /* [n_atoms * 3] template coordinates */
    const double *dist_weight;    /* [n_atoms] cols 61-66 of the template ATOM line */
    const uint32_t *res_mask;     /* [n_atoms] bit r: residue-type id r admissible */
    const uint64_t *name_mask;    /* [n_atoms] bit a: atom-name id a admissible */
    const uint8_t *any_residue;   /* [n_atoms] 1: residue type not checked (match_mode >= 100) */
    const uint8_t *group;         /* [n_atoms] template residue ordinal (atoms sharing chain + resSeq) */
    const uint8_t *chain;         /* [n_atoms] template chain ordinal */
    /* Optional residue-orientation table for the fused hit epilogue (enzymm/template.py:213-304,
     * jess_run.py:425-478); NULL: tm_hit.orientation is NaN.  Template residue r covers template atoms
     * 3r..3r+2.  orient_first/orient_second index into that triplet; orient_second == 9 means "from atom
     * orient_first to the midpoint of the other two".  orient_vec is the template-frame vector. */
    const uint32_t *residue_offsets; /* [n_templates + 1] */
    const uint8_t *orient_first;     /* [n_residues] */
    const uint8_t *orient_second;    /* [n_residues] */
    const double *orient_vec;        /* [n_residues * 3] */
} tm_library_desc;

/* Target structures, concatenated; atoms in file order. */
typedef struct tm_batch_desc {
    uint32_t n_structures;
    const uint32_t *atom_offsets; /* [n_structures + 1] */
    const double *xyz;            /* [n_atoms * 3] */
    const uint8_t *res_type;      /* [n_atoms] residue-type id, TM_RES_OTHER if unknown */
    const uint8_t *name_id;       /* [n_atoms] atom-name id, TM_NAME_OTHER if unknown */
    const uint32_t *residue;      /* [n_atoms] residue ordinal within its structure: equal (chain, resSeq, iCode) <=> equal ordinal */
    const uint8_t *chain;         /* [n_atoms] chain ordinal within its structure */
} tm_batch_desc;

/* Target structures as the raw columns of their atom tables (what Molecule.load yields); the library encodes them
 * with the vocabulary, on a team of host threads (tm_batch_create_from_columns). */
typedef struct tm_columns_desc {
    uint32_t n_structures;
    const uint32_t *atom_offsets;   /* [n_structures + 1] */
    const double *xyz;              /* [n_atoms * 3] */
    const char *name;               /* [n_atoms * 4] raw atom-name fields */
    const char *residue_name;       /* [n_atoms * 3] */
    const char *chain_id;           /* [n_atoms * 2] */
    const int32_t *residue_number;  /* [n_atoms] */
    const char *insertion_code;     /* [n_atoms] */
} tm_columns_desc;

/* One target structure as the raw columns of its atom table, wherever they lie in host memory: the hand-over of
 * Molecule objects to Jess.query (enzymm/jess_run.py:800-811, molecules from load_molecules :523-556) for a whole
 * list of molecules at once, without first copying their columns together (tm_batch_create_from_structures). */
typedef struct tm_structure_columns {
    uint64_t n_atoms;
    const double *xyz;              /* [n_atoms * 3] */
    const char *name;               /* [n_atoms * 4] raw atom-name fields */
    const char *residue_name;       /* [n_atoms * 3] */
    const char *chain_id;           /* [n_atoms * 2] */
    const int32_t *residue_number;  /* [n_atoms] */
    const char *insertion_code;     /* [n_atoms] */
} tm_structure_columns;

/* How raw PDB fields map to the ids of the host's vocabulary (tm_batch_create_from_pdb). */
typedef struct tm_vocab {
    uint32_t n_atom_names;
    const char *atom_names;       /* [n * 4] raw 4-character name fields (cols 13-16) */
    const uint8_t *atom_name_ids; /* [n] */
    uint32_t n_elements;
    const char *elements;         /* [n * 2] element columns (cols 13-14) for names not listed above */
    const uint8_t *element_ids;   /* [n] */
    uint32_t n_residues;
    const char *residue_names;    /* [n * 3] raw residue-name fields (cols 18-20) */
    const uint8_t *residue_ids;   /* [n] */
    /* Optional: residue names whose distinct residue numbers are counted per structure during ingestion
     * (enzymm's Match.query_residue_count, jess_run.py:488-497); 0 / NULL: tm_batch_residue_counts is unavailable. */
    uint32_t n_counted_residues;
    const char *counted_residue_names; /* [n * 3] */
} tm_vocab;

enum { TM_PDB_IGNORE_ENDMDL = 1u };

typedef struct tm_query_params {
    uint32_t n_selected;                /* templates to search */
    const uint32_t *template_index;     /* [n_selected] indices into the library */
    const double *rmsd_threshold;       /* [n_selected] */
    const double *distance_cutoff;      /* [n_selected] */
    const double *max_dynamic_distance; /* [n_selected] */
    uint32_t max_candidates;            /* per (structure, template) */
    uint32_t flags;                     /* TM_FLAG_* */
    uint64_t hit_capacity;              /* rows of the hit table / pairs with candidates; 0: automatic */
    void *stream;                       /* hipStream_t to launch on, NULL = default stream */
    uint64_t candidate_capacity;        /* complete assignments buffered for superposition; 0: automatic */
    /* Optional logistic filter fused into the hit epilogue (enzymm/jess_run.py:298-346): model set m holds
     * models model_offsets[m] .. model_offsets[m+1]-1, each {coef_rmsd, coef_orientation, intercept, threshold};
     * a hit is "predicted correct" when at least model_votes[m] of them fire.  model_set[i] < 0: no models for
     * selected template i (always predicted correct). */
    uint32_t n_model_sets;
    const int32_t *model_set;           /* [n_selected] or NULL */
    const uint32_t *model_offsets;      /* [n_model_sets + 1] */
    const uint32_t *model_votes;        /* [n_model_sets] */
    const double *models;               /* [n_models * 4] */
} tm_query_params;

/* One hit of a (structure, template) pair (the best one under TM_FLAG_BEST_MATCH). */
typedef struct tm_hit {
    uint32_t structure;    /* index into the batch */
    uint32_t template_sel; /* index into tm_query_params.template_index */
    uint32_t n_candidates; /* complete assignments evaluated for this pair */
    uint32_t flags;        /* TM_HIT_* */
    double rmsd;
    double rot[9];   /* row-major R: x_template = R * x_target + trans */
    double trans[3];
    uint32_t n_atoms;
    uint32_t atoms[TM_MAX_TEMPLATE_ATOMS]; /* target atom (file-order index within its structure) per template atom */
    uint32_t reserved;
    double orientation; /* mean angle (rad) between template and matched residue orientation vectors; NaN if unknown */
} tm_hit;

int tm_abi_version(void);
const char *tm_last_error(void);
int tm_device_count(int *count);
/* Host threads one native call of this process may use: $ENZYMM_AMD_HOST_THREADS, else min(hardware threads, affinity
 * mask, cgroup CPU quota, 16) / $LOCAL_WORLD_SIZE -- the ranks of a node (one process per GPU) share its host. */
int tm_host_threads(void);
/* A stream of its own for tm_query_params.stream (non-blocking with respect to the null stream): queries on
 * different streams overlap, e.g. the next batch's search with fetching the previous batch's hits. */
int tm_stream_create(int device, void **stream);
int tm_stream_destroy(int device, void *stream);

int tm_library_create(int device, const tm_library_desc *desc, tm_library **out);
void tm_library_destroy(tm_library *lib);

int tm_batch_create(int device, const tm_batch_desc *desc, tm_batch **out);
struct tm_vocab;
int tm_batch_create_from_columns(int device, const tm_columns_desc *desc, const struct tm_vocab *vocab, tm_batch **out);
struct tm_structure_columns;
int tm_batch_create_from_structures(int device, uint32_t n_structures, const struct tm_structure_columns *structures,
                                    const struct tm_vocab *vocab, tm_batch **out);
/* Device-side ingestion: raw PDB text is uploaded and parsed on the GPU (ATOM/HETATM records up to the first
 * ENDMDL, optional temperature-factor cutoff) -- replaces pyjess.Molecule.load + Molecule.conserved for batches,
 * enzymm/jess_run.py:538-545.  A file the device path cannot lay out (a residue whose records are not contiguous,
 * or more than 16384 residue runs) is flattened on the host inside this call; the batch is still one tm_batch. */
int tm_batch_create_from_pdb(int device, uint32_t n_files, const char *const *text, const uint64_t *length,
                             const tm_vocab *vocab, uint32_t flags, double min_bfactor, tm_batch **out);
/* The same from file paths: the files are read on a team of host threads straight into one page-locked buffer and
 * uploaded in a single copy -- the loop of load_molecules (enzymm/jess_run.py:523-556) without a Python-side read.
 * *bad_file = first file that could not be read (-1: none).  The text stays available (for reporting the matched
 * atoms) through tm_batch_file_text until the batch is destroyed. */
int tm_batch_create_from_files(int device, uint32_t n_files, const char *const *paths, const tm_vocab *vocab,
                               uint32_t flags, double min_bfactor, tm_batch **out, int64_t *bad_file);
int tm_batch_file_text(tm_batch *batch, uint32_t file, const char **text, uint64_t *length);
/* the whole text buffer: file f occupies [offsets[f], offsets[f + 1] - 1) of it; offsets has n_structures + 1 entries */
int tm_batch_text_layout(tm_batch *batch, const char **base, uint64_t *offsets);
/* pyjess.Molecule.load takes any PDB file (/root/reference/enzymm/jess_run.py:538).  A structure of up to
 * TM_NARROW_STRUCTURE_ATOMS atoms runs on the tuned kernels (16-bit atom indices); a larger one, up to
 * TM_MAX_STRUCTURE_ATOMS, is searched as well -- template at a time, its records in global memory, 24-bit indices --
 * with the same results at a lower rate.  Structures beyond TM_MAX_STRUCTURE_ATOMS are kept in the batch as EMPTY
 * structures -- no hits, the other structures of the batch unaffected -- and listed here with the atom counts they came
 * with.  *count receives how many there are; up to `capacity` of them are written.
 * The same holds for a structure of more than TM_MAX_STRUCTURE_RESIDUES residues (the kernels number a structure's
 * residues with 16 bits): emptied, listed, the rest of its batch searched -- on all three ways into a batch. */
int tm_batch_refused(const tm_batch *batch, uint32_t *structures, uint32_t *atoms, uint32_t capacity, uint32_t *count);
int tm_batch_atom_counts(tm_batch *batch, uint32_t *atom_offsets);            /* [n_structures + 1] */
/* tm_batch_create_from_structures / _from_columns return when their host-to-device copies are QUEUED (17 bytes per atom:
 * one packed word, a chain ordinal, the coordinates as int32 thousandths of an Angstrom when they are such values exactly,
 * else the structure's fp64 values; laid out in HBM by a kernel behind the copies); tm_query waits for them on its stream.
 * tm_batch_wait blocks until the batch is resident; tm_batch_upload_bytes reports what the constructor sent. */
int tm_batch_wait(tm_batch *batch);
int tm_batch_upload_bytes(const tm_batch *batch, uint64_t *bytes);
/* Diagnostic: the batch as it lies in HBM (any pointer may be NULL) -- fp64 coordinates [n_atoms * 3], records
 * [n_atoms * 4] (centred fp32 xyz + packed word), chain ordinals, file-order indices, per structure the offset of its
 * residue table [n_structures + 1] and the tables themselves (local index of every residue's first atom, n_res + 1 each).
 * Atoms are residue-contiguous.  Tests compare the layouts the different constructors produce. */
int tm_batch_copy_layout(tm_batch *batch, double *xyz, float *records, uint8_t *chain, uint32_t *orig,
                         uint32_t *residue_offsets, uint16_t *residue_start);
int tm_batch_line_offsets(tm_batch *batch, uint32_t *out, uint64_t n);        /* byte offset of each kept record in its file */
/* the same for n chosen records (indices into the batch's concatenated atoms, e.g. the matched atoms of a hit table):
 * a gather on the device instead of a copy of every offset */
int tm_batch_line_offsets_at(tm_batch *batch, const uint32_t *records, uint64_t n, uint32_t *out);
int tm_batch_residue_counts(tm_batch *batch, uint32_t *out);                  /* [n_structures]; TM_E_UNSUPPORTED without tm_vocab.counted_residue_names */
/* Batches made by tm_batch_create_from_files: the text records of the atoms a hit table matched -- what a report row of
 * enzymm needs of a matched molecule (Match.dump / dump2pdb, /root/reference/enzymm/jess_run.py:203-265) -- without a
 * host-side parse of the file and without asking the device: per structure the distinct matched atoms, ascending, as
 * keys[i] = structure << 32 | atom (file-order index) with the first `width` bytes of the atom's record in
 * lines[i * width ..] (padded with newlines past the end of the text).  *count = rows written (or needed, with
 * TM_E_OVERFLOW, when capacity is too small; the sum of tm_hit.n_atoms always suffices). */
int tm_batch_pick_records(tm_batch *batch, const tm_hit *hits, uint64_t n_hits, uint32_t width, uint64_t capacity,
                          uint64_t *keys, uint8_t *lines, uint64_t *count);
void tm_batch_destroy(tm_batch *batch);

/* Asynchronous on params->stream; the result owns the device-side hit table. */
int tm_query(const tm_library *lib, tm_batch *batch, const tm_query_params *params, tm_result **out);
int tm_result_wait(tm_result *res);                          /* blocks; TM_E_OVERFLOW if a result table or the batch's
                                                                candidate-index pool filled up: query again */
int tm_result_needed(tm_result *res, uint64_t *candidates, uint64_t *pairs); /* sizes a retry needs */
/* Structures of the batch this query could NOT search with its library: a wide structure (more than
 * TM_NARROW_STRUCTURE_ATOMS atoms) with more than 65535 atoms admissible to one template-atom kind -- the pair tables
 * hold list positions in 16 bits.  Such a structure gets empty candidate lists (no hits), the rest of its batch is
 * searched as usual and the query succeeds (until round 5 it ended with TM_E_UNSUPPORTED for the whole batch).  The
 * reference's pyjess.Jess.query would have searched it (/root/reference/enzymm/jess_run.py:803-811): the caller must
 * report these, as enzymm_amd.pyjess and Matcher.refused do.  Implies wait; *count = how many, up to `capacity` written. */
int tm_result_unsearched(tm_result *res, uint32_t *structures, uint32_t capacity, uint32_t *count);
int tm_result_hit_count(tm_result *res, uint64_t *n_hits);   /* implies wait */
int tm_result_copy_hits(tm_result *res, tm_hit *out, uint64_t capacity); /* sorted by (structure, template_sel, atoms) */
int tm_result_copy_candidates(tm_result *res, uint32_t *out, uint64_t n); /* [n_structures * n_selected], needs TM_FLAG_COUNT_CANDIDATES */
/* HIP-event durations on the launch stream: candidate index + template staging, k_search, superposition + selection */
int tm_result_timing(tm_result *res, float *index_ms, float *search_ms, float *finish_ms);
/* search_ms split into its two launches: the block search over the pair tables (k_search; with the template-at-a-time
 * launch, when the selection needs one) and the depth-first part below the surviving triples (k_deep) */
int tm_result_search_split(tm_result *res, float *front_ms, float *deep_ms);
/* Diagnostic: the most pages of surviving triples any structure of the query listed between the block search and the
 * depth-first part (64 triples to a page), and the triples in all pages.  Waits for the result. */
int tm_result_page_census(tm_result *res, uint32_t *max_pages, uint64_t *triples);
/* Waits until the rows of the last tm_result_append_hits of this result are in the table (an event behind that copy,
 * not the stream: later queries queued on the same stream keep running). */
int tm_result_wait_appended(tm_result *res);
const void *tm_result_device_hits(tm_result *res);           /* device pointer to the unsorted tm_hit table */
/* Appends this query's hit rows to a caller-owned DEVICE table (rows: tm_hit[capacity], *count: rows in use, both in
 * device memory), adding structure_offset to tm_hit.structure -- enqueued on the query's stream behind its kernels,
 * no host wait.  A rank of a sharded run (one process per GPU) collects the rows of all its batches this way and
 * sends them to rank 0 in ONE gather per pass: the single merge step of the reference's sharded flow
 * (nextflow/template_matcher.nf:65-78, collectFile).  Rows beyond capacity are dropped; *count still advances, so
 * count > capacity tells the caller to grow the table. */
int tm_result_append_hits(tm_result *res, void *rows, uint64_t capacity, uint64_t *count, uint32_t structure_offset);
void tm_result_destroy(tm_result *res);

/* ---- host-side PDB reader (no GPU involved) -----------------------------------------------------------
 * The atom table pyjess.Molecule.load builds for ONE file (enzymm/jess_run.py:538): ATOM/HETATM records up to
 * the first ENDMDL, fixed columns, decimal fields converted exactly.  Two calls: count, then fill the caller's
 * arrays.  A malformed record (shorter than 54 columns, or a numeric field that is not [spaces][sign]digits
 * [.digits][spaces]) gives TM_E_INVALID with the offending line's byte range in *bad_begin / *bad_length. */
typedef struct tm_pdb_columns {
    int32_t *serial;            /* [n] */
    char *name;                 /* [n * 4] cols 13-16, raw */
    char *altloc;               /* [n] */
    char *residue_name;         /* [n * 3] */
    char *chain_id;             /* [n * 2] cols 21-22 */
    int32_t *residue_number;    /* [n] */
    char *insertion_code;       /* [n] */
    double *xyz;                /* [n * 3] */
    double *occupancy;          /* [n] blank = 0 */
    double *temperature_factor; /* [n] blank = 0 */
    char *segment;              /* [n * 4] cols 73-76, stripped, NUL padded */
    char *element;              /* [n * 2] cols 77-78, stripped, NUL padded */
    int8_t *charge;             /* [n] cols 79-80 */
} tm_pdb_columns;

int tm_pdb_count_host(const char *text, uint64_t length, uint32_t flags, uint64_t *n_atoms, char header_id[4],
                      int *has_header);
int tm_pdb_parse_host(const char *text, uint64_t length, uint32_t flags, uint64_t capacity, const tm_pdb_columns *out,
                      uint64_t *bad_begin, uint64_t *bad_length);
/* The same for many files at once, on a team of host threads (the loop of load_molecules, jess_run.py:523-556).
 * count: n_atoms[f], header_ids[4 f .. 4 f + 3], has_header[f].  parse: file f fills rows atom_offsets[f] ..
 * atom_offsets[f + 1] - 1 of the concatenated arrays; on a malformed record *bad_file is the FIRST such file. */
/* File reading for the two calls above and for tm_batch_create_from_pdb, on host threads: sizes first, then every
 * file into its slice [offsets[f], offsets[f + 1]) of one caller-allocated buffer.  *bad_file = first file that
 * could not be read (-1: none). */
int tm_files_sizes(uint32_t n_files, const char *const *paths, uint64_t *sizes, int64_t *bad_file);
int tm_files_read(uint32_t n_files, const char *const *paths, const uint64_t *offsets, char *buffer, int64_t *bad_file);
int tm_pdb_count_host_many(uint32_t n_files, const char *const *text, const uint64_t *length, uint32_t flags,
                           uint64_t *n_atoms, char *header_ids, int *has_header);
int tm_pdb_parse_host_many(uint32_t n_files, const char *const *text, const uint64_t *length, uint32_t flags,
                           const uint64_t *atom_offsets, const tm_pdb_columns *out, int64_t *bad_file,
                           uint64_t *bad_begin, uint64_t *bad_length);

#ifdef __cplusplus
}
#endif
#endif /* ENZYMM_AMD_H */
