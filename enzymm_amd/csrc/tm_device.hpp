// Device-side data views shared by the kernels and the host API of libenzymm_amd.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "enzymm_amd.h"

namespace tmx {

constexpr int kMaxK = TM_MAX_TEMPLATE_ATOMS;        // template atoms per template
constexpr int kMaxG = TM_MAX_TEMPLATE_GROUPS;       // template residues
constexpr int kMaxPairs = kMaxK * (kMaxK - 1) / 2;  // 276 pair constraints
constexpr int kWave = 64;
// (experiment knobs: make variant NAME=occ EXTRA="-DTM_SEARCH_WAVES=12 -DTM_SEARCH_MIN_WAVES=6", tests/gpu_occ_exp.sh)
#ifndef TM_SEARCH_WAVES
#define TM_SEARCH_WAVES 16
#endif
constexpr int kSearchWaves = TM_SEARCH_WAVES;  // waves per search workgroup (one structure copy in LDS serves all)
constexpr int kSearchBlock = kWave * kSearchWaves;
constexpr int kPairQueue = 128;   // pair-table survivors buffered per wave
constexpr int kTripleQueue = 128;  // triple survivors waiting for their subtrees

// Pair tables (see k_tab_* in tm_kernels.hip): for an ordered pair of lead kinds (A, B) and a distance limit, every
// pair (i, j) of list positions -- i in the structure's kind list A, j in list B -- whose atoms lie in different
// residues and within the limit, stored row by row (CSR over i) as {i | j << 16, squared fp32 distance}.
struct TableDesc {
    uint16_t li_a, li_b;  // lead-kind indices
    float dmax2;          // squared distance limit (already holds the prefilter slack); FINITE -- the widest class uses 3.0e38f:
                          // tab_sweep folds its lane mask into the comparison as (counts ? d2 : +inf) <= dmax2
    // Tables of the same two kind lists differ in their distance limit only: the one with the largest limit (the group's
    // LEADER, src == its own index) is built by testing the lists' cross product; the others hold a subset of its
    // entries in the same order and are DERIVED from it by filtering (k_tab_build), which costs its entries, not the
    // cross product.  next: the next table derived from the same leader (0xffff: none); a leader's next starts the chain.
    uint16_t src, next;
};
static_assert(sizeof(TableDesc) == 12, "TableDesc layout");
constexpr int kTableClasses = 5;
constexpr double kTableClassLimit[kTableClasses] = {8.0, 12.0, 16.0, 24.0, 1e30};  // upper window edge (A) per class

// fp32 prefilter slack (Angstrom) added to every pair threshold; anything the slack lets through is re-checked in
// fp64 before it can become a candidate.  What it has to cover is the rounding of the CENTRED fp32 coordinates: half
// an ulp per coordinate, 2^-24 |c| -- for a structure whose atoms lie within kPrefilterExtent of its centroid at most
// sqrt(3) x 2 x 2.44e-4 = 8.5e-4 A on a distance (a 300-residue protein: 1e-5 A), plus 2e-5 A for the squared-distance
// arithmetic and the rounding of the window edges.  A batch with a structure beyond that extent (the upload measures
// it, tm_batch::extent) is searched with the slack doubled per doubling of the extent: prefilter_slack().
constexpr double kPrefilterSlack = 2e-3;
constexpr double kPrefilterExtent = 4096.0;
inline double prefilter_slack(double extent) {
    double slack = kPrefilterSlack;
    for (double reach = kPrefilterExtent; reach < extent; reach *= 2.0) slack *= 2.0;
    return slack;
}

// Atoms a structure may have for the 16-bit local atom indices of the tuned paths (kind lists, residue tables, queue
// entries, pages, candidates).  A WIDE structure -- more than that, up to TM_MAX_STRUCTURE_ATOMS -- keeps the upper bits of
// its atom indices in byte arrays beside the 16-bit ones (kl_pool_hi, res_start_hi, cands_hi; allocated only for batches
// that hold such a structure) and is searched by the template-at-a-time launch alone, with its records in global memory;
// the block search and k_deep pass it over (round 5: it was refused before).
constexpr uint32_t kNarrowAtoms = TM_NARROW_STRUCTURE_ATOMS;

// best_match: candidates within this of the (global) minimum RMSD are tied with it; the smallest atom-index tuple wins.
constexpr double kTieEps = 1e-9;

// bits of the packed per-atom meta word (float4.w of a target atom record)
//   [15:0] residue ordinal   [21:16] atom-name id   [26:22] residue-type id
__host__ __device__ inline uint32_t meta_pack(uint32_t res, uint32_t name, uint32_t rtype) {
    return (res & 0xffffu) | ((name & 63u) << 16) | ((rtype & 31u) << 22);
}
__host__ __device__ inline uint32_t meta_res(uint32_t m) { return m & 0xffffu; }
__host__ __device__ inline uint32_t meta_name(uint32_t m) { return (m >> 16) & 63u; }
__host__ __device__ inline uint32_t meta_rtype(uint32_t m) { return (m >> 22) & 31u; }

// Template library in HBM.  Per-atom arrays are in SEARCH order (see plan_template()).
struct LibView {
    uint32_t n_templates, n_kinds, n_lead;
    const uint32_t *t_off;   // [n_templates+1] atom offsets
    const double *a_xyz;     // [n_atoms*3]
    const double *a_w;       // [n_atoms] distance weight
    const uint16_t *a_kind;  // [n_atoms] -> kind table
    const uint8_t *a_group;  // [n_atoms] template residue index (search-order numbering)
    const uint8_t *a_ref;    // [n_atoms] search position of the residue's lead atom (== own position for a lead)
    const uint8_t *a_orig;   // [n_atoms] position of this atom in TEMPLATE (file) order
    const uint8_t *a_chain;  // [n_atoms] template chain ordinal
    // kind = distinct (any_residue, res_mask, name_mask) triple
    const uint32_t *k_resmask;
    const uint64_t *k_namemask;
    const uint8_t *k_any;
    const int32_t *k_lead;       // index among lead kinds, or -1
    const uint32_t *lead_kind;   // [n_lead] kind id of each lead kind
    // optional residue-orientation table (template order), NULL when absent
    const uint32_t *r_off;       // [n_templates+1]
    const uint8_t *r_first, *r_second;
    const double *r_vec;         // [n_residues*3]
};

// Target structures in HBM.  Atoms of a structure are stored residue-contiguous.
struct BatchView {
    uint32_t n_structs;
    const uint32_t *s_off;      // [n_structs+1] atom offsets
    const uint32_t *s_roff;     // [n_structs+1] offsets into res_start
    const double *q_xyz;        // [n_atoms*3] original coordinates (fp64 re-check, superposition)
    const float4 *q_rec;        // [n_atoms] centred fp32 xyz + meta word
    const uint8_t *q_chain;     // [n_atoms]
    const uint32_t *q_orig;     // [n_atoms] file-order index of this atom within its structure
    const uint16_t *res_start;  // per structure: n_res+1 local atom indices
    // candidate index for one library: for each (structure, lead kind) the admissible atoms
    const uint32_t *kl_off;   // [n_structs*(n_lead+1)] offsets into kl_pool
    const uint16_t *kl_pool;  // local atom indices, ascending (low 16 bits)
    const uint8_t *kl_pool_hi;   // bits 16-23 of the same entries; NULL: the batch has no wide structure
    const uint8_t *res_start_hi; // bits 16-23 of res_start, same indexing; NULL likewise
    // pair tables for one staged selection: row r of table t of structure s is entries
    // [tab_rows[tab_dir[s * n_tables + t] + r], tab_rows[... + r + 1]) of tab_ent
    uint32_t n_tables;
    const uint32_t *tab_dir;   // [n_structs * n_tables] first row slot of the table
    const uint2 *tab_seg;      // [n_structs * n_tables] (first entry, one past the last entry) of the table
    const uint32_t *tab_ok;    // [n_structs] 1: the structure's tables are complete (bit 0: row slots fit, bit 1: entries ran out)
    const uint4 *tab_ref;      // [n_structs * n_sel] per (structure, selected template): see k_tab_refs
    const uint32_t *tab_rows;  // row offsets (absolute entry indices); a table with n rows owns n + 1 slots
    const uint2 *tab_ent;      // {position in list A | position in list B << 16, squared distance (fp32 bits)}
};

// One selected template, flattened for the search kernel (written by k_stage_templates).
struct __align__(16) Staged {
    unsigned long long namemask[kMaxK];
    uint32_t resmask[kMaxK];
    uint16_t lead_li[kMaxK];  // lead levels: index among lead kinds; 0xffff otherwise
    uint32_t packed[kMaxK];   // group | ref << 8 | any << 13 | orig << 14 | tchain << 24 (see lvl_* in tm_kernels.hip)
    uint8_t orig[kMaxK];      // search position -> template (file) order, also part of `packed`
    // one word, one load per search: k | n_groups << 8 | triple_key << 16 (triple_key 0: rows of the triple table
    // are positions of lead 0, 1: of lead 1)
    uint32_t hdr;
    uint16_t pair_table;    // table (lead 0, lead 1); 0xffff: fewer than two residues
    uint16_t triple_table;  // table (lead 0 or 1, lead 2); 0xffff: fewer than three residues
    __host__ __device__ uint32_t k() const { return hdr & 0xffu; }
};

// What the table-driven stages of the search (the lead atoms of the first three template residues) need of a staged
// template, 32 bytes: k_search reads these 64 templates at a time, one per lane (search_block in tm_kernels.hip).
struct __align__(16) BlockHdr {
    float2 w10;      // fp32 window (lo^2, hi^2) of the pair (lead 1, lead 0)
    float2 wkey;     // ... of (lead 2, key lead): the stored distance of a triple-table entry is tested against it
    float2 wother;   // ... of (lead 2, the other of leads 0 and 1): computed from the atom records
    uint16_t li[3];  // lead-kind indices of leads 0, 1, 2 (0xffff: the template has fewer residues)
    uint16_t flags;  // bit 0: rows of the triple table are positions of lead 1 (else lead 0); bits 1-3: template chains of
                     // (1,0), (2,0), (2,1) equal; bits 4-7: residues; bits 8-12: ResCheck records (= LookHdr::n_rescheck)
};
static_assert(sizeof(BlockHdr) == 32, "BlockHdr layout");

// What the quick look at search level 3 needs of a staged template (48 bytes; k_search, stage D1): the windows of the
// fourth search position against the three placed leads and where its candidates come from -- the kind list of a
// fourth lead, or the atoms of the target residue one of the three leads was assigned to.
struct __align__(16) LookHdr {
    float2 w30, w31, w32;
    uint32_t nm_lo, nm_hi;  // residue-type level 3: admissible atom names ...
    uint32_t resmask;       // ... and residue types
    uint16_t li3;           // lead-kind index when level 3 is the lead of a fourth residue, else 0xffff
    uint8_t ref;            // else: which of leads 0, 1, 2 shares its residue
    uint8_t flags;          // bit 0: any residue type; bit 1: the template has no level 3 (three atoms)
    uint32_t n_rescheck;    // ResCheck records of the template (stage D0)
    uint32_t pad;
};
static_assert(sizeof(LookHdr) == 48, "LookHdr layout");

// Stage D0 of the block search: what it takes to ask, for a triple of placed leads, whether ONE OTHER atom of one of their
// three residues can exist at all -- a search level that is not a lead and belongs to template residue 0, 1 or 2: the
// names and residue types it admits and its windows against the three leads.  Such a level depends on nothing but the
// triple, so a triple for which any of them has no candidate cannot be completed, whatever the other leads will be.
// One record per such level, kResChecks slots per staged template.
struct __align__(16) ResCheck {
    uint32_t nm_lo, nm_hi;  // admissible atom names
    uint32_t resmask;       // ... and residue types
    uint32_t ref_any;       // which of leads 0, 1, 2 shares its residue | any residue type << 8
    float2 w0, w1, w2;      // fp32 windows (lo^2, hi^2) against leads 0, 1, 2
    uint32_t pad[2];
};
static_assert(sizeof(ResCheck) == 48, "ResCheck layout");
constexpr int kResChecks = kMaxK - 3;

// A triple (lead atoms of the first three template residues) that survived the tables and the look at level 3, waiting
// for the depth-first search below it (k_deep).  k_search writes them in pages of kPageTriples; a structure's pages
// are listed in its share of the page directory.
struct DeepEntry {
    uint32_t a01;  // target atoms of leads 0 and 1 (device order), 16 bits each
    uint32_t c;    // ... of lead 2
    uint32_t ti;   // staged template
};
constexpr uint32_t kPageTriples = 64;
#ifndef TM_DEEP_RUN
#define TM_DEEP_RUN 2
#endif
// triples a wave of k_deep claims at a time (a power of two).  Measured with the run list and two claims in flight: 2 beats
// 4 (bench batch 2.36-2.39 vs 2.42 ms, ten 25 k-atom complexes at loose cutoffs 28.5 vs 33 ms of k_deep: the last runs of a
// structure are what its workgroups wait for) and 8 loses 10 %.
constexpr uint32_t kDeepRun = TM_DEEP_RUN;

struct QueryView {
    uint32_t n_sel;
    const uint32_t *sel;        // [n_sel] template indices
    const double *rmsd_thr;     // [n_sel]
    const double *cutoff;       // [n_sel]
    const double *maxdyn;       // [n_sel]
    const Staged *staged;       // [n_sel]
    const BlockHdr *hdr;        // [n_sel]
    const LookHdr *look;        // [n_sel]
    const ResCheck *rescheck;   // [n_sel * kResChecks]
    const float2 *bounds;       // [n_sel*kMaxPairs] (lo^2, hi^2) fp32 prefilter bounds, search order
    const double2 *exact;       // [n_sel*kMaxPairs] fp64 (template distance, threshold) of every pair, search order: the re-check of a complete assignment
    const uint32_t *order;      // [n_sel] every per-selection array above is sorted by descending estimated cost;
                                // order[pos] = the caller's selection index (reported in tm_hit / candidate counts)
    uint32_t n_tables;
    const TableDesc *tables;    // [n_tables] pair tables this selection needs
    unsigned long long *cost;   // [n_sel] or NULL: shader-clock ticks spent per staged template (calibration query)
    uint32_t max_candidates;
    uint32_t flags;
    // optional logistic filter
    const int32_t *model_set;       // [n_sel] or NULL
    const uint32_t *model_offsets;  // [n_sets+1]
    const uint32_t *model_votes;    // [n_sets]
    const double *models;           // [n_models*4]
};

// A complete, fp64-verified assignment waiting for superposition.
struct __align__(16) Cand {
    uint32_t s, ti;
    uint32_t prev;               // previous candidate of the same (s, ti) pair, 0xffffffff = none
    uint32_t tie_next;           // best_match: next candidate (+ 1; 0: none) of the pair's list of candidates tied with its minimum RMSD
    uint16_t atoms[kMaxK];       // target atom (device order) per template atom, TEMPLATE order
};
static_assert(sizeof(Cand) == 64, "Cand layout");

// A (structure, template) pair that produced at least one candidate (or was cut short).  Its candidate count and the
// head of its candidate chain live in OutView::pair_count / pair_last: candidates of one pair are found by any number
// of waves (k_deep shares a structure's triples among all waves working on it), so both are updated atomically.
struct PairEntry {
    uint32_t s, ti;
};

struct OutView {
    Cand *cands;
    uint8_t *cands_hi;       // [cand_capacity * kMaxK] bits 16-23 of Cand::atoms; NULL: the batch has no wide structure
    double *cand_rmsd;
    double *cand_fit;        // [cand_fit_capacity * 12] rotation and translation of the candidates within their threshold
    unsigned long long cand_fit_capacity;  // (k_superpose keeps them for k_select; a candidate beyond is superposed again there)
    unsigned long long *n_cands;
    unsigned long long cand_capacity;
    PairEntry *pairs;
    unsigned long long *n_pairs;
    unsigned long long pair_capacity;
    tm_hit *hits;
    unsigned long long *n_hits;
    unsigned long long hit_capacity;
    uint32_t *overflow;      // bit 0: candidate buffer, bit 1: pair table, bit 2: hit table, bit 5: deep pages / directory
    // per (structure, CALLER's selection index): verified candidates | cut short << 31, and the last candidate of the
    // pair's chain + 1 (0: none).  Zeroed per query; pair_count doubles as the TM_FLAG_COUNT_CANDIDATES table.
    uint32_t *pair_count;    // [n_structs * n_sel]
    uint32_t *pair_last;     // [n_structs * n_sel]
    // best_match, per listed pair (initialised by whoever lists the pair, not per query): the minimum RMSD within the
    // threshold among its candidates (bits of a non-negative double: they order like integers; k_superpose), and the
    // list of the candidates within kTieEps of it (k_ties) -- k_select looks at those one or two candidates instead of
    // walking a chain of up to max_candidates
    unsigned long long *pair_min;  // [n_structs * n_sel]
    uint32_t *pair_tie;            // [n_structs * n_sel] last linked tied candidate + 1 (0: none)
    // triples waiting for k_deep
    DeepEntry *pages;                 // [page_capacity * kPageTriples]
    unsigned long long *n_pages;      // pages handed out
    unsigned long long page_capacity;
    uint2 *dir;                       // page directory: {page, triples in it}; structure s owns [dir_off[s], dir_off[s + 1])
    const uint32_t *dir_off;          // [n_structs + 1]
    uint32_t *dir_n;                  // [n_structs] pages listed (may exceed the share: overflow)
    uint32_t *dir_triples;            // [n_structs] triples in them
    uint32_t *dir_head;               // [n_structs] k_deep: next run of kDeepRun triples to claim, counted over the listed pages
};

}  // namespace tmx
