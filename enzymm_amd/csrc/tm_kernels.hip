// gfx950 kernels of the template matcher.
//
//   k_stage_templates  per selected template: search-order tables + fp32 (lo^2, hi^2) pair bounds
//   k_kind_lists       per structure: for every "lead kind" the ascending list of admissible atoms
//   k_search           wave-per-(structure, template) backtracking search; every complete assignment
//                      is re-checked in fp64 and appended to the candidate buffer
//   k_superpose        lane-per-candidate quaternion superposition (fp64) -> RMSD
//   k_select           lane-per-(structure, template) pair: best candidate -> hit table
//
// The search restates what Jess does for `Jess.query(..., best_match=True)`
// (/root/reference/enzymm/jess_run.py:803-811, semantics in SURVEY.md 8c-2): enumerate every
// assignment template atom -> target atom that is type-admissible, residue-consistent and
// satisfies all pair-distance constraints, superpose each, keep the minimum-RMSD one.
// The enumeration ORDER is this kernel's own (lead atoms of all template residues first); the
// SET of complete assignments does not depend on it.
#include <cstddef>
#include <type_traits>

#include "tm_device.hpp"

namespace tmx {

#ifdef TM_STATS
// diagnostic build only: [0..23] chunk evaluations per level, [24..47] survivors per level,
// [48] complete assignments, [49] searches, [50] pair tests (j-loop trips), [51] fp64 rejects,
// [52] pair-stage passes, [53] pair survivors, [67] triple-stage passes, [68] triple survivors, [64] look flushes,
// [66] triples looked at, [65] triples paged out, [60] / [62] triples whose level 3 is a lead / a residue atom, [61] candidate chunks of the
// former, [63] items of the latter, [55] workgroups, [56] workgroups with their kind lists in LDS,
// [57] kind-list entries of their structures
// [96+b] / [120+b]: distance tests of swept leader / non-leader tables whose cross product has bit length b; [144+b] / [168+b]: their entries
// [192+b]: queue entries whose search took 2^b ticks of 10 ns, [216+b]: their ticks in sum; [240+d]: entries of depth d
// [256+d]: entries handed over at depth d (lanes that wrote one)
__device__ unsigned long long g_stats[320];
#define TM_STAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
// Ablation (diagnostic and timing builds): query flag bit 8 (0x100) drops the pair-scan survivors, bit 9 (0x200)
// skips the search after template staging, bit 10 (0x400) drops the triple-scan survivors.
#else
#define TM_STAT(i, v) do { } while (0)
#endif
#ifdef TM_TIMES
// diagnostic build only (make variant NAME=times EXTRA=-DTM_TIMES, tests/gpu_stage_times.py): shader-clock cycles the
// WAVES of the block search spend per stage, waiting included -- [0] A, [1] B, [2] C, [3] D0, [4] D1 + queueing, [5] the
// depth-first search below queued triples, [6] waiting at the end of a unit, [7] unused, [8] the waves' whole time in
// the template loop of their units, [9] blocks, [10] C calls, [11] look flushes, [12] queue entries searched
__device__ unsigned long long g_times[16];
// (accumulated in the wave's search context C, written out once per unit: an atomic per stage visit took the launch from 1.8
// to 30 ms and measured mostly itself)
#define TM_T0(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define TM_T1(slot, v) (C.tacc[slot] += __builtin_amdgcn_s_memtime() - (v))
#define TM_TN(slot, n) (C.tacc[slot] += (unsigned long long)(n))
#else
#define TM_T0(v) ((void)0)
#define TM_T1(slot, v) ((void)0)
#define TM_TN(slot, n) ((void)0)
#endif
#ifdef TM_ABLATE
// timing build only: when every workgroup of the block search started and ended (10 ns ticks) and where it ran
// (HW_ID: CU, SE; XCC_ID), for tests/gpu_wg_timeline.py
struct WgLog { unsigned long long start, end, first_end, sum_end; uint32_t hw_id, xcc_id; };
constexpr uint32_t kWgLog = 16384;
__device__ WgLog g_wg_log[kWgLog];
__device__ WgLog g_tab_log[kWgLog];  // the same for the workgroups of k_tab_build
__device__ WgLog g_deep_log[kWgLog]; // ... and of k_deep
#endif

// ------------------------------------------------------------------ wave helpers

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
#ifndef TM_D0_INTERLEAVE
#define TM_D0_INTERLEAVE 0  // stage D0's records ordered (ordinal within residue, residue) instead of search order (experiment)
#endif
#ifndef TM_D0_SCAN
#define TM_D0_SCAN 1  // stage D0: names scanned in a wave-uniform loop, candidates tested in lock step (0: the per-lane walk of rounds 4-5)
#endif
#ifndef TM_D0_MAX_NAMES
#define TM_D0_MAX_NAMES 64  // stage D0 asks about levels that admit at most this many atom names (experiments: 2)
#endif
// -DTM_MARKS: comments in the ISA around the block search's stages (static instruction counts per stage; a diagnostic
// compile -- the markers pin the schedule -- never shipped)
#ifdef TM_MARKS
#define TM_MARK(name) asm volatile("; TM_MARK " name)
#else
#define TM_MARK(name) ((void)0)
#endif
// wave-wide vote straight into an SGPR pair (HIP's __ballot goes through an integer compare)
__device__ __forceinline__ unsigned long long ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
// number of set bits of `m` below this lane (v_mbcnt_lo + v_mbcnt_hi)
__device__ __forceinline__ uint32_t rank_in(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// Inclusive scans over the 64 lanes in six DPP steps (no LDS round trips): row_shr 1, 2, 4, 8 scan each row of 16
// lanes -- a lane whose source falls outside its row keeps the identity --, row_bcast:15 adds a row's total to the
// next row (rows 1 and 3), row_bcast:31 adds the total of the first 32 lanes to rows 2 and 3.
#define TM_DPP(x, ctrl, rows) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), ctrl, rows, 0xf, false)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {
    x += TM_DPP(x, 0x111, 0xf);
    x += TM_DPP(x, 0x112, 0xf);
    x += TM_DPP(x, 0x114, 0xf);
    x += TM_DPP(x, 0x118, 0xf);
    x += TM_DPP(x, 0x142, 0xa);
    x += TM_DPP(x, 0x143, 0xc);
    return x;
}
__device__ __forceinline__ uint32_t wave_scan_max(uint32_t x) {  // identity 0
    x = max(x, TM_DPP(x, 0x111, 0xf));
    x = max(x, TM_DPP(x, 0x112, 0xf));
    x = max(x, TM_DPP(x, 0x114, 0xf));
    x = max(x, TM_DPP(x, 0x118, 0xf));
    x = max(x, TM_DPP(x, 0x142, 0xa));
    x = max(x, TM_DPP(x, 0x143, 0xc));
    return x;
}
#undef TM_DPP
// value of `v` in lane `src` (any lane may ask for any lane; every lane of the wave must execute this)
__device__ __forceinline__ uint32_t lane_get(uint32_t v, uint32_t src) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)v);
}

__device__ __forceinline__ float lane_getf(float v, uint32_t src) { return __uint_as_float(lane_get(__float_as_uint(v), src)); }

// Orders this wave's LDS traffic: lanes exchange data through LDS without a workgroup barrier.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__host__ __device__ __forceinline__ int tri(int i) { return i * (i - 1) / 2; }  // pair(i, j<i) = tri(i) + j

__device__ __forceinline__ void pair_of(int p, int &i, int &j) {
    i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
    while (tri(i) > p) --i;
    while (tri(i + 1) <= p) ++i;
    j = p - tri(i);
}

// Per-level template constants in one word: template residue (search-order numbering, 0xff = unused level), search
// position of the residue's lead, "any residue type", position in template (file) order, template chain ordinal.
__host__ __device__ __forceinline__ uint32_t lvl_pack(uint32_t group, uint32_t ref, uint32_t any, uint32_t orig, uint32_t tchain) {
    return (group & 0xffu) | ((ref & 31u) << 8) | ((any & 1u) << 13) | ((orig & 31u) << 14) | ((tchain & 0xffu) << 24);
}
__host__ __device__ __forceinline__ uint32_t lvl_group(uint32_t p) { return p & 0xffu; }
__host__ __device__ __forceinline__ uint32_t lvl_ref(uint32_t p) { return (p >> 8) & 31u; }
__host__ __device__ __forceinline__ uint32_t lvl_any(uint32_t p) { return (p >> 13) & 1u; }
__host__ __device__ __forceinline__ uint32_t lvl_orig(uint32_t p) { return (p >> 14) & 31u; }
__host__ __device__ __forceinline__ uint32_t lvl_tchain(uint32_t p) { return p >> 24; }

// ------------------------------------------------------------------ k_stage_templates

__global__ __launch_bounds__(64) void k_stage_templates(LibView L, uint32_t n_sel, const uint32_t *__restrict__ sel,
                                                         const double *__restrict__ cutoff,
                                                         const double *__restrict__ maxdyn,
                                                         const uint16_t *__restrict__ pair_table,
                                                         const uint16_t *__restrict__ triple_table,
                                                         const uint8_t *__restrict__ triple_key,
                                                         Staged *__restrict__ staged, float2 *__restrict__ bounds,
                                                         BlockHdr *__restrict__ hdr, LookHdr *__restrict__ look,
                                                         uint16_t *__restrict__ lead8, ResCheck *__restrict__ rescheck,
                                                         double2 *__restrict__ exact, double slack) {
    const uint32_t ti = blockIdx.x;
    if (ti >= n_sel) return;
    const uint32_t t = sel[ti];
    const uint32_t off = L.t_off[t];
    const int k = (int)(L.t_off[t + 1] - off);
    const int npairs = tri(k);
    const double cut = cutoff[ti], mx = maxdyn[ti];
    Staged &S = staged[ti];
    const int lane = threadIdx.x;
    if (lane < kMaxK) {
        if (lane < k) {
            const uint32_t kind = L.a_kind[off + lane];
            const uint8_t ref = L.a_ref[off + lane];
            S.namemask[lane] = L.k_namemask[kind];
            S.resmask[lane] = L.k_resmask[kind];
            S.lead_li[lane] = ref == lane ? (uint16_t)L.k_lead[kind] : (uint16_t)0xffffu;
            S.orig[lane] = L.a_orig[off + lane];
            S.packed[lane] = lvl_pack(L.a_group[off + lane], ref, L.k_any[kind], L.a_orig[off + lane], L.a_chain[off + lane]);
        } else {
            S.namemask[lane] = 0; S.resmask[lane] = 0; S.lead_li[lane] = 0xffffu;
            S.orig[lane] = 0;
            S.packed[lane] = lvl_pack(0xff, 0, 0, 0, 0);
        }
    }
    if (lane == 0) {
        uint8_t g = 0;
        for (int i = 0; i < k; ++i) g = max(g, (uint8_t)(L.a_group[off + i] + 1));
        S.hdr = (uint32_t)k | ((uint32_t)g << 8) | ((uint32_t)triple_key[ti] << 16);
        S.pair_table = pair_table[ti];
        S.triple_table = triple_table[ti];
    }
    auto window = [&](int i, int j) {  // fp32 (lo^2, hi^2) of the pair constraint (i, j), slack included
        const double *a = L.a_xyz + 3 * (size_t)(off + i), *b = L.a_xyz + 3 * (size_t)(off + j);
        const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
        const double dt = sqrt(dx * dx + dy * dy + dz * dz);
        double thr = cut + L.a_w[off + i] + L.a_w[off + j];
        thr = (thr < mx ? thr : mx) + slack;
        const double lo = dt - thr, hi = dt + thr;
        const float lo2 = lo > 0.0 ? (float)(lo * lo) * (1.0f - 1e-6f) : 0.0f;
        const float hi2 = (float)(hi * hi) * (1.0f + 1e-6f);
        return make_float2(lo2, hi2);
    };
    for (int p = lane; p < npairs; p += 64) {
        int i, j;
        pair_of(p, i, j);
        bounds[(size_t)ti * kMaxPairs + p] = window(i, j);
        // what the fp64 re-check of a complete assignment compares against (on_complete), in the very arithmetic it used to
        // redo per completion: template distance and threshold of the pair
        const double *a = L.a_xyz + 3 * (size_t)(off + i), *b = L.a_xyz + 3 * (size_t)(off + j);
        const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
        const double dt = sqrt(dx * dx + dy * dy + dz * dz);
        double thr = cut + L.a_w[off + i] + L.a_w[off + j];
        thr = thr < mx ? thr : mx;
        exact[(size_t)ti * kMaxPairs + p] = make_double2(dt, thr);
    }
    // stage D0's records: the levels that are not leads and belong to template residues 0, 1, 2, in search order
    unsigned long long rc_mask = 0ull;
    {
        bool qual = lane < k && L.a_ref[off + min(lane, k - 1)] != lane && L.a_ref[off + min(lane, k - 1)] < 3 && k > 3;
#if TM_D0_MAX_NAMES < 64
        // (experiment: stage D0 asks only about levels that admit few atom names -- a level that matches by element admits a
        // dozen names, several atoms of a residue pass its name test and each costs the three windows; D0 is a filter of
        // necessary conditions, a level it does not ask about is left to the search)
        if (qual) qual = __popcll(L.k_namemask[L.a_kind[off + lane]]) <= TM_D0_MAX_NAMES;
#endif
        rc_mask = ballot64(qual);
        if (qual) {
            const uint32_t kind = L.a_kind[off + lane];
            const uint64_t nm = L.k_namemask[kind];
            ResCheck r = {};
            r.nm_lo = (uint32_t)nm; r.nm_hi = (uint32_t)(nm >> 32);
            r.resmask = L.k_any[kind] ? 0xffffffffu : L.k_resmask[kind];  // (any residue type: every type's bit, one test in D0's scan)
            r.ref_any = (uint32_t)L.a_ref[off + lane] | (L.k_any[kind] ? 256u : 0u);
            r.w0 = window(lane, 0); r.w1 = window(lane, 1); r.w2 = window(lane, 2);
            // a template whose first three residues hold more than kResChecks such levels (one residue of 23 or 24 atoms)
            // keeps the first kResChecks: D0 is a necessary-condition filter, a level it does not ask about is left to the search
#if TM_D0_INTERLEAVE
            // The records in the order (ordinal within its residue, residue): D0's first round then asks about one atom of each
            // of two RESIDUES instead of two atoms of the first -- whether a second atom of a residue fits is nearly decided by
            // the first, so a round over two residues ends more triples.  (Which levels a round asks about changes, the set of
            // levels does not: the same triples pass D0.)
            uint32_t slot = 0;
            {
                const uint32_t my_ref = L.a_ref[off + lane];
                unsigned long long of_ref[3];
                for (uint32_t r = 0; r < 3u; ++r) of_ref[r] = ballot64(qual && L.a_ref[off + min(lane, k - 1)] == (int)r);
                const uint32_t j = (uint32_t)__popcll(of_ref[my_ref] & ((1ull << lane) - 1ull));  // my ordinal within my residue
                for (uint32_t r = 0; r < 3u; ++r) {
                    const uint32_t cnt = (uint32_t)__popcll(of_ref[r]);
                    slot += min(cnt, j) + (r < my_ref && cnt > j ? 1u : 0u);
                }
            }
#else
            const uint32_t slot = rank_in(rc_mask);
#endif
            if (slot < (uint32_t)kResChecks) rescheck[(size_t)ti * kResChecks + slot] = r;
        }
    }
    if (lane == 0) {
        // the compact record of the table-driven stages (leads 0, 1, 2 are search positions 0, 1, 2)
        int g = 0;
        for (int i = 0; i < k; ++i) g = max(g, (int)L.a_group[off + i] + 1);
        BlockHdr h;
        const uint32_t key = triple_key[ti];
        h.w10 = g >= 2 ? window(1, 0) : make_float2(0.f, 0.f);
        h.wkey = g >= 3 ? window(2, (int)key) : make_float2(0.f, 0.f);
        h.wother = g >= 3 ? window(2, 1 - (int)key) : make_float2(0.f, 0.f);
        for (int i = 0; i < 3; ++i) h.li[i] = i < g ? (uint16_t)L.k_lead[L.a_kind[off + i]] : (uint16_t)0xffffu;
        const auto tc = [&](int i, int j) { return i < k && j < k && L.a_chain[off + i] == L.a_chain[off + j] ? 1u : 0u; };
        // (bits 8-12: the number of stage D0's records -- the block search has it in a register with the rest of the header
        // when a triple reaches D0, instead of asking the look record for it: one trip to L2 less at the head of the stage)
        h.flags = (uint16_t)(key | (tc(1, 0) << 1) | (tc(2, 0) << 2) | (tc(2, 1) << 3) | ((uint32_t)g << 4) |
                             (min((uint32_t)__popcll(rc_mask), (uint32_t)kResChecks) << 8));
        hdr[ti] = h;
        // ... and of the look at level 3 (search position 3: the lead of a fourth residue, or another atom of one of
        // the first three)
        LookHdr q = {};
        if (k > 3) {
            const uint32_t kind = L.a_kind[off + 3];
            const uint32_t ref = L.a_ref[off + 3];
            q.w30 = window(3, 0); q.w31 = window(3, 1); q.w32 = window(3, 2);
            q.nm_lo = (uint32_t)L.k_namemask[kind]; q.nm_hi = (uint32_t)(L.k_namemask[kind] >> 32);
            q.resmask = L.k_resmask[kind];
            q.li3 = ref == 3u ? (uint16_t)L.k_lead[kind] : (uint16_t)0xffffu;
            q.ref = (uint8_t)(ref == 3u ? 0u : ref);
            q.flags = L.k_any[kind] ? 1u : 0u;
        } else {
            q.li3 = 0xffffu;
            q.flags = 2u;
        }
        q.n_rescheck = min((uint32_t)__popcll(rc_mask), (uint32_t)kResChecks);
        look[ti] = q;
        for (int i = 0; i < kMaxG; ++i) lead8[(size_t)ti * kMaxG + i] = i < g ? (uint16_t)L.k_lead[L.a_kind[off + i]] : (uint16_t)0xffffu;
    }
}

// ------------------------------------------------------------------ k_kind_lists

// The candidate index: for every lead kind the ascending list of admissible atoms of every structure.
// One workgroup per structure of at most kKindSeg atoms (kKindWhole); a longer structure (a large complex) is cut into
// segments of kKindSeg atoms that are counted (kKindCount), given their places (k_kind_seg_offsets) and written
// (kKindWrite) by one workgroup each -- one workgroup per structure took 0.77 ms per 25 k-atom complex, on one CU.
//   1. the atoms' name and residue-type ids go to LDS (2 bytes per atom) and into a (name x residue type) histogram: the
//      length of every kind's list is a sum over the histogram (one thread per kind), not a scan of the atoms;
//   2. the atoms are bucketed by atom-name id, ascending inside a bucket: per 64-atom block one vote per distinct name
//      gives every atom its rank among the block's atoms of that name and the block's count, a scan over the blocks per
//      name and one over the names give the bucket positions (before: every wave scanned ALL atoms once per name);
//   3. a lead kind that accepts a single name -- nearly all of them -- scans that bucket for its residue types, a kind
//      that matches by element (ten names) scans the segment; the waves take kinds one at a time from a counter.
// Nothing in it waits on global memory inside a loop.
// Dynamic LDS (kind_lds_bytes): [n_lead + 1] counts/offsets, [n_lead] kind constants, the histogram, the per-(name, block)
// counts, then per atom: name | residue type << 8 (u16), the buckets (u16), the rank in its block (u8).
constexpr uint32_t kKindSeg = 4096;  // atoms a workgroup of the index build handles (64 blocks of 64: one lane per block)
struct KindSeg { uint32_t s, lo; };          // a segment of a long structure: structure, first atom
struct KindLong { uint32_t s, first, n; };   // a long structure: its segments are [first, first + n) of the segment list
enum { kKindWhole = 0, kKindCount = 1, kKindWrite = 2 };
__host__ __device__ inline size_t kind_lds_bytes(uint32_t n_lead, uint32_t cap_atoms) {
    return ((((size_t)n_lead + 1) * 4 + 15) & ~(size_t)15) + (size_t)n_lead * 16 + 2048 * 4 + 4096 * 2 + (size_t)cap_atoms * 5 + 16;
}

template <int kMode>
__global__ __launch_bounds__(512) void k_kind_lists(LibView L, BatchView B, const KindSeg *__restrict__ segs,
                                                     uint32_t *__restrict__ seg_cnt, uint32_t *__restrict__ kl_off,
                                                     uint16_t *__restrict__ kl_pool, uint8_t *__restrict__ kl_pool_hi,
                                                     unsigned long long *__restrict__ cursor,
                                                     unsigned long long capacity, uint32_t *__restrict__ overflow,
                                                     uint32_t cap_atoms) {
    extern __shared__ __align__(16) unsigned char kind_smem[];
    const uint32_t nl = L.n_lead;
    uint32_t *s_cnt = reinterpret_cast<uint32_t *>(kind_smem);                                          // [nl + 1]
    uint4 *s_kc = reinterpret_cast<uint4 *>(kind_smem + ((((size_t)nl + 1) * 4 + 15) & ~(size_t)15));   // [nl] {resmask, names lo, hi, any}
    uint32_t *s_hist = reinterpret_cast<uint32_t *>(s_kc + nl);                                         // [64 names][32 residue types]
    uint16_t *s_bc = reinterpret_cast<uint16_t *>(s_hist + 2048);                                       // [64 names][64 blocks]
    uint16_t *s_meta = s_bc + 4096;                                                                     // [m]
    uint16_t *s_bucket = s_meta + cap_atoms;                                                            // [m]
    uint8_t *s_rank = reinterpret_cast<uint8_t *>(s_bucket + cap_atoms);                                // [m]
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_noff[65];  // bucket of name id i: [s_noff[i], s_noff[i + 1])
    __shared__ uint32_t s_next;
    uint32_t s, lo = 0;
    if (kMode == kKindWhole) s = blockIdx.x;
    else { s = segs[blockIdx.x].s; lo = segs[blockIdx.x].lo; }
    const uint32_t a0 = B.s_off[s], n = B.s_off[s + 1] - a0;
    if (kMode == kKindWhole && n > kKindSeg) return;  // a long structure: the segment launches
    const uint32_t m = kMode == kKindWhole ? n : min(kKindSeg, n - lo);
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const float4 *rec = B.q_rec + a0 + lo;
    const uint32_t nblocks = (m + 63u) >> 6;

    for (uint32_t i = threadIdx.x; i < 2048u; i += blockDim.x) {
        s_hist[i] = 0u;
        reinterpret_cast<uint32_t *>(s_bc)[i] = 0u;
    }
    if (threadIdx.x == 0) s_next = 0u;
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        const uint32_t mw = __float_as_uint(rec[i].w);
        s_meta[i] = (uint16_t)(meta_name(mw) | (meta_rtype(mw) << 8));
        atomicAdd(&s_hist[meta_name(mw) * 32u + meta_rtype(mw)], 1u);
    }
    __syncthreads();
    // one thread per lead kind: its constants to LDS, and the length of its list from the histogram
    for (uint32_t li = threadIdx.x; li < nl; li += blockDim.x) {
        const uint32_t kind = L.lead_kind[li];
        const uint64_t nm = L.k_namemask[kind];
        const uint32_t rm = L.k_any[kind] != 0 ? 0xffffffffu : L.k_resmask[kind];
        s_kc[li] = make_uint4(rm, (uint32_t)nm, (uint32_t)(nm >> 32), 0u);
        if (kMode != kKindWrite) {
            uint32_t c = 0;
            for (uint64_t names = nm; names != 0ull; names &= names - 1ull) {
                const uint32_t *row = s_hist + 32u * ((uint32_t)__ffsll((long long)names) - 1u);
                for (uint32_t types = rm; types != 0u; types &= types - 1u) c += row[__ffs((int)types) - 1];
            }
            if (kMode == kKindCount) seg_cnt[(size_t)blockIdx.x * nl + li] = c;
            else s_cnt[li] = c;
        }
    }
    if (kMode == kKindCount) return;

    // ---- the name buckets.  Per block of 64 atoms: one vote per distinct name in it
    for (uint32_t b = wave; b < nblocks; b += nwaves) {
        const uint32_t i = (b << 6) + lane;
        const uint32_t name = i < m ? (uint32_t)s_meta[i] & 0xffu : 0xffu;
        unsigned long long remaining = ballot64(i < m);
        uint32_t rank = 0;
        while (remaining != 0ull) {
            const uint32_t cur = (uint32_t)__builtin_amdgcn_readlane((int)name, __ffsll((long long)remaining) - 1);
            const unsigned long long same = ballot64(name == cur);
            if (name == cur) rank = rank_in(same);
            if (lane == 0) s_bc[(cur << 6) + b] = (uint16_t)__popcll(same);
            remaining &= ~same;
        }
        if (i < m) s_rank[i] = (uint8_t)rank;
    }
    __syncthreads();
    // per name: the atoms of that name before every block (lane = block), and the bucket's length
    for (uint32_t name = wave; name < 64u; name += nwaves) {
        const uint32_t c = s_bc[(name << 6) + lane];
        const uint32_t incl = wave_scan_add(c);
        s_bc[(name << 6) + lane] = (uint16_t)(incl - c);
        if (lane == 63u) s_noff[name + 1] = incl;
    }
    if (threadIdx.x == 0) s_noff[0] = 0u;
    __syncthreads();
    if (wave == 0) s_noff[lane + 1] = wave_scan_add(s_noff[lane + 1]);  // (64 name ids: one per lane)
    if (kMode == kKindWhole && wave == 1) {  // exclusive scan of the kinds' counts, 64 at a time; the structure's slice of the pool
        uint32_t run = 0;
        for (uint32_t l0 = 0; l0 < nl; l0 += 64u) {
            const uint32_t li = l0 + lane;
            const uint32_t c = li < nl ? s_cnt[li] : 0u;
            const uint32_t incl = wave_scan_add(c);
            if (li < nl) s_cnt[li] = run + incl - c;
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) {
            s_cnt[nl] = run;
            unsigned long long base = atomicAdd(cursor, (unsigned long long)run);
            if (base + run > capacity) {
                atomicOr(overflow, 1u);
                base = ~0ull;
            }
            s_base = base;
        }
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
        const uint32_t name = (uint32_t)s_meta[i] & 0xffu;
        s_bucket[s_noff[name] + s_bc[(name << 6) + (i >> 6)] + s_rank[i]] = (uint16_t)i;
    }
    unsigned long long base = 0;
    if (kMode == kKindWhole) {
        base = s_base;
        uint32_t *off_out = kl_off + (size_t)s * (nl + 1);
        for (uint32_t li = threadIdx.x; li <= nl; li += blockDim.x)
            off_out[li] = base == ~0ull ? 0u : (uint32_t)(base + s_cnt[li]);
        if (base == ~0ull) return;
    }
    __syncthreads();

    // ---- the lists: a wave takes the next kind, 64 candidates at a time in ascending order
    for (;;) {
        uint32_t taken = 0;
        if (lane == 0) taken = atomicAdd(&s_next, 1u);
        const uint32_t li = uni(taken);
        if (li >= nl) break;
        const uint4 kc = s_kc[li];
        const uint32_t rm = uni(kc.x);
        const uint64_t nm = ((uint64_t)uni(kc.z) << 32) | uni(kc.y);
        unsigned long long w;
        if (kMode == kKindWhole) w = base + s_cnt[li];
        else {
            const uint32_t at = uni(seg_cnt[(size_t)blockIdx.x * nl + li]);  // (k_kind_seg_offsets: where this segment's part goes)
            if (at == 0xffffffffu) break;                                    // the structure did not fit the pool
            w = at;
        }
        auto emit = [&](bool ok, uint32_t v) {
            const unsigned long long mask = ballot64(ok);
            if (ok) {
                const uint32_t atom = lo + v;
                kl_pool[w + rank_in(mask)] = (uint16_t)atom;
                if (kl_pool_hi) kl_pool_hi[w + rank_in(mask)] = (uint8_t)(atom >> 16);  // (a batch with a wide structure)
            }
            w += __popcll(mask);
        };
        if ((nm & (nm - 1ull)) == 0ull && nm != 0ull) {
            const uint32_t name = (uint32_t)__ffsll((long long)nm) - 1u;
            const uint32_t e1 = s_noff[name + 1];
            for (uint32_t e0 = s_noff[name]; e0 < e1; e0 += 64u) {
                const uint32_t e = e0 + lane;
                uint32_t v = 0;
                bool ok = false;
                if (e < e1) {
                    v = s_bucket[e];
                    ok = (rm >> (s_meta[v] >> 8)) & 1u;
                }
                emit(ok, v);
            }
        } else {
            for (uint32_t b0 = 0; b0 < m; b0 += 64u) {
                const uint32_t i = b0 + lane;
                bool ok = false;
                if (i < m) {
                    const uint32_t mt = s_meta[i];
                    ok = ((nm >> (mt & 0xffu)) & 1ull) && ((rm >> (mt >> 8)) & 1u);
                }
                emit(ok, i);
            }
        }
    }
}
template __global__ void k_kind_lists<kKindWhole>(LibView, BatchView, const KindSeg *, uint32_t *, uint32_t *, uint16_t *, uint8_t *,
                                                  unsigned long long *, unsigned long long, uint32_t *, uint32_t);
template __global__ void k_kind_lists<kKindCount>(LibView, BatchView, const KindSeg *, uint32_t *, uint32_t *, uint16_t *, uint8_t *,
                                                  unsigned long long *, unsigned long long, uint32_t *, uint32_t);
template __global__ void k_kind_lists<kKindWrite>(LibView, BatchView, const KindSeg *, uint32_t *, uint32_t *, uint16_t *, uint8_t *,
                                                  unsigned long long *, unsigned long long, uint32_t *, uint32_t);

// Between the two segment launches, one workgroup per long structure: the segments' counts of every kind become the
// places of their parts -- a running sum over the segments per kind, the exclusive scan over the kinds, the structure's
// slice of the pool -- and the structure's list offsets are written.  Dynamic LDS: [n_lead + 1] words.
__global__ __launch_bounds__(256) void k_kind_seg_offsets(uint32_t nl, const KindLong *__restrict__ longs,
                                                           uint32_t *__restrict__ seg_cnt, uint32_t *__restrict__ kl_off,
                                                           unsigned long long *__restrict__ cursor, unsigned long long capacity,
                                                           uint32_t *__restrict__ overflow, uint32_t *__restrict__ unsearchable) {
    extern __shared__ uint32_t s_tot[];  // [nl + 1]
    __shared__ unsigned long long s_base;
    __shared__ uint32_t s_long;
    const KindLong g = longs[blockIdx.x];
    const uint32_t lane = lane_id(), wave = threadIdx.x >> 6;
    uint32_t *cnt = seg_cnt + (size_t)g.first * nl;
    if (threadIdx.x == 0) s_long = 0u;
    __syncthreads();
    for (uint32_t li = threadIdx.x; li < nl; li += blockDim.x) {
        uint32_t run = 0;
        for (uint32_t k = 0; k < g.n; ++k) {
            const uint32_t c = cnt[(size_t)k * nl + li];
            cnt[(size_t)k * nl + li] = run;
            run += c;
        }
        s_tot[li] = run;
        if (run > 0xffffu) s_long = 1u;  // (the pair tables hold list POSITIONS in 16 bits)
    }
    __syncthreads();
    if (wave == 0) {
        uint32_t run = 0;
        for (uint32_t l0 = 0; l0 < nl; l0 += 64u) {
            const uint32_t li = l0 + lane;
            const uint32_t c = li < nl ? s_tot[li] : 0u;
            const uint32_t incl = wave_scan_add(c);
            if (li < nl) s_tot[li] = run + incl - c;
            run += (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
        }
        if (lane == 0) {
            s_tot[nl] = run;
            // a kind list beyond 65535 atoms (a wide structure): THIS structure is not searchable with this library -- it gets
            // empty lists (no tables, no hits) and is listed by tm_result_unsearched; the rest of the batch is searched
            unsearchable[blockIdx.x] = s_long;
            unsigned long long base = ~0ull;
            if (s_long) atomicOr(overflow, 0x100u);
            else {
                base = atomicAdd(cursor, (unsigned long long)run);
                if (base + run > capacity) {
                    atomicOr(overflow, 1u);
                    base = ~0ull;
                }
            }
            s_base = base;
        }
    }
    __syncthreads();
    const unsigned long long base = s_base;
    uint32_t *off_out = kl_off + (size_t)g.s * (nl + 1);
    for (uint32_t li = threadIdx.x; li <= nl; li += blockDim.x)
        off_out[li] = base == ~0ull ? 0u : (uint32_t)(base + s_tot[li]);
    for (uint32_t li = threadIdx.x; li < nl; li += blockDim.x)
        for (uint32_t k = 0; k < g.n; ++k)
            cnt[(size_t)k * nl + li] = base == ~0ull ? 0xffffffffu : (uint32_t)(base + s_tot[li]) + cnt[(size_t)k * nl + li];
}

// Largest |centred coordinate| of a batch's atom records, as float bits (non-negative floats order like integers):
// what the fp32 prefilter's slack must cover (tm_device.hpp, prefilter_slack).
__global__ __launch_bounds__(256) void k_extent(const float4 *__restrict__ rec, uint32_t n, uint32_t *__restrict__ out) {
    float m = 0.f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const float4 r = rec[i];
        m = fmaxf(m, fmaxf(fabsf(r.x), fmaxf(fabsf(r.y), fabsf(r.z))));
    }
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out, __float_as_uint(m));
}

// A query resets a dozen counters and flags before its kernels run and copies a few afterwards; one small launch per
// group instead of a hipMemsetAsync / hipMemcpyAsync per word (the host side of a single-structure query is enqueue
// time: DESIGN.md 6, configs[1]).
struct ZeroList {
    void *p[8];
    uint32_t words[8];  // 32-bit words to clear at p[i]
};
__global__ __launch_bounds__(256) void k_zero(ZeroList z) {
    for (int i = 0; i < 8; ++i)
        for (uint32_t w = threadIdx.x; w < z.words[i]; w += blockDim.x) reinterpret_cast<uint32_t *>(z.p[i])[w] = 0u;
}

// one thread: what the index and table builds this query searches needed and whether they overflowed, carried into the
// query's own words (bits 3 and 4 of its overflow word; see carry_build_outcome in tm_api.hip)
__global__ void k_carry_outcome(const unsigned long long *__restrict__ index_cursor, const uint32_t *__restrict__ index_overflow,
                                const unsigned long long *__restrict__ tab_cursors, const uint32_t *__restrict__ tab_overflow,
                                unsigned long long *__restrict__ pool_used, unsigned long long *__restrict__ tab_used,
                                uint32_t *__restrict__ query_overflow) {
    *pool_used = *index_cursor;
    uint32_t bits = (*index_overflow & 1u) ? 8u : 0u;
    if (*index_overflow & 0x100u) bits |= 0x100u;  // a kind list beyond 16-bit positions (tm_result_wait: TM_E_UNSUPPORTED)
    if (tab_cursors) {
        tab_used[0] = tab_cursors[0];
        tab_used[1] = tab_cursors[1];
        if (*tab_overflow) bits |= 16u;
    }
    if (bits) atomicOr(query_overflow, bits);
}

// ------------------------------------------------------------------ pair tables
//
// The first two search levels of every template used to test the whole cross product of two kind lists against one
// distance window (about 300 tests per template, of which 1.5 % pass), and the third level did the same for every
// survivor.  Templates share lead kinds: the 6780 default templates need about a thousand distinct (kind A, kind B,
// distance class) combinations.  So the distance work is done ONCE per structure and combination -- every pair of
// list positions within the class's distance limit, stored row by row with its squared distance -- and a template's
// first level becomes a scan of one short table segment (14 entries on average), its second a lookup of a few rows.
//
//   k_tab_layout   per structure: where each table's row slots live (bump allocation)
//   k_tab_build    per (structure, share of its tables): row offsets, bump allocation of the entries, entries
//
// Inside a table the order is deterministic: rows by ascending position in list A, entries by ascending position in
// list B, row offsets relative to the table's first entry.  WHERE a table sits in the entry pool is not (bump
// allocation), and does not matter.

struct TabWork {  // one workgroup of k_tab_count / k_tab_fill: tables part, part + parts, ... of structure s
    uint32_t s;
    uint16_t part, parts;
};

__global__ __launch_bounds__(256) void k_tab_layout(uint32_t n_lead, BatchView B, const TableDesc *__restrict__ tables,
                                                     uint32_t *__restrict__ tab_dir,
                                                     unsigned long long *__restrict__ cursors,
                                                     unsigned long long row_capacity, uint32_t *__restrict__ overflow,
                                                     uint32_t *__restrict__ tab_ok) {
    __shared__ uint32_t s_part[256];
    __shared__ unsigned long long s_base;
    const uint32_t s = blockIdx.x, nt = B.n_tables;
    const uint32_t *kl = B.kl_off + (size_t)s * (n_lead + 1);
    const uint32_t per = (nt + blockDim.x - 1) / blockDim.x;
    const uint32_t lo = min(threadIdx.x * per, nt), hi = min(lo + per, nt);
    uint32_t sum = 0;
    for (uint32_t t = lo; t < hi; ++t) sum += kl[tables[t].li_a + 1] - kl[tables[t].li_a] + 1u;
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (uint32_t i = 0; i < blockDim.x; ++i) { const uint32_t v = s_part[i]; s_part[i] = run; run += v; }
        unsigned long long base = atomicAdd(&cursors[0], (unsigned long long)run);
        if (base + run > row_capacity) {
            atomicOr(overflow, 1u);
            base = ~0ull;
        }
        s_base = base;
        tab_ok[s] = base == ~0ull ? 0u : 1u;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    uint32_t run = (uint32_t)base + s_part[threadIdx.x];
    for (uint32_t t = lo; t < hi; ++t) {
        tab_dir[(size_t)s * nt + t] = base == ~0ull ? 0u : run;
        run += kl[tables[t].li_a + 1] - kl[tables[t].li_a] + 1u;
    }
}

// a table with more than this many (position A, position B) pairs is built by all waves of its workgroup together
// (k_tab_build's shared_limit; ENZYMM_AMD_TAB_SHARED overrides it -- tests send ordinary structures down that path)
constexpr unsigned long long kTabShared = 256ull * 1024ull;
#ifndef TM_DERIVE_MIN
#define TM_DERIVE_MIN 512
#endif
constexpr unsigned long long kDeriveMin = TM_DERIVE_MIN;  // cross product above which a group's smaller-limit tables are derived from its leader
constexpr int kTabBlock = 1024;  // threads of a k_tab_build workgroup: 16 waves share one LDS copy of the lead records

// One sweep over the tiles of one table.  A tile lays TI rows x TJ columns over the 64 lanes, row-major (TJ = the
// power of two covering list B, at most 64); `lead(p)` returns the atom record at kind-list position p.  A lane's
// pair passes when both positions exist, the atoms lie in different residues and within the distance limit.
//   count sweep (kFill = false): only the number of passing pairs (one scalar popcount of the vote per tile)
//   fill sweep  (kFill = true):  rows are visited in order, so the running count is the prefix sum: rows[i] = passing
//                                pairs before row i, and every passing pair is written at running count + its rank
//                                among the passing lanes of the tile (lanes of earlier rows come first in the vote).
// Two tiles are in flight per iteration (two independent chains of LDS read -> distance -> vote): the loop is bound
// by those latencies, not by issue.  When list B fits one tile its operand is loaded once per table.
// Returns the number of passing pairs.
// A sweep may cover a block of rows of a larger table (several waves share a big table): `rows` then points at the
// block's first row slot, `i_base` is that row's number and `row_base` the number of passing pairs in the rows before
// it -- entries and row offsets are written for the whole table's numbering.  The return value counts this block only.
template <class T>
__device__ __forceinline__ T *lane_at(T *uniform, uint32_t bytes) {
    return reinterpret_cast<T *>(reinterpret_cast<char *>(uniform) + bytes);
}
template <bool kFill, class Lead, class LeadB>
__device__ __forceinline__ uint32_t tab_sweep(Lead lead, LeadB lead_b, uint32_t a_lo, uint32_t n_a, uint32_t b_lo,
                                              uint32_t n_b, float dmax2, uint32_t *rows, uint2 *ent,
                                              uint32_t row_base = 0u, uint32_t i_base = 0u) {
    const uint32_t lane = threadIdx.x & 63;
    uint32_t running = row_base;
    if (n_b == 0u || n_a == 0u) {  // nothing to pair with: every row is empty
        if (kFill)
            for (uint32_t i = lane; i < n_a; i += 64u) rows[i] = row_base;
        return 0u;
    }
    const uint32_t shift = n_b <= 1u ? 0u : min(6u, 32u - (uint32_t)__clz((int)(n_b - 1u)));
    const uint32_t TJ = 1u << shift, TI = 64u >> shift;
    const uint32_t li = lane >> shift, lj = lane & (TJ - 1u);
    // (the last step is ONE comparison -- of the distance, or of +inf for a lane that does not count: the vote is then the
    // comparison's own lane mask; a conjunction of three is voted on through a select and a second comparison)
    auto test = [&](const float4 &ra, const float4 &rb, bool valid, float &d2) {
        const float dx = ra.x - rb.x, dy = ra.y - rb.y, dz = ra.z - rb.z;
        d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        const bool counts = valid & (meta_res(__float_as_uint(ra.w)) != meta_res(__float_as_uint(rb.w)));
        return (counts ? d2 : __builtin_inff()) <= dmax2;
    };
    if (n_b <= TJ) {
        // ---- list B is one tile wide: the column operand never changes, rows go by 2 x TI
        const bool vj = lj < n_b;
        const float4 rb = lead_b(b_lo, lj, vj);
        for (uint32_t i0 = 0; i0 < n_a; i0 += 2u * TI) {
            const uint32_t iA = i0 + li, iB = iA + TI;
            const bool vA = iA < n_a, vB = iB < n_a;
            const float4 raA = lead(a_lo, iA, vA), raB = lead(a_lo, iB, vB);
            float dA, dB;
            const bool okA = test(raA, rb, vA & vj, dA), okB = test(raB, rb, vB & vj, dB);
            const unsigned long long allA = ballot64(okA), allB = ballot64(okB);
            const uint32_t nA = (uint32_t)__popcll(allA);
            if (kFill) {
                // (the votes below the first lane of a tile row are those of the tile's earlier rows: a row's offset)
                // (addresses as a wave-uniform pointer plus a 32-bit byte offset per lane: the store takes the pointer from
                // scalar registers, no 64-bit address arithmetic per lane)
                const uint32_t rA = rank_in(allA), rB = nA + rank_in(allB);
                if (okA) *lane_at(ent + running, rA << 3) = make_uint2((iA + i_base) | (lj << 16), __float_as_uint(dA));
                if (okB) *lane_at(ent + running, rB << 3) = make_uint2((iB + i_base) | (lj << 16), __float_as_uint(dB));
                if (lj == 0u) {
                    if (vA) *lane_at(rows + i0, li << 2) = running + rA;
                    if (vB) *lane_at(rows + i0 + TI, li << 2) = running + rB;
                }
            }
            running += nA + (uint32_t)__popcll(allB);
        }
        return running - row_base;
    }
    // ---- list B is longer than 64: one row per tile (TI = 1), columns go by 2 x 64
    for (uint32_t i = 0; i < n_a; ++i) {
        const float4 ra = lead(a_lo, i, true);
        if (kFill)
            if (lane == 0u) rows[i] = running;
        for (uint32_t j0 = 0; j0 < n_b; j0 += 128u) {
            const uint32_t jA = j0 + lane, jB = jA + 64u;
            const bool vA = jA < n_b, vB = jB < n_b;
            const float4 rbA = lead_b(b_lo, jA, vA), rbB = lead_b(b_lo, jB, vB);
            float dA, dB;
            const bool okA = test(ra, rbA, vA, dA), okB = test(ra, rbB, vB, dB);
            const unsigned long long allA = ballot64(okA), allB = ballot64(okB);
            const uint32_t nA = (uint32_t)__popcll(allA);
            if (kFill) {
                if (okA) *lane_at(ent + running, rank_in(allA) << 3) = make_uint2((i + i_base) | (jA << 16), __float_as_uint(dA));
                if (okB) *lane_at(ent + running, (nA + rank_in(allB)) << 3) = make_uint2((i + i_base) | (jB << 16), __float_as_uint(dB));
            }
            running += nA + (uint32_t)__popcll(allB);
        }
    }
    return running - row_base;
}

// The same sweep for a LONG structure (a complex of tens of thousands of atoms; k_tab_gather laid the record of every
// kind-list position out in global memory and gave every 16 consecutive positions of a list their bounding sphere):
// 97 % of the cross product of two lists of such a structure lies beyond any class limit, and the lists follow the
// chains, so 16 consecutive entries are neighbours in space.  Per row (a wave-uniform atom) the chunks of list B are
// tested first, lane = chunk -- centre within limit + radius (+ 0.01 A: the triangle inequality in fp32) --, and only the
// surviving chunks are laid over the lanes, four (row, chunk) items of 16 columns per tile, rows ascending and chunks
// ascending inside a row: the entries come out in the order of the plain sweep, so the tables are the same byte for
// byte.  A row without a surviving chunk takes an empty item: the lane that starts a row writes the row's offset.
constexpr uint32_t kCullMin = 128;  // list B at least this long
template <bool kFill>
__device__ __forceinline__ uint32_t tab_sweep_culled(const float4 *__restrict__ lr_a, uint32_t n_a,
                                                     const float4 *__restrict__ lr_b, uint32_t n_b,
                                                     const float4 *__restrict__ sph_b, float dmax2, uint32_t *rows, uint2 *ent,
                                                     uint32_t row_base = 0u, uint32_t i_base = 0u) {
    const uint32_t lane = threadIdx.x & 63, slot = lane >> 4, e = lane & 15u;
    const uint32_t n_chunks = (n_b + 15u) >> 4;
    const float lim = __builtin_sqrtf(dmax2) + 0.01f;
    uint32_t running = row_base, filled = 0;
    uint32_t my_row = 0, my_chunk = 0xffffffffu;
    bool my_first = false;
    auto flush = [&]() {
        const bool item = slot < filled;
        const uint32_t j = (my_chunk << 4) + e;  // (an empty item: beyond every list)
        const bool valid = item & (my_chunk != 0xffffffffu) & (j < n_b);
        const float4 rb = lr_b[valid ? j : 0u], ra = lr_a[item ? my_row : 0u];
        const float dx = ra.x - rb.x, dy = ra.y - rb.y, dz = ra.z - rb.z;
        const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        const bool counts = valid & (meta_res(__float_as_uint(ra.w)) != meta_res(__float_as_uint(rb.w)));
        const bool ok = (counts ? d2 : __builtin_inff()) <= dmax2;
        const unsigned long long all = ballot64(ok);
        if (kFill) {
            const uint32_t rk = rank_in(all);
            if (ok) *lane_at(ent + running, rk << 3) = make_uint2((my_row + i_base) | (j << 16), __float_as_uint(d2));
            if (item && e == 0u && my_first) rows[my_row] = running + rk;
        }
        running += (uint32_t)__popcll(all);
        filled = 0;
    };
    const float4 far = make_float4(1.0e18f, 1.0e18f, 1.0e18f, 0.f);
    const float4 sp_first = lane < n_chunks ? sph_b[lane] : far;  // (a list of at most 1024 entries: one pass per row)
    for (uint32_t i0 = 0; i0 < n_a; i0 += 64u) {
        const uint32_t nrows = min(64u, n_a - i0);
        const float4 ra = lr_a[i0 + min(lane, nrows - 1u)];  // the block's rows, lane = row
        for (uint32_t r = 0; r < nrows; ++r) {
            const float ax = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ra.x), (int)r));
            const float ay = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ra.y), (int)r));
            const float az = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(ra.z), (int)r));
            bool first = true;
            for (uint32_t c0 = 0; c0 < n_chunks; c0 += 64u) {
                float4 sp = sp_first;
                if (c0 != 0u) sp = c0 + lane < n_chunks ? sph_b[c0 + lane] : far;
                const float dx = sp.x - ax, dy = sp.y - ay, dz = sp.z - az, t = lim + sp.w;
                unsigned long long keep = ballot64(__builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx)) <= t * t);
                while (keep != 0ull) {
                    const uint32_t c = c0 + (uint32_t)__ffsll((long long)keep) - 1u;
                    keep &= keep - 1ull;
                    if (slot == filled) { my_row = i0 + r; my_chunk = c; my_first = first; }
                    first = false;
                    if (++filled == 4u) flush();
                }
            }
            if (first) {  // no chunk of list B near this row: an empty item carries the row's offset
                if (slot == filled) { my_row = i0 + r; my_chunk = 0xffffffffu; my_first = true; }
                if (++filled == 4u) flush();
            }
        }
    }
    if (filled != 0u) flush();
    return running - row_base;
}

// A table derived from its group's leader (same kind lists, smaller distance limit): the leader's entries that lie
// within the limit, in the leader's order -- exactly what a sweep of the cross product with that limit writes.  One
// streaming pass, lane = entry of the leader: survivors are compacted into ent_c (ballot + rank), and the first entry the
// leader holds of a row fixes that row's offset (an entry carries its row: the low 16 bits of its first word).  Rows the
// leader has no entry of take the offset of the next row that has one (a second, short pass over the rows, back to
// front).  Returns the number of entries written.
__device__ __forceinline__ uint32_t tab_derive(const uint32_t *rows_l, const uint2 *ent_l, uint32_t total_l, uint32_t n_a,
                                               float dmax2, uint32_t *rows_c, uint2 *ent_c, uint32_t lane) {
    uint32_t running = 0, prev_row = 0xffffffffu;
    for (uint32_t e0 = 0; e0 < total_l; e0 += 64u) {
        const uint32_t e = e0 + lane;
        uint2 ent = make_uint2(0xffffffffu, 0x7f800000u);  // past the end: row 0xffff, distance +inf
        if (e < total_l) ent = ent_l[e];
        const bool ok = e < total_l && __uint_as_float(ent.y) <= dmax2;
        const unsigned long long mask = ballot64(ok);
        const uint32_t before = running + rank_in(mask);
        const uint32_t row = ent.x & 0xffffu;
        uint32_t left = lane_get(row, (lane + 63u) & 63u);  // the row of the entry before this one
        if (lane == 0u) left = prev_row;
        if (e < total_l && row != left) rows_c[row] = before;
        if (ok) ent_c[before] = ent;
        running += (uint32_t)__popcll(mask);
        prev_row = (uint32_t)__builtin_amdgcn_readlane((int)row, 63);
    }
    // rows without an entry in the leader: the offset of the next row that has one (offsets never decrease with the row,
    // so walking the rows back to front, "the nearest row with an entry behind me" is the smallest offset seen so far)
    wave_sync();  // (the offsets written above are read by other lanes of this wave: its stores are done, none of those
                  // lines was read before, so nothing stale sits in this CU's L1; a device-scope fence would write L2 back)
    uint32_t carry = running;
    for (uint32_t base = (n_a + 63u) & ~63u; base != 0u; base -= 64u) {
        const uint32_t i = base - 1u - lane;  // lane 0 takes the last row of the chunk
        uint32_t v = 0xffffffffu;
        if (i < n_a && rows_l[i + 1u] != rows_l[i]) v = rows_c[i];
        // running minimum over the lanes = bitwise complement of the running maximum of the complements
        const uint32_t m = min(~wave_scan_max(~v), carry);
        if (i < n_a && v == 0xffffffffu) rows_c[i] = m;
        carry = (uint32_t)__builtin_amdgcn_readlane((int)m, 63);
    }
    return running;
}

// The same table from a leader of at most kDeriveVotes x 64 entries (most are: a few hundred entries), without the scattered
// row stores, the exchange with the neighbouring lane and the second sweep over freshly written offsets: the vote of every
// 64-entry chunk and the count before it stay in the wave (chunk k in lane k of three registers); a row's offset is then
// the count before the chunk its first leader entry falls into plus the votes below that entry -- one load of the
// leader's offset per row (asked for before the entries are streamed), three lane reads, a population count.
constexpr uint32_t kDeriveVotes = 63;
__device__ __forceinline__ uint32_t tab_derive_small(const uint32_t *rows_l, const uint2 *ent_l, uint32_t total_l, uint32_t n_a,
                                                     float dmax2, uint32_t *rows_c, uint2 *ent_c, uint32_t lane) {
    uint32_t first_row = 0;
    if (lane < n_a) first_row = rows_l[lane];
    uint32_t vote_lo = 0, vote_hi = 0, before = 0;  // lane k: chunk k; the lane after the last chunk: no votes, the total
    uint32_t running = 0, k = 0;
    for (uint32_t e0 = 0; e0 < total_l; e0 += 64u, ++k) {
        const uint32_t e = e0 + lane;
        uint2 ent = make_uint2(0u, 0x7f800000u);
        if (e < total_l) ent = ent_l[e];
        const bool ok = e < total_l && __uint_as_float(ent.y) <= dmax2;
        const unsigned long long mask = ballot64(ok);
        if (ok) ent_c[running + rank_in(mask)] = ent;
        if (lane == k) { vote_lo = (uint32_t)mask; vote_hi = (uint32_t)(mask >> 32); before = running; }
        running += (uint32_t)__popcll(mask);
    }
    if (lane == k) before = running;
    for (uint32_t i0 = 0; i0 < n_a; i0 += 64u) {
        const uint32_t i = i0 + lane;
        uint32_t e = first_row;
        if (i0 != 0u) e = i < n_a ? rows_l[i] : 0u;
        const uint32_t c = e >> 6, bit = e & 63u;
        const unsigned long long votes = ((unsigned long long)lane_get(vote_hi, c) << 32) | lane_get(vote_lo, c);
        const uint32_t count = lane_get(before, c) + (uint32_t)__popcll(votes & ((1ull << bit) - 1ull));
        if (i < n_a) rows_c[i] = count;
    }
    return running;
}

// All tables of one work item (structure, part), one wave per table.  Entries are bump-allocated per WAVE in chunks of
// tab_chunk() entries (one atomic per chunk; one per table would serialise a million atomics on one address): a table whose
// cross product fits the rest of the wave's chunk is swept ONCE -- rows and entries written as they are found, the
// chunk advanced by what was written; a larger table is counted first, allocated exactly, then filled.  kLds: the
// atom records of every kind-list position were gathered into LDS first, so tiles read their operands at
// consecutive LDS addresses.
// entries a wave reserves at a time for a structure of n atoms: about 3.5 per atom, between 1 Ki and 8 Ki
__host__ __device__ inline uint32_t tab_chunk(uint32_t n_atoms) {
    uint32_t c = 1024;
    while (c < 8192 && c < 2 * n_atoms) c <<= 1;
    return c;
}
// kSrc: where the atom record of kind-list position p comes from -- 0: rec[pool[p]] in global memory; 1: s_lr[p], a copy
// of the record per POSITION in LDS (one read per operand, 16 B per position: about 100 KB for a 300-residue structure,
// one workgroup per CU); 2: the structure's atom records and its kind lists in LDS (two dependent reads per operand,
// 16 B per atom + 2 B per position: about 50 KB, two workgroups per CU).
// kWide (kSrc 0 only): a structure of more than kNarrowAtoms atoms -- a kind-list entry is pool | pool_hi << 16.
// 3: the record of every kind-list position in GLOBAL memory (k_tab_gather: s_lr[p - p0]) with the bounding spheres of its
// 16-entry chunks (sph_s): one read per operand, and the sweeps of long lists skip the chunks beyond the limit.
template <int kSrc, bool kWide = false>
__device__ __forceinline__ void tab_build_tables(const BatchView &B, const TableDesc *__restrict__ tables,
                                                 const TabWork w, const float4 *rec, const uint16_t *pool,
                                                 const float4 *s_lr, const uint32_t *kl,
                                                 uint32_t *__restrict__ tab_rows, uint2 *__restrict__ tab_ent,
                                                 uint2 *__restrict__ tab_seg, unsigned long long *__restrict__ cursors,
                                                 unsigned long long ent_capacity, uint32_t *__restrict__ overflow,
                                                 uint32_t *__restrict__ tab_ok, uint32_t dbg_mode, uint32_t *s_count,
                                                 unsigned long long *s_first, uint32_t n_lead, float4 *s_stage,
                                                 uint32_t stage_cap, unsigned long long shared_limit, bool rows_ok,
                                                 unsigned long long derive_min, const float4 *sph_s = nullptr, uint32_t p0 = 0u) {
    const uint32_t nt = B.n_tables;
    const uint32_t wave = uni(threadIdx.x >> 6), n_waves = blockDim.x >> 6, lane = threadIdx.x & 63;
    // the record at position `first + k` of a kind list; `valid`: k lies inside the list.  A lane beyond the list is
    // masked out of the vote whatever it reads: only a read from GLOBAL memory has to stay inside the arrays (position
    // `first` then) -- an LDS read past the list returns a neighbouring list's atom, past the allocation zero
    auto lead = [&](uint32_t first, uint32_t k, bool valid) -> float4 {
        if constexpr (kSrc == 3) return s_lr[first - p0 + (valid ? k : 0u)];
        else if constexpr (kSrc == 1) return s_lr[first + k];
        else if constexpr (kSrc == 2) {
            const uint32_t atom = ((const __attribute__((address_space(3))) uint16_t *)pool)[first + k];
            const auto *r = (const __attribute__((address_space(3))) float *)rec + 4u * atom;
            return make_float4(r[0], r[1], r[2], r[3]);
        } else if constexpr (kWide) {
            const uint32_t at = first + (valid ? k : 0u);
            return rec[(uint32_t)pool[at] | ((uint32_t)B.kl_pool_hi[at] << 16)];
        } else return rec[pool[first + (valid ? k : 0u)]];
    };
    // entries [first, first + n) of the pool, or ~0 when the pool is exhausted (the query then reports the overflow)
    auto allocate = [&](unsigned long long n) -> unsigned long long {
        unsigned long long first = 0;
        if (lane == 0) {
            first = atomicAdd(&cursors[1], n);
            if (first + n > ent_capacity) {
                atomicOr(overflow, 2u);
                atomicOr(&tab_ok[w.s], 2u);
                first = ~0ull;
            }
        }
        return uni64(first);
    };
    // one sweep of rows [a_first, a_first + na) of list A against list B (li_b: its lead-kind index)
    auto sweep_any = [&](auto fill_tag, uint32_t a_first, uint32_t na, uint32_t b_first, uint32_t nb, uint32_t li_b, float dmax2,
                         uint32_t *rows, uint2 *ent, uint32_t row_base, uint32_t i_base) -> uint32_t {
        constexpr bool kFill = decltype(fill_tag)::value;
        if constexpr (kSrc == 3) {
            if (nb >= kCullMin && dmax2 < 1.0e6f)
                return tab_sweep_culled<kFill>(s_lr + (a_first - p0), na, s_lr + (b_first - p0), nb,
                                               sph_s + (((b_first - p0) >> 4) + li_b), dmax2, rows, ent, row_base, i_base);
        }
        return tab_sweep<kFill>(lead, lead, a_first, na, b_first, nb, dmax2, rows, ent, row_base, i_base);
    };
    const std::true_type fill_yes{};
    const std::false_type fill_no{};
    const uint32_t kTabChunk = tab_chunk(B.s_off[w.s + 1] - B.s_off[w.s]);
    unsigned long long chunk = ~0ull;  // the wave's current chunk and how much of it is used
    uint32_t used = kTabChunk;
    // (tables are taken one at a time from a counter of the workgroup, not dealt out by wave number: with a fixed share
    // the waves of a workgroup ran out of tables after 0.87 of its duration on average, the first after 0.75 --
    // tests/gpu_wg_timeline.py)
    uint32_t *next_table = s_count + kTabBlock / 64;
    for (;;) {
        uint32_t taken = 0;
        if (lane == 0) taken = atomicAdd(next_table, 1u);
        const uint32_t t = w.part + w.parts * uni(taken);
        if (t >= nt) break;
        const TableDesc T = tables[t];
        // wave-uniform values, said so: loop counters and branches of the sweep stay scalar
        const uint32_t a_lo = uni(kl[T.li_a]), n_a = uni(kl[T.li_a + 1]) - a_lo, b_lo = uni(kl[T.li_b]),
                       n_b = uni(kl[T.li_b + 1]) - b_lo;
        uint32_t *rows = tab_rows + B.tab_dir[(size_t)w.s * nt + t];
        const unsigned long long bound = (unsigned long long)n_a * n_b;
        if (bound > shared_limit) continue;  // a big table: all waves build it together, below
        // Tables of one group (same two lists, different limits): when the cross product is large, only the leader is
        // swept and the others are derived from its entries by the wave that builds it (below); a small cross product
        // -- a few dozen tiles out of LDS -- is cheaper to sweep per table than to read a leader back from L2.
        const bool derive = bound > derive_min;
        if (derive && T.src != t) continue;
        unsigned long long first = 0;
        uint32_t total = 0;
        if (dbg_mode == 2u) {
            // timing experiment: no sweeps
        } else if (bound <= kTabChunk) {
            if (used + (uint32_t)bound > kTabChunk) {
                chunk = allocate(kTabChunk);
                used = 0;
            }
            if (rows_ok && chunk != ~0ull) {
                first = chunk + used;
                total = sweep_any(fill_yes, a_lo, n_a, b_lo, n_b, T.li_b, T.dmax2, rows, tab_ent + first, 0u, 0u);
            } else {
                // no room (or no row slots): count only, and let the wave's chunk fill up exactly as it would have --
                // the entry cursor then ends on what a build with room hands out, and ONE retry is sized right
                first = ~0ull;
                total = sweep_any(fill_no, a_lo, n_a, b_lo, n_b, T.li_b, T.dmax2, nullptr, nullptr, 0u, 0u);
            }
            used += total;
        } else {
            total = sweep_any(fill_no, a_lo, n_a, b_lo, n_b, T.li_b, T.dmax2, nullptr, nullptr, 0u, 0u);
            first = allocate(total);
            if (!rows_ok) first = ~0ull;
            if (first != ~0ull && dbg_mode == 0u) sweep_any(fill_yes, a_lo, n_a, b_lo, n_b, T.li_b, T.dmax2, rows, tab_ent + first, 0u, 0u);
        }
        if (lane == 0 && rows_ok) {
            tab_seg[(size_t)w.s * nt + t] = first == ~0ull ? make_uint2(0u, 0u) : make_uint2((uint32_t)first, (uint32_t)first + total);
            rows[n_a] = total;
        }
#ifdef TM_STATS
        {
            const uint32_t bl = min(23u, 64u - (uint32_t)__clzll((long long)bound));
            TM_STAT((T.src == t ? 96 : 120) + bl, bound);
            TM_STAT((T.src == t ? 144 : 168) + bl, total);
            TM_STAT(284 + min(15u, bl), 1);
        }
#endif
        // ---- the tables derived from this one (same lists, smaller limits): its entries within their limit
        if (derive && T.next != 0xffffu && dbg_mode != 2u) {
            const bool have = first != ~0ull && rows_ok && dbg_mode == 0u;
            if (have) wave_sync();  // the leader's rows and entries are read back below (by this wave, which wrote them)
            for (uint32_t d = T.next; d != 0xffffu;) {
                const TableDesc Td = tables[d];
                uint32_t *rows_d = tab_rows + B.tab_dir[(size_t)w.s * nt + d];
                unsigned long long first_d = ~0ull;
                uint32_t total_d = 0;
                // room for as many entries as the leader has (an upper bound), from the wave's chunk like a small table;
                // a build that only counts (no room, no row slots) moves the cursors exactly as one with room does
                if (total <= kTabChunk) {
                    if (used + total > kTabChunk) {
                        chunk = allocate(kTabChunk);
                        used = 0;
                    }
                    if (have && chunk != ~0ull) first_d = chunk + used;
                } else {
                    first_d = allocate(total);
                    if (!have) first_d = ~0ull;
                }
                if (first_d != ~0ull)
                    total_d = total <= kDeriveVotes * 64u
                                  ? tab_derive_small(rows, tab_ent + first, total, n_a, Td.dmax2, rows_d, tab_ent + first_d, lane)
                                  : tab_derive(rows, tab_ent + first, total, n_a, Td.dmax2, rows_d, tab_ent + first_d, lane);
                else total_d = sweep_any(fill_no, a_lo, n_a, b_lo, n_b, T.li_b, Td.dmax2, nullptr, nullptr, 0u, 0u);
                if (total <= kTabChunk) used += total_d;
                if (lane == 0 && rows_ok) {
                    tab_seg[(size_t)w.s * nt + d] = first_d == ~0ull ? make_uint2(0u, 0u) : make_uint2((uint32_t)first_d, (uint32_t)first_d + total_d);
                    rows_d[n_a] = total_d;
                }
                d = Td.next;
            }
        }
    }
    // ---- big tables of this part (kind lists of several hundred atoms: large complexes): the rows are cut into one
    // block per wave; every wave counts its block, the counts are summed up in LDS, the entries allocated once and
    // every wave fills its block behind the blocks before it.  One wave per table left a 25 k-atom complex waiting for
    // its largest table (9 ms for a single structure).
    // (a structure whose longest kind list squared stays below the limit has none: most have none, and looking at
    // every table's descriptor to find that out cost more than the tables of a 300-residue structure take to build)
    uint32_t longest = 0;
    for (uint32_t k = lane; k < n_lead; k += 64u) longest = max(longest, kl[k + 1] - kl[k]);
    longest = wave_scan_max(longest);
    longest = (uint32_t)__builtin_amdgcn_readlane((int)longest, 63);
    if ((unsigned long long)longest * longest <= shared_limit) return;  // the same in every wave of the workgroup
    for (uint32_t t = w.part; t < nt; t += w.parts) {
        const TableDesc T = tables[t];
        const uint32_t a_lo = uni(kl[T.li_a]), n_a = uni(kl[T.li_a + 1]) - a_lo, b_lo = uni(kl[T.li_b]),
                       n_b = uni(kl[T.li_b + 1]) - b_lo;
        if ((unsigned long long)n_a * n_b <= shared_limit) continue;  // same value in every wave: the barriers below are safe
        uint32_t *rows = tab_rows + B.tab_dir[(size_t)w.s * nt + t];
        const uint32_t block = (n_a + n_waves - 1u) / n_waves;
        const uint32_t i0 = min(n_a, wave * block), i1 = min(n_a, i0 + block);
        // a structure too large for the LDS copy of all its lead records reads them through L2 (two dependent loads
        // per operand): list B of a big table, which every row block sweeps in full, is staged in that unused LDS
        const bool staged = kSrc == 0 && n_b <= stage_cap;
        if (staged) {
            for (uint32_t j = threadIdx.x; j < n_b; j += blockDim.x) s_stage[j] = lead(b_lo, j, true);
            __syncthreads();
        }
        // (two instantiations: a run-time choice per operand load would compile into a flat load)
        auto build = [&](auto lead_b) {
            uint32_t mine = 0;
            if constexpr (kSrc == 3) mine = dbg_mode == 2u ? 0u : sweep_any(fill_no, a_lo + i0, i1 - i0, b_lo, n_b, T.li_b, T.dmax2, nullptr, nullptr, 0u, 0u);
            else mine = dbg_mode == 2u ? 0u : tab_sweep<false>(lead, lead_b, a_lo + i0, i1 - i0, b_lo, n_b, T.dmax2, nullptr, nullptr);
            if (lane == 0) s_count[wave] = mine;
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t run = 0;
                for (uint32_t k = 0; k < n_waves; ++k) { const uint32_t c = s_count[k]; s_count[k] = run; run += c; }
                TM_STAT(280, 1); TM_STAT(281, (unsigned long long)n_a * n_b); TM_STAT(282, run); TM_STAT(283, n_b > stage_cap ? 1 : 0);
                unsigned long long first = atomicAdd(&cursors[1], (unsigned long long)run);
                if (first + run > ent_capacity) {
                    atomicOr(overflow, 2u);
                    atomicOr(&tab_ok[w.s], 2u);
                    first = ~0ull;
                }
                if (!rows_ok) first = ~0ull;
                *s_first = first;
                if (rows_ok) {
                    tab_seg[(size_t)w.s * nt + t] = first == ~0ull ? make_uint2(0u, 0u) : make_uint2((uint32_t)first, (uint32_t)first + run);
                    rows[n_a] = run;
                }
            }
            __syncthreads();
            const unsigned long long first = *s_first;
            if (first != ~0ull && dbg_mode == 0u) {
                if constexpr (kSrc == 3) sweep_any(fill_yes, a_lo + i0, i1 - i0, b_lo, n_b, T.li_b, T.dmax2, rows + i0, tab_ent + first, s_count[wave], i0);
                else tab_sweep<true>(lead, lead_b, a_lo + i0, i1 - i0, b_lo, n_b, T.dmax2, rows + i0, tab_ent + first, s_count[wave], i0);
            }
        };
        if (staged) build([&](uint32_t first, uint32_t k, bool valid) -> float4 { return s_stage[k]; });
        else build(lead);
        __syncthreads();  // s_count, s_first and the staged list are reused by the next big table
    }
}

// Before the table build of LONG structures (those the host sends to the large-structure launch): the record of every
// kind-list position, position by position, and the bounding sphere of every 16 consecutive positions of a list (centre =
// mean, radius = farthest member, rounded up).  A structure's region is laid out by the host (6 positions per atom: a
// protein needs 3); a structure whose lists do not fit keeps ~0 as its base and is built the old way, records through
// the kind lists.  The sphere of chunk c of list L sits at ((first position of L - p0) >> 4) + L + c of the structure's
// sphere region: lists start at any position, chunks start at their list's start, and the + L keeps the slots of
// neighbouring lists apart.
struct TabGather { uint32_t s, lr_base, sph_base, cap; };
struct TabAux {
    const float4 *lr, *sph;
    const uint32_t *lr_base, *sph_base;  // per structure; lr_base ~0: not gathered
};
__global__ __launch_bounds__(256) void k_tab_gather(uint32_t nl, BatchView B, const TabGather *__restrict__ items,
                                                     float4 *__restrict__ lr, float4 *__restrict__ sph,
                                                     uint32_t *__restrict__ lr_base_out, uint32_t *__restrict__ sph_base_out) {
    const TabGather g = items[blockIdx.x];
    const uint32_t *kl = B.kl_off + (size_t)g.s * (nl + 1);
    const uint32_t p0 = kl[0], plen = kl[nl] - p0;
    const bool fits = plen <= g.cap;
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        lr_base_out[g.s] = fits ? g.lr_base : 0xffffffffu;
        sph_base_out[g.s] = g.sph_base;
    }
    if (!fits) return;
    const float4 *rec = B.q_rec + B.s_off[g.s];
    const uint32_t grp = threadIdx.x >> 4, e = threadIdx.x & 15u;
    for (uint32_t li = blockIdx.y; li < nl; li += gridDim.y) {
        const uint32_t lo = kl[li] - p0, len = kl[li + 1] - kl[li];
        for (uint32_t c = grp; (c << 4) < len; c += blockDim.x >> 4) {
            const uint32_t k = (c << 4) + e;
            const bool v = k < len;
            float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
            if (v) {
                const size_t p = (size_t)p0 + lo + k;
                const uint32_t atom = (uint32_t)B.kl_pool[p] | (B.kl_pool_hi ? (uint32_t)B.kl_pool_hi[p] << 16 : 0u);
                r = rec[atom];
                lr[(size_t)g.lr_base + lo + k] = r;
            }
            float sx = r.x, sy = r.y, sz = r.z;  // (lanes beyond the list add zero)
            for (int o = 8; o > 0; o >>= 1) { sx += __shfl_xor(sx, o, 16); sy += __shfl_xor(sy, o, 16); sz += __shfl_xor(sz, o, 16); }
            const float inv = 1.f / (float)min(16u, len - (c << 4));
            const float cx = sx * inv, cy = sy * inv, cz = sz * inv;
            const float dx = r.x - cx, dy = r.y - cy, dz = r.z - cz;
            float d2 = v ? dx * dx + dy * dy + dz * dz : 0.f;
            for (int o = 8; o > 0; o >>= 1) d2 = fmaxf(d2, __shfl_xor(d2, o, 16));
            if (e == 0u) sph[(size_t)g.sph_base + (lo >> 4) + li + c] = make_float4(cx, cy, cz, __builtin_sqrtf(d2) * 1.0001f + 1.0e-3f);
        }
    }
}

// the records-through-L2 build of a work item's structure: a wide structure (24-bit kind-list entries) takes its own instantiation
template <class... Args>
__device__ __forceinline__ void tab_build_global(const BatchView &B, const TableDesc *__restrict__ tables, const TabWork w, Args... args) {
    if (B.s_off[w.s + 1] - B.s_off[w.s] > kNarrowAtoms) tab_build_tables<0, true>(B, tables, w, args...);
    else tab_build_tables<0, false>(B, tables, w, args...);
}

// Dynamic LDS: [float4 x lr_cap] lead records (the record of the atom at every position of the structure's kind
// lists), [u32 x (n_lead + 1)] kind-list offsets.  A structure whose kind lists exceed lr_cap reads through L2.
// kCompact: the compact layout below (atom_cap != 0) -- 80 KB, TWO 16-wave workgroups per CU, which takes eight waves
// per SIMD = at most 64 VGPRs (the attribute: left to itself the compiler settles on 67 and one workgroup per CU).
#ifndef TM_TAB_EU
#define TM_TAB_EU 8
#endif
// kLayout 0: a record per list position in LDS (one workgroup per CU); 1: the compact layout; 2: a batch of mostly large
// structures (complexes of tens of thousands of atoms): records through L2, 64 KB of LDS for the staged list B of the big
// tables -- two workgroups per CU as well, and the same eight waves per SIMD (as part of layout 0's kernel that path had
// 75 VGPRs: one workgroup per CU where the host had sized the LDS for two).
constexpr int kTabRecords = 0, kTabCompact = 1, kTabLarge = 2;
template <int kLayout>
__global__ __launch_bounds__(kTabBlock) __attribute__((amdgpu_waves_per_eu(kLayout == kTabRecords ? 4 : TM_TAB_EU, kLayout == kTabRecords ? 4 : TM_TAB_EU))) void k_tab_build(uint32_t n_lead, BatchView B, const TableDesc *__restrict__ tables,
                                                          const TabWork *__restrict__ work, uint32_t *__restrict__ tab_ok,
                                                          uint32_t *__restrict__ tab_rows, uint2 *__restrict__ tab_ent,
                                                          uint2 *__restrict__ tab_seg,
                                                          unsigned long long *__restrict__ cursors,
                                                          unsigned long long ent_capacity, uint32_t *__restrict__ overflow,
                                                          uint32_t lr_cap, unsigned long long shared_limit,
                                                          uint32_t atom_cap, unsigned long long derive_min, TabAux aux) {
    extern __shared__ __align__(16) unsigned char tab_smem[];
    __shared__ uint32_t s_count[kTabBlock / 64 + 1];  // (the last word: the next table of the workgroup's share)
    __shared__ unsigned long long s_first;
    const TabWork w = work[blockIdx.x];
    if (threadIdx.x == 0) s_count[kTabBlock / 64] = 0u;
    __syncthreads();
#ifdef TM_STATS
    const unsigned long long t_wg = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef TM_ABLATE
    if (threadIdx.x == 0 && blockIdx.x < kWgLog) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g_tab_log[blockIdx.x].start = __builtin_amdgcn_s_memrealtime();
        g_tab_log[blockIdx.x].hw_id = hw;
        g_tab_log[blockIdx.x].xcc_id = xcc;
    }
#endif
    // Bit 0 of tab_ok[s]: the structure's row slots fit (k_tab_layout; never changes while this kernel runs).  Bit 1 is
    // set by whichever workgroup of the structure runs out of entries -- at any moment, so nothing here branches on it:
    // a structure without row slots, or a wave whose entries do not fit, keeps sweeping in count-only mode, so that
    // cursors[1] ends on what a build with room would have handed out.
    const bool rows_ok = (tab_ok[w.s] & 1u) != 0u;
#ifdef TM_ABLATE
    const uint32_t dbg_mode = lr_cap >> 30;  // timing experiments: 1 no fill sweep, 2 no sweeps at all
    lr_cap &= 0x3fffffffu;
#else
    const uint32_t dbg_mode = 0;
#endif
    const uint32_t *kl = B.kl_off + (size_t)w.s * (n_lead + 1);
    const float4 *rec = B.q_rec + B.s_off[w.s];
    const uint32_t p0 = kl[0], plen = kl[n_lead] - p0;
    if constexpr (kLayout == kTabCompact) {
        // compact layout (host: the batch's largest structure leaves room for two workgroups per CU):
        // [float4 x atom_cap] atom records, [u16 x lr_cap] kind lists, [u32 x (n_lead + 1)] their offsets
        const uint32_t n = B.s_off[w.s + 1] - B.s_off[w.s];
        float4 *s_atoms = reinterpret_cast<float4 *>(tab_smem);
        uint16_t *s_pool = reinterpret_cast<uint16_t *>(s_atoms + atom_cap);
        uint32_t *s_kl2 = reinterpret_cast<uint32_t *>(s_pool + ((lr_cap + 7u) & ~7u));
        if (plen + 7u <= lr_cap && n <= atom_cap) {
            // all loads of the first pass before the first LDS store (one trip to L2, not one per array and iteration); the
            // kind lists in 16-byte words from the 16-byte boundary below their first entry, the offsets rebased to that
            const uint32_t rebase = p0 & ~7u, n_pool = (p0 - rebase + plen + 7u) / 8u;
            const uint32_t tid = threadIdx.x, step = blockDim.x;
            const uint4 *g_rec = reinterpret_cast<const uint4 *>(rec);
            const uint4 *g_pool = reinterpret_cast<const uint4 *>(B.kl_pool + rebase);
            uint4 *l_rec = reinterpret_cast<uint4 *>(s_atoms), *l_pool = reinterpret_cast<uint4 *>(s_pool);
            constexpr uint32_t kRecPer = 3;
            const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);  // (named registers, initialised: see k_search)
            uint4 v_rec0 = zero4, v_rec1 = zero4, v_rec2 = zero4, v_pool = zero4;
            uint32_t v_kl = 0;
            if (tid < n) v_rec0 = g_rec[tid];
            if (tid + step < n) v_rec1 = g_rec[tid + step];
            if (tid + 2u * step < n) v_rec2 = g_rec[tid + 2u * step];
            if (tid < n_pool) v_pool = g_pool[tid];
            if (tid <= n_lead) v_kl = kl[tid];
            if (tid < n) l_rec[tid] = v_rec0;
            if (tid + step < n) l_rec[tid + step] = v_rec1;
            if (tid + 2u * step < n) l_rec[tid + 2u * step] = v_rec2;
            if (tid < n_pool) l_pool[tid] = v_pool;
            if (tid <= n_lead) s_kl2[tid] = v_kl - rebase;
            for (uint32_t i = tid + kRecPer * step; i < n; i += step) l_rec[i] = g_rec[i];
            for (uint32_t i = tid + step; i < n_pool; i += step) l_pool[i] = g_pool[i];
            for (uint32_t i = tid + step; i <= n_lead; i += step) s_kl2[i] = kl[i] - rebase;
            __syncthreads();
#ifdef TM_STATS
            // [93] sum of the waves' working ticks, [94] (last wave's end - start of work) x waves, [95] staging ticks x waves
            __shared__ unsigned long long s_end;
            const unsigned long long t_work = __builtin_amdgcn_s_memrealtime();
            if (threadIdx.x == 0) s_end = 0ull;
            __syncthreads();
#endif
            tab_build_tables<2>(B, tables, w, s_atoms, s_pool, nullptr, s_kl2, tab_rows, tab_ent, tab_seg, cursors,
                                ent_capacity, overflow, tab_ok, dbg_mode, s_count, &s_first, n_lead, nullptr, 0u, shared_limit, rows_ok, derive_min);
#ifdef TM_ABLATE
            if ((threadIdx.x & 63) == 0 && blockIdx.x < kWgLog) {
                const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
                atomicMax(&g_tab_log[blockIdx.x].end, t_end);
                atomicMin(&g_tab_log[blockIdx.x].first_end, t_end);
                atomicAdd(&g_tab_log[blockIdx.x].sum_end, t_end);
            }
#endif
#ifdef TM_STATS
            {
                const unsigned long long t_done = __builtin_amdgcn_s_memrealtime();
                TM_STAT(93, t_done - t_work);
                if ((threadIdx.x & 63) == 0) atomicMax(&s_end, t_done);
                __syncthreads();
                if (threadIdx.x == 0) {
                    atomicAdd(&g_stats[94], (s_end - t_work) * (unsigned long long)(kTabBlock / 64));
                    atomicAdd(&g_stats[95], (t_work - t_wg) * (unsigned long long)(kTabBlock / 64));
                }
            }
#endif
        } else {
            tab_build_global(B, tables, w, rec, B.kl_pool, nullptr, kl, tab_rows, tab_ent, tab_seg, cursors,
                                ent_capacity, overflow, tab_ok, dbg_mode, s_count, &s_first, n_lead, s_atoms, atom_cap, shared_limit, rows_ok, derive_min);
        }
    } else if constexpr (kLayout == kTabLarge) {
        float4 *s_lr = reinterpret_cast<float4 *>(tab_smem);
        (void)plen;
        const uint32_t lrb = aux.lr_base ? aux.lr_base[w.s] : 0xffffffffu;
        if (lrb != 0xffffffffu)  // gathered (k_tab_gather): position records and chunk spheres in global memory
            tab_build_tables<3>(B, tables, w, rec, B.kl_pool, aux.lr + lrb, kl, tab_rows, tab_ent, tab_seg, cursors, ent_capacity,
                                overflow, tab_ok, dbg_mode, s_count, &s_first, n_lead, s_lr, lr_cap, shared_limit, rows_ok, derive_min,
                                aux.sph + aux.sph_base[w.s], p0);
        else
            tab_build_global(B, tables, w, rec, B.kl_pool, nullptr, kl, tab_rows, tab_ent, tab_seg, cursors,
                             ent_capacity, overflow, tab_ok, dbg_mode, s_count, &s_first, n_lead, s_lr, lr_cap, shared_limit, rows_ok, derive_min);
    } else {
        float4 *s_lr = reinterpret_cast<float4 *>(tab_smem);
        uint32_t *s_kl = reinterpret_cast<uint32_t *>(s_lr + lr_cap);
        if (plen <= lr_cap) {
            for (uint32_t i = threadIdx.x; i < plen; i += blockDim.x) s_lr[i] = rec[B.kl_pool[p0 + i]];
            for (uint32_t i = threadIdx.x; i <= n_lead; i += blockDim.x) s_kl[i] = kl[i] - p0;
            __syncthreads();
            tab_build_tables<1>(B, tables, w, rec, B.kl_pool, s_lr, s_kl, tab_rows, tab_ent, tab_seg, cursors,
                                ent_capacity, overflow, tab_ok, dbg_mode, s_count, &s_first, n_lead, nullptr, 0u, shared_limit, rows_ok, derive_min);
        } else {
            tab_build_global(B, tables, w, rec, B.kl_pool, nullptr, kl, tab_rows, tab_ent, tab_seg, cursors,
                                ent_capacity, overflow, tab_ok, dbg_mode, s_count, &s_first, n_lead, s_lr, lr_cap, shared_limit, rows_ok, derive_min);
        }
    }
}
#define TM_TAB_INSTANTIATE(layout)                                                                                                          \
    template __global__ void k_tab_build<layout>(uint32_t, BatchView, const TableDesc *, const TabWork *, uint32_t *, uint32_t *, uint2 *, \
                                                 uint2 *, unsigned long long *, unsigned long long, uint32_t *, uint32_t,                    \
                                                 unsigned long long, uint32_t, unsigned long long, TabAux);
TM_TAB_INSTANTIATE(kTabRecords)
TM_TAB_INSTANTIATE(kTabCompact)
TM_TAB_INSTANTIATE(kTabLarge)
#undef TM_TAB_INSTANTIATE

constexpr uint32_t kRefLenMask = (1u << 25) - 1u;
constexpr uint32_t kRefSingly = 1u << 31;
// Where every (structure, selected template) search finds its tables: {first entry of its pair table, number of
// entries (25 bits) | kRefSingly, first row slot of its triple table, first entry of its triple table}.  One 16-byte
// load per search, lane = template.  A template with an empty lead list in this structure cannot match: it gets an
// empty pair segment and costs nothing further.  kRefSingly marks the searches the block search hands to the
// template-at-a-time path: templates of fewer than three residues (enzymm's match_small_templates) and pair tables
// of 2^25 entries or more.
__global__ __launch_bounds__(256) void k_tab_refs(BatchView B, uint32_t n_lead, uint32_t n_sel,
                                                   const uint16_t *__restrict__ pair_table,
                                                   const uint16_t *__restrict__ triple_table,
                                                   const uint16_t *__restrict__ lead8, uint4 *__restrict__ tab_ref) {
    // dynamic LDS: one byte per lead kind -- does this structure have an atom of it? (sixteen dependent loads per thread
    // from the structure's offsets otherwise)
    extern __shared__ unsigned char s_has[];
    const uint32_t s = blockIdx.x;
    const uint32_t ti = blockIdx.y * blockDim.x + threadIdx.x;
    const uint32_t *kl = B.kl_off + (size_t)s * (n_lead + 1);
    for (uint32_t li = threadIdx.x; li < n_lead; li += blockDim.x) s_has[li] = kl[li + 1] != kl[li] ? 1 : 0;
    __syncthreads();
    if (ti >= n_sel) return;
    const size_t base = (size_t)s * B.n_tables;
    const uint32_t tp = pair_table[ti], tt = triple_table[ti];
    static_assert(kMaxG == 8, "a template's lead kinds are read as one 16-byte word");
    const uint4 leads = reinterpret_cast<const uint4 *>(lead8)[ti];
    const uint32_t lw[4] = {leads.x, leads.y, leads.z, leads.w};
    bool feasible = true;
#pragma unroll
    for (int g = 0; g < kMaxG; ++g) {
        const uint32_t li = (lw[g >> 1] >> ((g & 1) * 16)) & 0xffffu;
        if (li != 0xffffu && s_has[li] == 0) feasible = false;
    }
    uint4 r = make_uint4(0u, 0u, 0u, 0u);
    const bool wide = B.s_off[s + 1] - B.s_off[s] > kNarrowAtoms;  // (searched template at a time, whatever the template)
    if (B.tab_ok[s] == 1u && feasible) {
        if (tt == 0xffffu || wide) {
            r.y = kRefSingly;
        } else {
            const uint2 seg = B.tab_seg[base + tp], seg3 = B.tab_seg[base + tt];
            r.x = seg.x;
            r.y = seg.y - seg.x;
            if (r.y >= kRefLenMask) r.y = kRefSingly;
            r.z = B.tab_dir[base + tt];
            r.w = seg3.x;
        }
    }
    tab_ref[(size_t)s * n_sel + ti] = r;
}

// ------------------------------------------------------------------ k_search

// kMode of k_search: 0 the block search (stages A - D1, surviving triples to pages); 1 the template-at-a-time search of
// the templates the block search leaves out; 2 the depth-first search below the paged triples (k_deep).
constexpr int kModeBlocks = 0, kModeSingly = 1, kModeDeep = 2;

// Per-wave LDS: only what lanes exchange.  All search state lives in registers (see WaveRegs).  What a launch mode
// does not use is not reserved: the block search has no template constants in LDS, the depth-first part no queues.
template <int kMode>
struct __align__(16) WaveScratchT {
    static constexpr bool kSearches = kMode != kModeBlocks;  // runs the per-level search (dfs, eval_chunk, completion)
    static constexpr bool kQueues = kMode != kModeDeep;      // runs the table stages
    float2 bnd[kSearches ? kMaxPairs : 1];           // every pair bound of the current template (pair (i, j < i) at tri(i) + j)
    uint16_t surv[kSearches ? kMaxG : 1][64];        // per template residue: one 64-entry chunk of its lead list
    uint8_t surv_hi[kMode == 1 ? kMaxG : 1][64];     // the template-at-a-time launch on a wide structure: bits 16-23 of surv
    uint2 pairs[kQueues ? kPairQueue : 1];           // pair-table survivors {position in list 0 | position in list 1 << 16, slot}
    uint2 triples[kQueues ? kTripleQueue : 1];       // triple survivors {a0 | a1 << 16, a2 | slot << 16} waiting for the look at level 3
    uint32_t owner[64];                              // item-parallel stages: who owns the item at which its segment / row starts
    // candidate slots this wave has reserved: {next, end, size of the next reservation} (on_complete, slab_release)
    uint32_t slab[4];
};

// The block search also runs the depth-first search below the (few) triples its stage D0/D1 lets through, at the end of
// each block: by then the stage queues are empty, so the template's pair bounds and the lead-list chunks of that search
// use the same LDS.
#ifndef TM_SHARED_TRIPLES
#define TM_SHARED_TRIPLES 320
#endif
constexpr uint32_t kSharedTriples = TM_SHARED_TRIPLES;
#ifndef TM_MIN_BLOCK
#define TM_MIN_BLOCK 32
#endif
#ifndef TM_MIN_BLOCK_GLOBAL
#define TM_MIN_BLOCK_GLOBAL 2  // ... of a structure that is not in LDS (search_templates)
#endif
#ifndef TM_BLOCKS_GIVE_ABOVE
#define TM_BLOCKS_GIVE_ABOVE 1  // the block search (structure in LDS) hands over survivors of every level but the last
#endif
#ifndef TM_DEEP_GIVE_ABOVE
#define TM_DEEP_GIVE_ABOVE 3  // k_deep on a structure in global memory hands over survivors of levels below k - this (dfs)
#endif
constexpr uint32_t kMinBlock = TM_MIN_BLOCK;  // templates a wave claims at least (guided block sizes: 64 while much of the chunk is left)
#ifndef TM_RES_FIRST
#define TM_RES_FIRST 2
#endif
#ifndef TM_D0_UNROLL
#define TM_D0_UNROLL 1  // stage D0's loop over a residue's atoms: atoms looked at per trip (experiments: 2)
#endif
constexpr uint32_t kResFirst = TM_RES_FIRST;  // stage D0: levels looked at in its first round
// Table entries of TM_B_BATCH passes of stage B (TM_C_BATCH of stage C) are asked for TOGETHER: a pass is one dependent
// trip to L2 or HBM, four waves per SIMD do not hide it, and the block search is a chain of forty such trips per block of
// 64 templates (round 6).  The survivors of a group of passes are queued before any of them is worked off, so the queues
// hold 63 + 64 x batch entries -- inside the union below, whose other member (the depth-first part's bounds and lead
// chunks, 3232 bytes) is larger than both queues anyway.
#ifndef TM_B_BATCH
#define TM_B_BATCH 2
#endif
#ifndef TM_C_BATCH
#define TM_C_BATCH 2
#endif
#ifndef TM_C_RESTART
#define TM_C_RESTART 0
#endif
// With the structure in LDS (fewer than 8192 atoms: tm_api.hip) a queued pair is ONE word -- two 13-bit list positions and
// the block's 6-bit template slot -- so the same queue memory holds twice the pairs, and stage B asks for the entries of
// TM_B_BATCH_LDS passes at a time; a structure in global memory (list positions up to 65535) keeps 8-byte entries.
#ifndef TM_B_BATCH_LDS
#define TM_B_BATCH_LDS 4
#endif
constexpr uint32_t kBBatch = TM_B_BATCH, kBBatchLds = TM_B_BATCH_LDS, kCBatch = TM_C_BATCH;
static_assert(TM_B_BATCH_LDS + 1 <= 2 * (TM_B_BATCH + 1), "the packed pair queue holds twice the entries of the plain one");
constexpr uint32_t kPackedPos = 13, kPackedPosMask = (1u << kPackedPos) - 1u;
constexpr int kBlockPairQueue = 64 * (TM_B_BATCH + 1), kBlockTripleQueue = 64 * (TM_C_BATCH + 1);
template <>
struct __align__(16) WaveScratchT<kModeBlocks> {
    union {
        struct {
            uint2 pairs[kBlockPairQueue];
            uint2 triples[kBlockTripleQueue];
        };
        struct {
            float2 bnd[kMaxPairs];
            uint16_t surv[kMaxG][64];
        };
    };
    uint32_t owner[64];
    uint32_t slab[4];
};

// The workgroup's queue of triples that passed D0/D1 (in LDS, filled once: no wrap-around): any wave that comes to the end
// of a block takes triples from it and searches below them, so a block with many survivors is not its wave's burden
// alone.  What does not fit goes to the structure's pages in global memory (k_deep).
// An entry is a partial assignment: the target atoms of the first `depth` search levels (lead levels: at most kMaxG).
// The block search queues triples (depth 3); a wave deep in a large subtree hands the other survivors of a lead level
// to waves that have run out of work (dfs: give_away) -- one subtree of thousands of evaluations held its workgroup's
// CU for 0.25 ms after the other fifteen waves had finished (tests/gpu_wg_timeline.py).
struct SharedEntry {
    uint16_t atoms[kMaxK];
    uint32_t depth;
    uint32_t ti;  // staged template; 0xffffffff: reserved, not written yet
};
struct SharedTriples {
    SharedEntry entry[kSharedTriples];
    uint32_t head, tail;  // next to take; next to reserve (may run past kSharedTriples: those went to pages, or were not handed over)
    uint32_t busy;        // waves of the workgroup that have not run out of work
};

// Search state of one wave, held in VGPRs and indexed BY LANE: lane L holds the entry of search
// level L.  Uniform reads are v_readlane (SGPR result), uniform writes v_writelane: the DFS state
// machine touches no memory at all.
struct WaveRegs {
    // template constants per level (loaded once per template by lane < k)
    uint32_t pk, lead, resmask, nm_lo, nm_hi;  // pk: lvl_pack word
    float bnd_lo, bnd_hi;  // NOT by level: lane p holds the window of pair p (pair (i, j < i) = tri(i) + j), p < 64
    // per level: candidate source and cursor
    uint32_t lo, cnt;      // lead: pool offset/length of the kind list; else first atom/size of the residue
    uint32_t cur, base;    // next chunk to evaluate; chunk the pending mask refers to
    uint32_t pend_lo, pend_hi;
    // per level: assigned target atom
    uint32_t aidx, ameta, achain;
    float ax, ay, az;
    // per template residue (lane = group): which chunk of its lead list sits in WaveScratch::surv
    uint32_t surv_base;
};

__device__ __forceinline__ uint32_t RL(uint32_t v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float RLf(float v, int lane) {
    return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), lane));
}
// "write lane": a compare + select (this clang has no v_writelane builtin)
__device__ __forceinline__ void WL(uint32_t &v, int lane, uint32_t x) { v = lane_id() == lane ? x : v; }
__device__ __forceinline__ void WLf(float &v, int lane, float x) { v = lane_id() == lane ? x : v; }

// kIgnoreChain is a compile-time property of the launch: the default (ignore_chain=True in enzymm) drops
// every chain test from the instruction stream.
// A structure too large for LDS (a complex of tens of thousands of atoms) is searched with its records in global
// memory: every step of the depth-first part then waits for L2 or HBM two or three times over.  What that part touches
// below the lead levels is small, though -- the other atoms of the residues the leads were assigned to --, so k_deep
// keeps those residues in LDS (per wave: one slot per template residue, filled when a non-lead level of that residue
// is first prepared and valid for as long as the residue's lead stays the same atom).
constexpr uint32_t kResCacheAtoms = 32;
#ifndef TM_RUN_LDS
#define TM_RUN_LDS 4096
#endif
#ifndef TM_DEEP_SHARE
#define TM_DEEP_SHARE 8
#endif
constexpr uint32_t kDeepShare = TM_DEEP_SHARE;  // k_deep: triples per wave a structure must have for one more of its workgroups to take part
constexpr uint32_t kRunLds = TM_RUN_LDS;  // k_deep: runs of triples a workgroup's run list holds (a structure with more is claimed by page position)
static_assert((kDeepRun & (kDeepRun - 1u)) == 0u && kPageTriples % kDeepRun == 0u, "a run-list entry keeps its triple count in the low bits of an entry position");
struct __align__(16) ResCache {
    float4 rec[kMaxG][kResCacheAtoms];
    uint32_t tag[kMaxG];    // lead atom whose residue sits in the slot (0xffffffff: none)
    uint32_t first[kMaxG];  // first atom of that residue
    uint32_t count[kMaxG];  // its atoms
};

template <bool kIgnoreChain, bool kPoolLds, int kMode, bool kCache = false, bool kWideOk = false, bool kRecLds = false>
struct SearchCtx {
    static constexpr bool kResCache = kCache;
    // the structure's atom records are in LDS (a read past the end of the arrays returns zero instead of faulting: index
    // clamps that only keep a masked-out lane inside the arrays are left out)
    static constexpr bool kRecInLds = kRecLds;
    static constexpr int kLaunchMode = kMode;
    static constexpr bool kPoolInLds = kPoolLds;
    ResCache *rc;               // kResCache: this wave's slots
    const float4 *rec;          // structure atom records (LDS or global)
    const uint16_t *res_start;  // residue -> first atom (LDS or global)
    // kind lists of this structure in LDS (when they fit: kPoolLds) or the batch-wide pool in global memory (offsets
    // are rebased accordingly).  Which of the two is a compile-time property of the search body a workgroup runs: a
    // run-time choice per load either becomes ONE flat_load through a selected generic pointer (64-bit address
    // arithmetic per lane, both wait counters) or a branch per load, and both measured slower.
    const __attribute__((address_space(3))) uint16_t *pool_lds;
    const uint16_t *pool_glb;
    // a WIDE structure (more than kNarrowAtoms atoms; the template-at-a-time launch on global memory only): bits 16-23 of
    // its kind-list entries and of its residue table, NULL for every other structure and in every other launch mode
    const uint8_t *pool_hi;
    const uint8_t *res_hi;
    static constexpr bool kMayBeWide = kMode == 1 && !kPoolLds && kWideOk;  // (kModeSingly, records in global memory)
    __device__ __forceinline__ uint32_t pool_at(uint32_t i) const {
        if constexpr (kPoolLds) return pool_lds[i];
        else if constexpr (kMayBeWide) return pool_hi ? (uint32_t)pool_glb[i] | ((uint32_t)pool_hi[i] << 16) : (uint32_t)pool_glb[i];
        else return pool_glb[i];
    }
    __device__ __forceinline__ uint32_t res_at(uint32_t r) const {  // first atom of residue r
        if constexpr (kMayBeWide) return res_hi ? (uint32_t)res_start[r] | ((uint32_t)res_hi[r] << 16) : (uint32_t)res_start[r];
        else return res_start[r];
    }
    const uint8_t *qchain;      // global, per atom of this structure
    const uint32_t *tab_rows;   // pair tables: row offsets (global) ...
    const uint2 *tab_ent;       // ... and entries
    const float2 *bnd;          // this template's pair bounds (global)
    WaveScratchT<kMode> *S;
    int lane, k, n_groups;
    static constexpr bool ignore_chain = kIgnoreChain;
#ifdef TM_TIMES
    mutable unsigned long long tacc[13];
#endif
};

// Tests up to 64 candidates for `level`; returns the survivor mask.
//
// A lead level takes its candidates from a kind list: lane = candidate, and the constraints
// against the levels already assigned are tested one level at a time (their atoms are broadcast
// from the lane-indexed registers).
// Any other level takes the atoms of one target residue, of which the name/residue-type masks
// usually leave one or two: those are tested TRANSPOSED -- lane j holds assigned level j, its own
// bound (level, j) and its own group, the candidate is broadcast -- so the cost is per surviving
// candidate, not per assigned level.
template <class Ctx>
__device__ __forceinline__ unsigned long long eval_chunk(const Ctx &C, WaveRegs &W, int level, uint32_t base) {
    auto &S = *C.S;
    const int lane = C.lane;
    const uint32_t cnt = RL(W.cnt, level), lo = RL(W.lo, level);
    const bool lead = RL(W.lead, level) != 0;
    const uint32_t mygroup = lvl_group(RL(W.pk, level));
    // this level's row of pair bounds, one entry per lane j < level, loaded before the candidate gather
    float2 brow = make_float2(0.f, 0.f);
    if (lane < level) {  // (read through a pointer that carries the LDS address space)
        const unsigned long long raw = ((const __attribute__((address_space(3))) unsigned long long *)S.bnd)[tri(level) + lane];
        brow = make_float2(__uint_as_float((uint32_t)raw), __uint_as_float((uint32_t)(raw >> 32)));
    }
    const uint32_t idx = base + lane;
    bool ok = idx < cnt;
    uint32_t cand;
    if (lead) {
        if (RL(W.surv_base, (int)mygroup) != base) {  // the chunk is not the one cached in LDS
            const uint32_t entry = ok ? C.pool_at(lo + idx) : 0u;
            S.surv[mygroup][lane] = (uint16_t)entry;
            if constexpr (Ctx::kMayBeWide) S.surv_hi[mygroup][lane] = (uint8_t)(entry >> 16);
            WL(W.surv_base, (int)mygroup, base);
            wave_sync();
        }
        cand = S.surv[mygroup][lane];
        if constexpr (Ctx::kMayBeWide) cand |= (uint32_t)S.surv_hi[mygroup][lane] << 16;
    } else {
        cand = ok ? lo + idx : lo;
    }
    const float4 r = lead ? C.rec[cand] : residue_record(C, mygroup, cand);
    const uint32_t m = __float_as_uint(r.w);
    uint32_t cchain = 0, mychain = 0;
    if (!C.ignore_chain) {
        cchain = C.qchain[cand];
        mychain = lvl_tchain(RL(W.pk, level));
    }
    TM_STAT(level, 1);
    TM_STAT(lead ? 82 : 83, 1);
    unsigned long long alive;
    if (lead) {
        const uint32_t cres = meta_res(m);
        alive = ballot64(ok);
        for (int j = 0; j < level && alive; ++j) {
            const float dx = r.x - RLf(W.ax, j), dy = r.y - RLf(W.ay, j), dz = r.z - RLf(W.az, j);
            const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            const float blo = RLf(brow.x, j), bhi = RLf(brow.y, j);
            const bool same_t = lvl_group(RL(W.pk, j)) == mygroup;
            const bool same_q = meta_res(RL(W.ameta, j)) == cres;
            ok = ok & (d2 >= blo) & (d2 <= bhi) & (same_t == same_q) & (cand != RL(W.aidx, j));
            if (!C.ignore_chain) {
                const bool tc = lvl_tchain(RL(W.pk, j)) == mychain;
                const bool qc = RL(W.achain, j) == cchain;
                ok = ok & (tc == qc);
            }
            TM_STAT(50, 1);
            alive = ballot64(ok);
        }
    } else {
        const unsigned long long nm = ((unsigned long long)RL(W.nm_hi, level) << 32) | RL(W.nm_lo, level);
        const uint32_t rm = RL(W.resmask, level);
        const bool any = lvl_any(RL(W.pk, level)) != 0;
        ok = ok & (((nm >> meta_name(m)) & 1ull) != 0ull) & (any | (((rm >> meta_rtype(m)) & 1u) != 0u));
        alive = 0ull;
        const bool active = lane < level;
        const bool same_t = lvl_group(W.pk) == mygroup;
        const bool tc = lvl_tchain(W.pk) == mychain;
        for (unsigned long long todo = ballot64(ok); todo; todo &= todo - 1ull) {
            const int b = __ffsll((long long)todo) - 1;
            const float dx = RLf(r.x, b) - W.ax, dy = RLf(r.y, b) - W.ay, dz = RLf(r.z, b) - W.az;
            const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            const bool same_q = meta_res(W.ameta) == meta_res(RL(m, b));
            bool good = (d2 >= brow.x) & (d2 <= brow.y) & (same_t == same_q) & (W.aidx != lo + base + (uint32_t)b);
            if (!C.ignore_chain) good = good & (tc == (W.achain == RL(cchain, b)));
            TM_STAT(50, 1);
            if (ballot64(active & !good) == 0ull) alive |= 1ull << b;
        }
    }
    TM_STAT(24 + level, __popcll(alive));
    return alive;
}

// Prepares `level` after levels < level were assigned.
template <class Ctx>
__device__ __forceinline__ void init_level(const Ctx &C, WaveRegs &W, int level) {
    WL(W.cur, level, 0u);
    WL(W.pend_lo, level, 0u);
    WL(W.pend_hi, level, 0u);
    if (RL(W.lead, level) == 0) {
        // same template residue as an already placed lead atom: scan the atoms of that target residue
        const uint32_t pk = RL(W.pk, level);
        const int ref = (int)lvl_ref(pk);
        if constexpr (Ctx::kResCache) {
            const uint32_t g = lvl_group(pk), lead_atom = RL(W.aidx, ref);
            ResCache &R = *C.rc;
            if (uni(R.tag[g]) != lead_atom) {  // the slot holds another residue (or none): fetch this one
                const uint32_t res = meta_res(RL(W.ameta, ref));
                const uint32_t s0 = C.res_at(res), n = C.res_at(res + 1) - s0;
                wave_sync();  // (earlier reads of the slot are done)
                if ((uint32_t)C.lane < min(n, kResCacheAtoms)) R.rec[g][C.lane] = C.rec[s0 + (uint32_t)C.lane];
                if (C.lane == 0) { R.tag[g] = lead_atom; R.first[g] = s0; R.count[g] = n; }
                wave_sync();
            }
            WL(W.lo, level, uni(R.first[g]));
            WL(W.cnt, level, uni(R.count[g]));
        } else {
            const uint32_t res = meta_res(RL(W.ameta, ref));
            const uint32_t s0 = C.res_at(res), s1 = C.res_at(res + 1);  // uniform address: same value in every lane
            WL(W.lo, level, s0);
            WL(W.cnt, level, s1 - s0);
        }
    }
}

// The record of atom `atom`, which belongs to the target residue of template residue `g` (a non-lead level's candidate):
// from the residue cache when there is one and the residue fits a slot.
template <class Ctx>
__device__ __forceinline__ float4 residue_record(const Ctx &C, uint32_t g, uint32_t atom) {
    if constexpr (Ctx::kResCache) {
        const ResCache &R = *C.rc;
        const uint32_t first = uni(R.first[g]);
        if (uni(R.count[g]) <= kResCacheAtoms) return R.rec[g][min(atom - first, kResCacheAtoms - 1u)];
    }
    return C.rec[atom];
}

template <class Ctx>
__device__ __forceinline__ void assign_record(const Ctx &C, WaveRegs &W, int level, uint32_t cand, const float4 &r) {
    WL(W.aidx, level, cand);
    WLf(W.ax, level, r.x);
    WLf(W.ay, level, r.y);
    WLf(W.az, level, r.z);
    WL(W.ameta, level, __float_as_uint(r.w));
    if (!C.ignore_chain) WL(W.achain, level, (uint32_t)C.qchain[cand]);
}

template <class Ctx>
__device__ __forceinline__ void assign_level(const Ctx &C, WaveRegs &W, int level, uint32_t cand) {
    // uniform address: every lane receives the same record
    if (Ctx::kResCache && RL(W.lead, level) == 0u) assign_record(C, W, level, cand, residue_record(C, lvl_group(RL(W.pk, level)), cand));
    else assign_record(C, W, level, cand, C.rec[cand]);
}

// All arguments of k_search as one kernel-argument struct.  Code that runs rarely (a complete
// assignment, a pair with candidates) re-reads what it needs from the kernel-argument segment
// through cold_args() instead of keeping two dozen pointers in scalar registers for the whole
// search: the search loop is short of SGPRs, and every spilled SGPR costs VALU instructions.
// One (structure, round) work item; with the XCD class of the workgroup it names a template chunk.
struct WorkItem {
    uint32_t s;
    uint16_t round, rounds;
};

struct SearchArgs {
    LibView L;
    BatchView B;
    QueryView Q;
    OutView O;
    const WorkItem *work;      // (structure, round) items, heaviest structures first
    uint32_t n_items;
    uint32_t chunk_xcd_shift;  // template chunks of an item: (1 << chunk_xcd_shift) spread over the XCDs x item.rounds
    uint32_t max_atoms, max_res, pool_cap;
    // k_search<..., kSingly = true>: staged templates that may need the template-at-a-time search (those of fewer than
    // three residues; every template when a structure is large enough for a pair table of 2^25 entries)
    const uint32_t *singly;
    uint32_t n_singly;
    // block search with the structure in LDS: one counter per chunk class (1 << chunk_xcd_shift of them) -- the launch is
    // one workgroup per CU, every workgroup takes (work item, chunk of its class) units until none are left (nullptr: one
    // unit per workgroup, found from blockIdx)
    uint32_t *queue;
    uint32_t queue_once;  // experiments and tests: a workgroup leaves after one unit (the launch then has one workgroup per unit)
    uint32_t wide_only;   // 1: the launch searches the wide structures (template at a time, global memory); 0: every other structure
};
typedef const __attribute__((address_space(4))) SearchArgs *ColdArgsPtr;
__device__ __forceinline__ ColdArgsPtr cold_args() {
    ColdArgsPtr p = (ColdArgsPtr)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(p));  // opaque: loads through p stay where they are written
    return p;
}

// The (structure, template) pair a wave is searching.  How many candidates the pair has and where its chain of
// candidates ends is shared state (OutView::pair_count / pair_last), not the wave's: the triples of one pair may be
// searched by several waves at once (k_deep).
struct PairState {
    uint32_t s, ti;           // structure, staged-template index
    bool truncated;           // this wave met the pair's max_candidates limit
};

// A complete assignment passed the fp32 prefilter: re-check every pair in fp64
// (| |q_i - q_j| - |t_i - t_j| | <= thr_ij exactly as the oracle does) and append it to the
// candidate buffer.  Returns false when the search must stop (max_candidates reached).
template <class Ctx>
__device__ __forceinline__ bool on_complete(const Ctx &C, WaveRegs &W, PairState &P) {
    const int lane = C.lane, k = C.k, npairs = tri(k);
    // completions are rare: everything but the assignment itself is fetched here, not per search
    ColdArgsPtr A = cold_args();
    const uint32_t max_candidates = A->Q.max_candidates;
    const double2 *exact = A->Q.exact + (size_t)P.ti * kMaxPairs;  // (template distance, threshold) per pair, staged once
    const double *qx = A->B.q_xyz + 3 * (size_t)A->B.s_off[P.s];
    bool bad = false;
#if defined(TM_STATS) || defined(TM_ABLATE)
    if (A->Q.flags & 0x40000u) return true;  // phase timing: complete assignments dropped where they are found
#endif
    for (int p0 = 0; p0 < npairs; p0 += 64) {  // uniform trip count: the shuffles below need every lane
        const int p = p0 + lane;
        int i = 1, j = 0;
        if (p < npairs) pair_of(p, i, j);
        const uint32_t ai = (uint32_t)__shfl((int)W.aidx, i), aj = (uint32_t)__shfl((int)W.aidx, j);
        if (p < npairs) {
            const double *qa = qx + 3 * (size_t)ai, *qb = qx + 3 * (size_t)aj;
            const double dx = qa[0] - qb[0], dy = qa[1] - qb[1], dz = qa[2] - qb[2];
            const double dq = sqrt(dx * dx + dy * dy + dz * dz);
            const double2 e = exact[p];
            if (fabs(dq - e.x) > e.y) bad = true;
        }
    }
    TM_STAT(48, 1);
    if (ballot64(bad)) {  // fp32 prefilter false positive
        TM_STAT(51, 1);
        return true;
    }
#if defined(TM_STATS) || defined(TM_ABLATE)
    if (A->Q.flags & 0x80000u) return true;  // phase timing: complete assignments verified, not numbered or stored
#endif
    // The pair's candidate count: every verified assignment takes a number; those beyond max_candidates give it back
    // and mark the pair as cut short (so the count ends on min(assignments found, max_candidates) whatever the order
    // in which the waves working on the pair get here).
    const size_t pair = (size_t)P.s * A->Q.n_sel + A->Q.order[P.ti];
    uint32_t number = 0, first_mark = 0;
    if (lane == 0) {
        number = atomicAdd(&A->O.pair_count[pair], 1u) & 0x7fffffffu;
        if (number >= max_candidates) {
            atomicSub(&A->O.pair_count[pair], 1u);
            first_mark = (atomicOr(&A->O.pair_count[pair], 0x80000000u) >> 31) ^ 1u;
        }
    }
    number = uni(number);
    if (number >= max_candidates) {
        P.truncated = true;
        // a pair cut short before its first candidate (max_candidates == 0) is still reported
        if (lane == 0 && first_mark && max_candidates == 0u) {
            if (A->O.pair_min) { A->O.pair_min[pair] = ~0ull; A->O.pair_tie[pair] = 0u; }
            const unsigned long long e = atomicAdd(A->O.n_pairs, 1ull);
            if (e >= A->O.pair_capacity) atomicOr(A->O.overflow, 2u);
            else A->O.pairs[e] = PairEntry{P.s, P.ti};
        }
        return false;
    }
    // A slot of the candidate buffer: waves reserve slots in slabs (1, 2, 4 ... 64 at a time) -- one counter for every
    // candidate of the launch is what millions of completions (large complexes under loose thresholds) would queue at.
    // What a wave leaves unused of its last slab is marked invalid when the wave is done (slab_release).
    auto &S = *C.S;
    wave_sync();  // (lane 0 keeps the wave's slab cursor in LDS)
    uint32_t next = uni(S.slab[0]);
    const uint32_t end = uni(S.slab[1]);
    if (next == end) {
        const uint32_t want = uni(S.slab[2]);
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(A->O.n_cands, (unsigned long long)want);
        base = uni64(base);
        if (base + want > A->O.cand_capacity) {
            // (the query is run again with room: tm_result_needed.  From here on the wave asks for ONE slot per candidate, so
            // that the counter ends on what the query needs, not on up to 64 times that)
            if (lane == 0) { atomicOr(A->O.overflow, 1u); S.slab[2] = 1u; }
            // (the slots between the counter's old value and the end of the buffer are nobody's now: marked invalid like the
            // unused rest of a slab -- k_superpose reads every slot below the capacity, whatever memory it was handed)
            if (base + (unsigned long long)lane < A->O.cand_capacity) A->O.cands[base + (unsigned long long)lane].s = 0xffffffffu;
            return true;
        }
        next = (uint32_t)base;
        if (lane == 0) { S.slab[1] = next + want; S.slab[2] = min(64u, 2u * want); }
    }
    const unsigned long long slot = next;
    if (lane == 0) S.slab[0] = next + 1u;
    wave_sync();
    Cand *c = A->O.cands + slot;
    if (lane == 0) {
        // the candidate becomes the new end of the pair's chain; whoever links the first one lists the pair
        const uint32_t before = atomicExch(&A->O.pair_last[pair], (uint32_t)slot + 1u);
        c->s = P.s;
        c->ti = P.ti;
        c->prev = before - 1u;  // 0 -> 0xffffffff: none
        c->tie_next = 0;
        if (before == 0u) {
            if (A->O.pair_min) { A->O.pair_min[pair] = ~0ull; A->O.pair_tie[pair] = 0u; }  // (read by later kernels only)
            const unsigned long long e = atomicAdd(A->O.n_pairs, 1ull);
            if (e >= A->O.pair_capacity) atomicOr(A->O.overflow, 2u);
            else A->O.pairs[e] = PairEntry{P.s, P.ti};
        }
    }
    if (lane < kMaxK) {
        if (lane < k) c->atoms[lvl_orig(W.pk)] = (uint16_t)W.aidx;
        else c->atoms[lane] = 0xffffu;
        if constexpr (Ctx::kMayBeWide) {  // (a batch with a wide structure: bits 16-23 of every candidate's atoms)
            uint8_t *hi = A->O.cands_hi;
            if (hi) hi[slot * kMaxK + (lane < k ? lvl_orig(W.pk) : (uint32_t)lane)] = lane < k ? (uint8_t)(W.aidx >> 16) : (uint8_t)0;
        }
    }
    return true;
}

// A wave starts without reserved candidate slots, and marks what it leaves unused invalid when it is done.
template <class Ctx>
__device__ __forceinline__ void slab_init(const Ctx &C) {
    if (C.lane == 0) { C.S->slab[0] = 0u; C.S->slab[1] = 0u; C.S->slab[2] = 1u; }
    wave_sync();
}
template <class Ctx>
__device__ __forceinline__ void slab_release(const Ctx &C) {
    wave_sync();
    const uint32_t next = uni(C.S->slab[0]), end = uni(C.S->slab[1]);
    Cand *cands = cold_args()->O.cands;
    for (uint32_t i = next + (uint32_t)C.lane; i < end; i += 64u) cands[i].s = 0xffffffffu;
}

// Depth-first search over levels >= floor (levels < floor are already assigned).
// Returns false when the pair was cut short by max_candidates.
// Survivors `give` of a chunk of lead level `level` (levels below it assigned): queued as partial assignments of
// level + 1 atoms for the waves that wait for work.  Returns what stays with this wave.
template <class Ctx>
__device__ __forceinline__ unsigned long long give_away(const Ctx &C, const WaveRegs &W, const PairState &P, int level,
                                                        unsigned long long pend, unsigned long long give, uint32_t chunk_base,
                                                        SharedTriples &Q2) {
    const uint32_t lane = (uint32_t)C.lane;
    const uint32_t n = (uint32_t)__popcll(give);
    uint32_t base = 0;
    if (lane == 0u) base = atomicAdd(&Q2.tail, n);
    base = uni(base);
    // (what fits: written as block_look writes it -- `base < capacity ? min(n, capacity - base) : 0` came out of the compiler as
    // min(n, capacity - base) without the guard, and a reservation past the end of the queue, which happens as soon as the
    // queue is full, sent every lane's entry to the last slot: subtrees lost -- test_queued_block_search_and_subtree_handover)
    const uint32_t r = rank_in(give);
    const bool here = ((give >> lane) & 1ull) != 0ull && base + r < kSharedTriples;
    // (one 16-bit store per level, the value a scalar: this runs once per large subtree and must not cost the search
    // loop around it a register)
    SharedEntry *e = &Q2.entry[min(base + r, kSharedTriples - 1u)];
    for (int j = 0; j < level; ++j) {
        const uint32_t a = RL(W.aidx, j);
        if (here) e->atoms[j] = (uint16_t)a;
    }
    if (here) {
        // (a lead level: the chunk's candidates sit in LDS; any other: lane b is atom b of the chunk of its residue)
        e->atoms[level] = RL(W.lead, level) != 0u ? C.S->surv[lvl_group(RL(W.pk, level))][lane]
                                                  : (uint16_t)(RL(W.lo, level) + chunk_base + lane);
        e->depth = (uint32_t)level + 1u;
#ifdef TM_STATS
        {  // [300]: entries whose atoms do not add up to what their writer stored with them (a publication race would show here)
            uint32_t sum = 0;
            for (int j = 0; j < level; ++j) sum += RL(W.aidx, j) & 0xffffu;
            sum += e->atoms[level];
            e->depth |= sum << 8;
        }
#endif
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __hip_atomic_store(&e->ti, P.ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
#ifdef TM_STATS
    if (here) atomicAdd(&g_stats[256 + level + 1], 1ull);
#endif
    return pend & ~ballot64(here);
}

template <class Ctx>
__device__ __forceinline__ bool dfs(const Ctx &C, WaveRegs &W, PairState &P, int floor, SharedTriples *share = nullptr) {
    const int k = C.k;
    // every atom a lead and at most three of them (a template of one atom per residue): the tables placed them all
    if (floor >= k) return on_complete(C, W, P);
    int level = floor;
    init_level(C, W, level);
    for (;;) {
        unsigned long long pend = ((unsigned long long)RL(W.pend_hi, level) << 32) | RL(W.pend_lo, level);
        if (pend == 0ull) {
            // evaluate further chunks of this level until one has survivors
            uint32_t cur = RL(W.cur, level);
            const uint32_t cnt = RL(W.cnt, level);
            uint32_t base = 0;
            while (cur < cnt) {
                pend = eval_chunk(C, W, level, cur);
                base = cur;
                cur += 64;
                if (pend) break;
            }
            if (pend == 0ull) {  // level exhausted: backtrack
                if (level == floor) return true;
                --level;
                continue;
            }
            WL(W.cur, level, cur);
            WL(W.base, level, base);
            if constexpr (Ctx::kLaunchMode == kModeBlocks || Ctx::kResCache) {
                // survivors while waves of the workgroup wait for work: they take all but one -- all of them when this level
                // has further chunks to look at (a long kind list with a survivor here and there: the subtrees are searched
                // while the list is still being read).  At ANY level: the subtrees that held a CU for 0.25 ms after fifteen
                // waves had finished fan out below the lead levels -- two interchangeable atoms per residue (OD1/OD2,
                // NH1/NH2, ring flips), eight residues
                const unsigned long long give = cur < cnt ? pend : pend & (pend - 1ull);
                if (share != nullptr && give != 0ull && level < k - (Ctx::kResCache ? TM_DEEP_GIVE_ABOVE : TM_BLOCKS_GIVE_ABOVE)) {
                    const uint32_t busy = uni(__hip_atomic_load(&share->busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    const uint32_t tail = uni(__hip_atomic_load(&share->tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                    // (test hooks, ENZYMM_AMD_HANDOVER: 0 = never; 2 = always, whether or not a wave waits -- every queue
                    // entry of a small selection then goes through the hand-over)
                    const uint32_t hook = cold_args()->Q.flags;
                    if ((busy < (uint32_t)kSearchWaves || (hook & 0x20000u)) && tail < kSharedTriples && !(hook & 0x10000u))
                        pend = give_away(C, W, P, level, pend, (hook & 0x20000u) ? pend : give, base, *share);
                }
                if (pend == 0ull) continue;  // all handed over: on with the level's next chunk (its pending mask is still empty)
            }
        }
        const int b = __ffsll((long long)pend) - 1;
        pend &= pend - 1ull;
        WL(W.pend_lo, level, (uint32_t)pend);
        WL(W.pend_hi, level, (uint32_t)(pend >> 32));
        uint32_t cand;
        if (RL(W.lead, level)) {
            cand = uni((uint32_t)C.S->surv[lvl_group(RL(W.pk, level))][b]);
            if constexpr (Ctx::kMayBeWide) cand |= uni((uint32_t)C.S->surv_hi[lvl_group(RL(W.pk, level))][b]) << 16;
        } else cand = RL(W.lo, level) + RL(W.base, level) + (uint32_t)b;
        assign_level(C, W, level, cand);
        if (level < k - 1) {
            ++level;
            init_level(C, W, level);
            continue;
        }
        if (!on_complete(C, W, P)) return false;
    }
}

// fp32 prefilter of one pair constraint; the bounds carry kPrefilterSlack, which also covers the
// fused multiply-adds (the exact test is repeated in fp64 before a candidate is emitted)
__device__ __forceinline__ bool in_window(const float4 &p, const float4 &q, const float2 &b) {
    const float dx = p.x - q.x, dy = p.y - q.y, dz = p.z - q.z;
    const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
    return (d2 >= b.x) & (d2 <= b.y);
}

// Quick look at level 3 for a triple that survived the tables, before the general search machinery is set up for
// it: is there ANY candidate for the template's fourth search position that fits the three assigned atoms?  Four in
// five triples end here.  The three atoms are wave-uniform values, so a candidate costs three plain window tests.
// The test is the one eval_chunk makes for level 3, minus the chain rule (a superset is all a filter needs; the
// search proper repeats every test).
template <class Ctx>
__device__ __forceinline__ bool level3_has_candidate(const Ctx &C, const WaveRegs &W, uint32_t x0, uint32_t x1,
                                                     uint32_t x2, const float4 &q0, const float4 &q1, const float4 &q2) {
    const uint32_t lane = (uint32_t)C.lane;
    // windows (3, 0), (3, 1), (3, 2): lanes 3, 4, 5 of the per-pair registers (no LDS round trip)
    const float2 w0 = make_float2(RLf(W.bnd_lo, 3), RLf(W.bnd_hi, 3)), w1 = make_float2(RLf(W.bnd_lo, 4), RLf(W.bnd_hi, 4)),
                 w2 = make_float2(RLf(W.bnd_lo, 5), RLf(W.bnd_hi, 5));
    const uint32_t res0 = meta_res(__float_as_uint(q0.w)), res1 = meta_res(__float_as_uint(q1.w)),
                   res2 = meta_res(__float_as_uint(q2.w));
    if (RL(W.lead, 3) != 0u) {  // the lead atom of a fourth residue: its kind list, 64 entries at a time
        const uint32_t lo = RL(W.lo, 3), cnt = RL(W.cnt, 3);
        for (uint32_t base = 0; base < cnt; base += 64u) {
            const uint32_t idx = base + lane;
            const float4 r = C.rec[C.pool_at(lo + min(idx, cnt - 1u))];
            const uint32_t cres = meta_res(__float_as_uint(r.w));
            const bool ok = (idx < cnt) & in_window(r, q0, w0) & in_window(r, q1, w1) & in_window(r, q2, w2) &
                            (cres != res0) & (cres != res1) & (cres != res2);
            if (ballot64(ok) != 0ull) return true;
        }
        return false;
    }
    // another atom of an assigned residue: the atoms of that target residue (at most a few dozen)
    const uint32_t ref = lvl_ref(RL(W.pk, 3));
    const uint32_t res = ref == 0u ? res0 : ref == 1u ? res1 : res2, own = ref == 0u ? x0 : ref == 1u ? x1 : x2;
    const uint32_t s0 = C.res_at(res), s1 = C.res_at(res + 1);
    const unsigned long long nm = ((unsigned long long)RL(W.nm_hi, 3) << 32) | RL(W.nm_lo, 3);
    const uint32_t rm = RL(W.resmask, 3);
    const bool any = lvl_any(RL(W.pk, 3)) != 0u;
    for (uint32_t base = s0; base < s1; base += 64u) {
        const uint32_t idx = base + lane;
        const float4 r = C.rec[min(idx, s1 - 1u)];
        const uint32_t m = __float_as_uint(r.w);
        const bool ok = (idx < s1) & (idx != own) & (((nm >> meta_name(m)) & 1ull) != 0ull) &
                        (any | (((rm >> meta_rtype(m)) & 1u) != 0u)) & in_window(r, q0, w0) & in_window(r, q1, w1) &
                        in_window(r, q2, w2);
        if (ballot64(ok) != 0ull) return true;
    }
    return false;
}

// The pair tables a (structure, template) search reads: segment of the (lead 0, lead 1) table, first row slot of
// the (lead 0 or 1, lead 2) table.  Uniform values, fetched when the template is claimed.
struct TableRefs {
    uint32_t pair_begin, pair_end;  // entries of the pair table
    uint32_t triple_rows;           // first row slot of the triple table (row offsets are relative to ...
    uint32_t triple_begin;          // ... the table's first entry)
    uint32_t key_is_1;              // rows of the triple table are positions of lead 1 (else lead 0)
    uint32_t flags;                 // diagnostic builds: the query flags (phase ablation)
};

// What a search fetches from the tables while the template is being set up, one value per lane: the first 64
// entries of the pair table, the first 64 row slots and the first 64 entries of the triple table.  Most tables are
// smaller than that, and the triple stage then runs without a dependent global load.
struct TablePrefetch {
    uint2 pair_ent;
    uint32_t triple_row;
    uint2 triple_ent;
};

// The lead atoms of the first three template residues come from the structure's pair tables (k_tab_*):
//
//   pair stage    the (lead 0, lead 1) table's segment is scanned 64 entries at a time; an entry survives when its
//                 stored squared distance lies in the template's window (1, 0).  Survivors are queued in LDS.
//   triple stage  every queued pair names a row of the (key lead, lead 2) table -- the atoms of list 2 near the key
//                 atom.  The rows of up to 64 queued pairs are laid end to end over the lanes (prefix sum of the row
//                 lengths, binary search for the owning pair), each item is tested against the windows (2, key)
//                 (stored distance) and (2, other) (computed), and what survives is searched depth-first.
//
// Different residues are guaranteed by the tables for (0, 1) and (key, 2); (other, 2) is tested here.  Templates with
// two residues search below the pairs, templates with one residue below every atom of their lead list.
template <class Ctx>
__device__ __forceinline__ bool table_scan(const Ctx &C, WaveRegs &W, PairState &P, const TableRefs &TR,
                                           const TablePrefetch &F) {
    auto &S = *C.S;
    // (the template-at-a-time path keeps its own layout of the two queues inside the block search's)
    uint32_t *queue = reinterpret_cast<uint32_t *>(S.pairs);
    uint32_t (*queue3)[3] = reinterpret_cast<uint32_t (*)[3]>(S.triples);  // (64 triples x 12 B: the atoms in full words -- a wide structure's do not fit 16 bits)
    const int lane = C.lane;
    const int G = C.n_groups;
    const uint32_t lo0 = RL(W.lo, 0), lo1 = RL(W.lo, 1), lo2 = RL(W.lo, 2);
    if (G == 1) {
        const uint32_t n0 = RL(W.cnt, 0);
        for (uint32_t i = 0; i < n0; ++i) {
            assign_level(C, W, 0, uni((uint32_t)C.pool_at(lo0 + i)));
            if (!dfs(C, W, P, 1)) return false;
        }
        return true;
    }
    const float2 b10 = make_float2(RLf(W.bnd_lo, 0), RLf(W.bnd_hi, 0));  // window (1, 0): pair 0
    const float2 b20 = make_float2(RLf(W.bnd_lo, 1), RLf(W.bnd_hi, 1)), b21 = make_float2(RLf(W.bnd_lo, 2), RLf(W.bnd_hi, 2));
    const bool key1 = TR.key_is_1 != 0u;
    const float2 b_key = key1 ? b21 : b20, b_other = key1 ? b20 : b21;
    const bool tc10 = lvl_tchain(RL(W.pk, 0)) == lvl_tchain(RL(W.pk, 1));
    const bool tc20 = lvl_tchain(RL(W.pk, 0)) == lvl_tchain(RL(W.pk, 2)), tc21 = lvl_tchain(RL(W.pk, 1)) == lvl_tchain(RL(W.pk, 2));

    uint32_t e = TR.pair_begin;
    const uint32_t e_end = TR.pair_end;
    bool have_first = true;  // F.pair_ent holds entry e + lane of the first round (fetched when the template was set up)
    while (e < e_end) {
        // ---- pair stage: scan until the queue may not take another 64 entries (or the segment ends)
        uint32_t nq = 0;
        wave_sync();  // the queue of the previous round has been consumed
        do {
            const uint32_t idx = e + (uint32_t)lane;
            bool ok = idx < e_end;
            uint2 ent = F.pair_ent;
            if (!have_first) {
                ent = make_uint2(0u, 0u);
                if (ok) ent = C.tab_ent[idx];
            }
            have_first = false;
            const float d2 = __uint_as_float(ent.y);
            ok = ok & (d2 >= b10.x) & (d2 <= b10.y);
            if (!C.ignore_chain) {
                if (ok) ok = (C.qchain[C.pool_at(lo0 + (ent.x & 0xffffu))] == C.qchain[C.pool_at(lo1 + (ent.x >> 16))]) == tc10;
            }
            const unsigned long long m = ballot64(ok);
            TM_STAT(52, 1);
            if (m) {
                if (ok) queue[nq + rank_in(m)] = ent.x;
                nq += (uint32_t)__popcll(m);
            }
            e += 64u;
        } while (e < e_end && nq + 64u <= (uint32_t)kPairQueue);
        TM_STAT(53, nq);
        TM_STAT(54, nq != 0u ? 1 : 0);                      // pair-stage rounds with a survivor ...
        TM_STAT(58, e_end - TR.pair_begin <= 64u ? 1 : 0);  // ... and rounds of segments that fit one tile
        TM_STAT(59, (e_end - TR.pair_begin <= 64u && nq == 0u) ? 1 : 0);
#if defined(TM_STATS) || defined(TM_ABLATE)
        if (TR.flags & 0x100u) nq = 0;
#endif
        if (nq == 0u) continue;
        wave_sync();
        if (G == 2) {
            for (uint32_t q = 0; q < nq; ++q) {
                const uint32_t pos = uni(queue[q]);
                const uint32_t a0 = C.pool_at(lo0 + (pos & 0xffffu)), a1 = C.pool_at(lo1 + (pos >> 16));
                const float4 r0 = C.rec[a0], r1 = C.rec[a1];
                assign_record(C, W, 0, a0, r0);
                assign_record(C, W, 1, a1, r1);
                if (!dfs(C, W, P, 2)) return false;
            }
            continue;
        }
        // ---- triple stage, up to 64 queued pairs at a time
        for (uint32_t p0 = 0; p0 < nq; p0 += 64u) {
            const uint32_t np = min(64u, nq - p0);
            const bool vp = (uint32_t)lane < np;
            const uint32_t pos_p = vp ? queue[p0 + (uint32_t)lane] : 0u;
            const uint32_t key = key1 ? pos_p >> 16 : pos_p & 0xffffu;
            // the row of the key atom: offsets from the prefetched slots (lanes key, key + 1), or from memory
            uint32_t r0 = lane_get(F.triple_row, key), r1 = lane_get(F.triple_row, key + 1u);
            if (vp && key >= 63u) {
                r0 = C.tab_rows[TR.triple_rows + key];
                r1 = C.tab_rows[TR.triple_rows + key + 1u];
            }
            const uint32_t len = vp ? r1 - r0 : 0u;
            const uint32_t incl = wave_scan_add(len);  // inclusive prefix sum of the row lengths over the lanes
            const uint32_t excl = incl - len;
            const uint32_t total = RL(incl, 63);
            const uint32_t row_base = r0 - excl;  // entry of item w of this pair's row: row_base + w (mod 2^32)
            for (uint32_t w0 = 0; w0 < total; w0 += 64u) {
                // lanes past the last item repeat it (every address below stays valid) and are masked out of the vote
                const uint32_t w = min(w0 + (uint32_t)lane, total - 1u);
                // Owner of item w: the pair whose row holds it = the number of rows that end at or before w.
                // Owner of item w: each pair writes its lane number where its row starts (or at item w0, when the row
                // started in an earlier pass); a running maximum spreads it.  (Counting row ends with readlane instead,
                // for few pairs, measured 2-4 % slower.)
                S.owner[lane] = 0u;
                {
                    const uint32_t start = max(excl, w0);
                    if (len != 0u && start < w0 + 64u && incl > w0) S.owner[start - w0] = (uint32_t)lane;
                }
                wave_sync();
                const uint32_t p = wave_scan_max(S.owner[lane]);
                wave_sync();
                bool ok = w0 + (uint32_t)lane < total;
                const uint32_t e_rel = lane_get(row_base, p) + w;  // entry, relative to the table's first
                const uint32_t pos = lane_get(pos_p, p);
                uint2 ent = make_uint2(lane_get(F.triple_ent.x, e_rel & 63u), lane_get(F.triple_ent.y, e_rel & 63u));
                if (e_rel >= 64u) ent = C.tab_ent[TR.triple_begin + e_rel];
                const float dk2 = __uint_as_float(ent.y);
                ok = ok & (dk2 >= b_key.x) & (dk2 <= b_key.y);
                const uint32_t a0 = C.pool_at(lo0 + (pos & 0xffffu)), a1 = C.pool_at(lo1 + (pos >> 16));
                const uint32_t c2 = C.pool_at(lo2 + (ent.x >> 16));
                const float4 rc = C.rec[c2], ro = C.rec[key1 ? a0 : a1];
                ok = ok & in_window(rc, ro, b_other) & (meta_res(__float_as_uint(rc.w)) != meta_res(__float_as_uint(ro.w)));
                if (!C.ignore_chain) {
                    if (ok) {
                        const uint8_t qc = C.qchain[c2];
                        ok = ((qc == C.qchain[a0]) == tc20) & ((qc == C.qchain[a1]) == tc21);
                    }
                }
                unsigned long long m = ballot64(ok);
                TM_STAT(2, 1);
                TM_STAT(24 + 2, __popcll(m));
#if defined(TM_STATS) || defined(TM_ABLATE)
                if (TR.flags & 0x400u) m = 0ull;
#endif
                if (m == 0ull) continue;
                if (ok) {
                    const uint32_t slot = rank_in(m);
                    queue3[slot][0] = a0;
                    queue3[slot][1] = a1;
                    queue3[slot][2] = c2;
                }
                const uint32_t nq3 = (uint32_t)__popcll(m);
                wave_sync();
                for (uint32_t q = 0; q < nq3; ++q) {
                    const uint32_t x0 = uni(queue3[q][0]), x1 = uni(queue3[q][1]), c = uni(queue3[q][2]);
                    const float4 q0 = C.rec[x0], q1 = C.rec[x1], q2 = C.rec[c];
                    if (C.k > 3 && !level3_has_candidate(C, W, x0, x1, c, q0, q1, q2)) continue;
#if defined(TM_STATS) || defined(TM_ABLATE)
                    if (TR.flags & 0x800u) continue;  // phase timing: the look at level 3, but no search below it
#endif
                    assign_record(C, W, 0, x0, q0);
                    assign_record(C, W, 1, x1, q1);
                    assign_record(C, W, 2, c, q2);
                    if (!dfs(C, W, P, 3)) return false;
                }
                wave_sync();
            }
        }
    }
    return true;
}

__host__ __device__ inline size_t search_res_bytes(uint32_t max_res) {
    return (((size_t)max_res + 1) * sizeof(uint16_t) + 15) & ~size_t(15);
}
__host__ __device__ inline size_t search_kl_words(uint32_t n_lead) { return ((size_t)n_lead + 1 + 3) & ~size_t(3); }

// One template's per-level constants as loaded by lane L for level L (plus one lead-pair bound): what the
// depth-first part of a search needs.  Loaded only for templates with a triple that survived the tables.
struct TemplateRegs {
    uint32_t pk, resmask, nm_lo, nm_hi, li;
    float2 bound;  // bounds[lane]
    uint32_t hdr;  // k | n_groups << 8 | triple_key << 16
};

__device__ __forceinline__ void load_template(const QueryView &Q, uint32_t ti, int lane, TemplateRegs &R) {
    const Staged &T = Q.staged[ti];
    int l = lane < kMaxK ? lane : kMaxK - 1;  // every lane loads: no dependence on k
    // (opaque to the optimiser: the per-lane offsets below are worked out here, per template -- hoisted out of the queued
    // block search's loop over units they stayed in registers for the whole kernel, and nine of those went to scratch memory)
    asm volatile("" : "+v"(l), "+v"(lane));
    const unsigned long long nm = T.namemask[l];
    R.pk = T.packed[l];
    R.resmask = T.resmask[l];
    R.nm_lo = (uint32_t)nm;
    R.nm_hi = (uint32_t)(nm >> 32);
    R.li = T.lead_li[l];
    R.bound = Q.bounds[(size_t)ti * kMaxPairs + lane];
    R.hdr = T.hdr;
}

// Lane L < k takes level L's constants of the template in R: the search state machine (dfs, eval_chunk,
// level3_has_candidate) runs on W.  Returns false when a lead list of the template is empty in this structure.
template <class Ctx>
__device__ __forceinline__ bool setup_template(Ctx &C, WaveRegs &W, const TemplateRegs &R, const uint32_t *kl,
                                               const QueryView &Q, uint32_t ti) {
    const int lane = C.lane;
    const uint32_t hdr = uni(R.hdr);
    const int k = (int)(hdr & 0xffu);
    C.k = k;
    C.n_groups = (int)((hdr >> 8) & 0xffu);
    C.bnd = Q.bounds + (size_t)ti * kMaxPairs;
    W = {};
    W.surv_base = 0xffffffffu;
    bool empty = false;
    if (lane < k) {
        W.pk = R.pk;
        W.resmask = R.resmask;
        W.nm_lo = R.nm_lo;
        W.nm_hi = R.nm_hi;
        W.lead = R.li != 0xffffu ? 1u : 0u;
        if (W.lead) {
            const uint32_t o0 = kl[R.li], o1 = kl[R.li + 1];
            W.lo = o0;
            W.cnt = o1 - o0;
            empty = o1 == o0;
        }
    }
    // every pair bound of the template goes to LDS (the first 64 came with the template's constants, the rest is fetched
    // now); the first windows are also read from the registers
    C.S->bnd[lane] = R.bound;
    int p_first = 64 + lane;
    asm volatile("" : "+v"(p_first));  // (as in load_template: the address of the lane's first bound is not kept across units)
    for (int p = p_first; p < tri(k); p += 64) C.S->bnd[p] = C.bnd[p];
    W.bnd_lo = R.bound.x;
    W.bnd_hi = R.bound.y;
    wave_sync();
    return ballot64(empty) == 0ull;
}

// Template-at-a-time search (the round-2 path): templates of one or two residues, which have no triple table, and
// pair tables too long for a block's 25-bit segment length.
template <class Ctx>
__device__ void search_singly(Ctx &C, const SearchArgs &A, WaveRegs &W, uint32_t s, uint32_t ti, const uint32_t *kl) {
    const QueryView &Q = A.Q;
    const BatchView &B = A.B;
    const int lane = C.lane;
    TemplateRegs R;
    load_template(Q, ti, lane, R);
    const bool feasible = setup_template(C, W, R, kl, Q, ti);
    const uint32_t hdr = uni(R.hdr);
    const int G = C.n_groups;
    TableRefs TR = {};
    TablePrefetch F = {};
    const Staged &T = Q.staged[ti];
    const uint32_t tp = T.pair_table, tt = T.triple_table;
    uint32_t pair_len = 0, triple_len = 0;
    if (tp != 0xffffu) {
        const uint2 seg = B.tab_seg[(size_t)s * B.n_tables + tp];
        TR.pair_begin = uni(seg.x);
        pair_len = uni(seg.y - seg.x);
    }
    TR.pair_end = TR.pair_begin + pair_len;
    if (tt != 0xffffu) {
        const uint2 seg = B.tab_seg[(size_t)s * B.n_tables + tt];
        TR.triple_rows = uni(B.tab_dir[(size_t)s * B.n_tables + tt]);
        TR.triple_begin = uni(seg.x);
        triple_len = uni(min(seg.y - seg.x, 64u));
    }
    TR.key_is_1 = hdr >> 16;
    TR.flags = Q.flags;
    F.pair_ent = B.tab_ent[TR.pair_begin + min((uint32_t)lane, max(pair_len, 1u) - 1u)];
    F.triple_row = B.tab_rows[TR.triple_rows + min((uint32_t)lane, G >= 3 ? RL(W.cnt, (int)TR.key_is_1) : 0u)];
    F.triple_ent = B.tab_ent[TR.triple_begin + min((uint32_t)lane, max(triple_len, 1u) - 1u)];
    PairState P;
    P.s = s; P.ti = ti;
    P.truncated = false;
    TM_STAT(49, 1);
    if (feasible) table_scan(C, W, P, TR, F);
    wave_sync();
}

// ------------------------------------------------------------------ block search (k_search)
//
// A wave searches a BLOCK of up to 64 templates of its workgroup's structure at a time.  The lead atoms of the first
// three template residues come from the structure's pair tables, and those stages are data-parallel over the whole
// block: the work items of all its templates are laid end to end over the lanes, so a pass of 64 lanes is full
// whatever the sizes of the individual templates' table segments (14 entries on average).
//
//   A   lane = template: its table references (16 B) and its BlockHdr (32 B) are loaded; prefix sum of the pair-segment
//       lengths.  What later stages need of a template stays in that lane's registers (SlotRegs) and is fetched by
//       the lanes that work on its items with ds_bpermute.
//   B   lane = entry of a pair segment: the stored squared distance against the template's window (1, 0); survivors
//       {positions, slot} are queued in LDS.
//   C   (per 64 queued pairs) lane = pair: its row of the (key, lead 2) table; then lane = item of a row: window
//       (2, key) on the stored distance, window (2, other) computed from the atom records; survivors {atoms, slot} are
//       queued in LDS.
//   D1  (per 64 queued triples) lane = triple: the candidates for the template's fourth search position -- the kind
//       list of a fourth lead, or the atoms of an assigned residue -- then lane = candidate: the three windows against
//       the placed atoms.  A triple with no candidate at all ends here (four in five do).
//
// What survives goes to global memory in pages of 64 triples, listed per structure, and k_deep searches below them:
// any wave of any workgroup that has the structure in LDS takes the next run of triples.  One long search -- a
// template with thousands of surviving triples in a large complex under loose thresholds -- is thereby shared out
// over as many waves as work on its structure, instead of being one wave's tail.
//
// Round 2 searched one template at a time in one wave: every search paid the set-up of the per-level constants
// (1 KB from L2 per search: 4.7 GB per launch), an exposed round trip to its tables, ran its pair and triple stages
// with a quarter of the lanes and its look at level 3 one triple at a time.

struct SlotRegs {  // lane = template slot of the block
    uint32_t ti;                // staged template index
    uint32_t base;              // first entry of its pair segment - items of the block before it (mod 2^32)
    uint32_t trows, tbegin;     // triple table: first row slot, first entry
    uint32_t lo0, lo1, lo2;     // kind lists of leads 0, 1, 2: first pool position
    uint32_t flags;             // BlockHdr::flags
    float w10lo, w10hi, wklo, wkhi, wolo, wohi;
};


// The page of surviving triples a wave is filling (uniform values).
struct PageWriter {
    uint32_t page;   // 0xffffffff: none yet
    uint32_t fill;
    SharedTriples *shared;  // the workgroup's queue (NULL: every surviving triple is paged out)
};

constexpr uint32_t kNoPage = 0xffffffffu, kVirtualPage = 0xfffffffeu;

// Lists the wave's current page in its structure's share of the directory.  A page that could not be had (the pool ran
// out: kVirtualPage) is still COUNTED, like a page that does not fit the share: dir_n[s] ends on what the structure
// needed, and the one retry the overflow costs is sized right.
__device__ __forceinline__ void page_publish(PageWriter &PW, uint32_t s, uint32_t lane) {
    if (PW.page == kNoPage || PW.fill == 0u) return;
    if (lane == 0) {
        ColdArgsPtr cold = cold_args();
        const uint32_t at = atomicAdd(&cold->O.dir_n[s], 1u);
        atomicAdd(&cold->O.dir_triples[s], PW.fill);
        const uint32_t lo = cold->O.dir_off[s], hi = cold->O.dir_off[s + 1];
        // (an entry inside the share is written whatever happened: k_deep of an overflowed query must not read garbage)
        if (at < hi - lo) cold->O.dir[lo + at] = PW.page != kVirtualPage ? make_uint2(PW.page, PW.fill) : make_uint2(0u, 0u);
        if (at >= hi - lo || PW.page == kVirtualPage) atomicOr(cold->O.overflow, 32u);
    }
    PW.page = kNoPage;
    PW.fill = 0u;
}

// Appends the triples of the lanes in `m` (each lane holds one: a01, c, ti) to the wave's page(s).
__device__ __forceinline__ void page_append(PageWriter &PW, uint32_t s, uint32_t lane, unsigned long long m, uint32_t a01,
                                            uint32_t c, uint32_t ti) {
    const bool mine = ((m >> lane) & 1ull) != 0ull;
    uint32_t rank = rank_in(m), left = (uint32_t)__popcll(m);
    while (left != 0u) {
        if (PW.page == kNoPage) {
            ColdArgsPtr cold = cold_args();
            unsigned long long page = 0;
            if (lane == 0) page = atomicAdd(cold->O.n_pages, 1ull);
            page = uni64(page);
            PW.page = page < cold->O.page_capacity ? (uint32_t)page : kVirtualPage;  // (the query then reports TM_E_OVERFLOW)
            PW.fill = 0u;
        }
        const uint32_t room = kPageTriples - PW.fill, take = min(room, left);
        if (mine && rank < take && PW.page != kVirtualPage) {
            DeepEntry *e = cold_args()->O.pages + (size_t)PW.page * kPageTriples + PW.fill + rank;
            e->a01 = a01; e->c = c; e->ti = ti;
        }
        rank -= take;  // lanes already written wrap to huge ranks: never < take again
        PW.fill += take;
        left -= take;
        if (PW.fill == kPageTriples) page_publish(PW, s, lane);
    }
}

// Stages D0 and D1 over the first `nt` queued triples: those that pass both go to the wave's page.
//
//   D0  Can the OTHER atoms of the three placed residues exist?  The search levels that are not leads and belong to
//       template residues 0, 1, 2 depend on nothing but the triple: lane = (triple, such level) -- laid end to end like the
//       table stages -- walks the atoms of the target residue its lead sits in and asks for one that carries an admissible
//       name and residue type and fits the windows against the three leads.  A triple with a level that has no candidate
//       cannot be completed whatever the remaining leads will be: it ends here (round 3 looked at search level 3 only,
//       and the depth-first part found the same out one level evaluation -- two hundred dependent instructions -- at a time).
//   D1  the lead of a FOURTH residue (search level 3 of templates with more than three residues): is there any candidate
//       in its kind list that fits the three leads?  Only for the triples D0 left.
template <class Ctx>
__device__ __forceinline__ void block_look(Ctx &C, const SearchArgs &A, SlotRegs &R, PageWriter &PW, uint32_t s,
                                           const uint32_t *kl, uint32_t nt, uint32_t from = 0u) {
    auto &S = *C.S;
    const uint32_t lane = (uint32_t)C.lane;
    const bool vt = lane < nt;
    uint2 t = make_uint2(0u, 0u);
    if (vt) t = S.triples[from + lane];
    const uint32_t slot = t.y >> 16, c = t.y & 0xffffu, x0 = t.x & 0xffffu, x1 = t.x >> 16;
    const uint32_t ti = lane_get(R.ti, slot);

    // ---- D0: lane = (triple, level of residues 0-2 that is not a lead), in two rounds: the first two such levels of
    // every triple, then the others of the triples that are left (most are not: 42 M items per launch become 25 M)
    TM_MARK("D0_begin");
    TM_T0(t_d0);
    TM_TN(11, 1);
    bool fail = false;
    {
        // (= LookHdr::n_rescheck, from the block's header; the exchange is made by EVERY lane: a lane that does not take part
        // hands out nothing, and the slots of a partly filled flush sit in lanes beyond its triples)
        const uint32_t slot_flags = lane_get(R.flags, slot);
        const uint32_t n_all = vt ? (slot_flags >> 8) & 31u : 0u;
        for (uint32_t round = 0; round < 2u; ++round) {
            const uint32_t j_first = round == 0u ? 0u : min(n_all, kResFirst);
            const uint32_t n_rl = fail ? 0u : (round == 0u ? min(n_all, kResFirst) : n_all - j_first);
            const uint32_t incl = wave_scan_add(n_rl), excl = incl - n_rl;
            const uint32_t total = RL(incl, 63);
            TM_STAT(62, total);
            for (uint32_t w0 = 0; w0 < total; w0 += 64u) {
                TM_MARK("D0_pass_begin");
                S.owner[lane] = 0u;
                {
                    const uint32_t start = max(excl, w0);
                    if (n_rl != 0u && start < w0 + 64u && incl > w0) S.owner[start - w0] = lane;
                }
                wave_sync();
                const uint32_t p = wave_scan_max(S.owner[lane]);
                wave_sync();
                const bool item = w0 + lane < total;
                const uint32_t w = min(w0 + lane, total - 1u);
                const uint32_t j = w - lane_get(excl, p) + lane_get(j_first, p);  // which of the triple's records
                const uint32_t pti = lane_get(ti, p);
                // (the triple's three records are asked for BEFORE the level's record: the compiler had put the queue entry into
                // the registers the first 16 bytes of the record were on their way to, and so waited for L2 before it even
                // asked LDS -- the two chains now run side by side)
                const uint2 pt = S.triples[from + p];  // (the owner's queue entry: its three atoms)
                const uint32_t px0 = pt.x & 0xffffu, px1 = pt.x >> 16, pc = pt.y & 0xffffu;
                const float4 l0 = C.rec[px0], l1 = C.rec[px1], l2 = C.rec[pc];
                const uint4 *rec4 = reinterpret_cast<const uint4 *>(A.Q.rescheck + (size_t)pti * kResChecks + j);
                const uint4 ra = rec4[0], rb = rec4[1];
                const uint2 rw2 = *reinterpret_cast<const uint2 *>(rec4 + 2);
                const uint32_t ref = ra.w & 0xffu;
                const uint32_t res = meta_res(__float_as_uint(ref == 0u ? l0.w : ref == 1u ? l1.w : l2.w));
                uint32_t a = C.res_start[res];
                const uint32_t a_end = C.res_start[res + 1u];
                const unsigned long long nm = ((unsigned long long)ra.y << 32) | ra.x;
                const bool any = (ra.w & 256u) != 0u;
                (void)any;  // (the scan reads the flag out of ra.z: see k_stage_templates)
                const float2 w0b = make_float2(__uint_as_float(rb.x), __uint_as_float(rb.y)), w1b = make_float2(__uint_as_float(rb.z), __uint_as_float(rb.w)),
                             w2b = make_float2(__uint_as_float(rw2.x), __uint_as_float(rw2.y));
                bool found = false;
                TM_MARK("D0_resloop_begin");
#if TM_D0_SCAN
                // Round 6.  The loop this replaces walked the residue's atoms with `found` and the end of the residue as per-lane
                // exits: a wave ran as many trips as its longest residue has atoms (a dozen), and in nearly every trip SOME lane
                // had an admissible name, so the three windows -- forty instructions under a handful of lanes -- were issued a
                // dozen times per pass.  Now the names are scanned first, in a loop whose trip count is the wave's (no per-lane
                // control flow, two atoms per trip: their packed words are asked for together), and leave a bit per admissible
                // atom; then the lanes test their candidates in lock step, lowest first -- once for a level that admits one name
                // (every lane tests its one atom in the same trip), a few times for a level that matches by element.  The same
                // atoms pass the same tests: a superset filter as before, parity cannot move.  Residues of more than 32 atoms
                // (hydrogens, ligands) go 32 atoms at a time.
                const uint32_t n_res_atoms = item && a_end > a ? a_end - a : 0u;
                uint32_t fnd = 0u;
                for (uint32_t base = 0; ballot64(base < n_res_atoms && fnd == 0u) != 0ull; base += 32u) {
                    const bool here = base < n_res_atoms && fnd == 0u;
                    const uint32_t n_here = here ? min(32u, n_res_atoms - base) : 0u;
                    const uint32_t a_here = a + (here ? base : 0u);  // (a lane that is done keeps reading inside its residue)
                    const uint32_t a_last = a_here + (n_here != 0u ? n_here - 1u : 0u);
                    uint32_t cand = 0u;
                    for (uint32_t i = 0; i < 32u; i += 2u) {
                        if (ballot64(i < n_here) == 0ull) break;
                        // (a lane past its residue repeats its last atom when the records are in global memory -- inside the
                        // arrays --, with the records in LDS it reads on; its bits are masked off after the loop)
                        const uint32_t i0 = Ctx::kRecInLds ? a_here + i : min(a_here + i, a_last);
                        const uint32_t i1 = Ctx::kRecInLds ? a_here + i + 1u : min(a_here + i + 1u, a_last);
                        const uint32_t m0 = __float_as_uint(C.rec[i0].w), m1 = __float_as_uint(C.rec[i1].w);
                        // (ra.z holds every type's bit when the level admits any residue type: k_stage_templates)
                        const uint32_t b0 = (uint32_t)(nm >> meta_name(m0)) & (ra.z >> meta_rtype(m0)) & 1u;
                        const uint32_t b1 = (uint32_t)(nm >> meta_name(m1)) & (ra.z >> meta_rtype(m1)) & 1u;
                        cand |= (b0 << i) | (b1 << (i + 1u));
                    }
                    cand &= n_here >= 32u ? 0xffffffffu : (1u << n_here) - 1u;
                    // the placed leads are not candidates (only the one this level shares its residue with can lie here)
                    {
                        const uint32_t d0 = px0 - a_here, d1 = px1 - a_here, d2 = pc - a_here;
                        cand &= ~((d0 < 32u ? 1u << d0 : 0u) | (d1 < 32u ? 1u << d1 : 0u) | (d2 < 32u ? 1u << d2 : 0u));
                    }
                    for (;;) {
                        if (ballot64(cand != 0u) == 0ull) break;
                        const uint32_t bit = cand != 0u ? (uint32_t)__builtin_ctz(cand) : 0u;
                        const float4 r = C.rec[a_here + bit];
                        const bool ok = (cand != 0u) & in_window(r, l0, w0b) & in_window(r, l1, w1b) & in_window(r, l2, w2b);
                        fnd |= ok ? 1u : 0u;
                        cand = ok ? 0u : cand & (cand - 1u);
                    }
                }
                found = fnd != 0u;
#elif TM_D0_UNROLL == 2
                // two atoms of the residue per trip: their packed words are asked for together (one LDS round trip for two
                // names; the loop is a chain of dependent round trips -- five on average, one per atom)
                auto admits = [&](uint32_t m, uint32_t at) {
                    return (((nm >> meta_name(m)) & 1ull) != 0ull) & (any | (((ra.z >> meta_rtype(m)) & 1u) != 0u)) & (at != px0) & (at != px1) & (at != pc);
                };
                for (; a < a_end && !found; a += 2u) {
                    const uint32_t b = a + 1u < a_end ? a + 1u : a;
                    const uint32_t m0 = __float_as_uint(C.rec[a].w), m1 = __float_as_uint(C.rec[b].w);
                    const bool c0 = admits(m0, a), c1 = admits(m1, b) & (b != a);
                    if (c0) {
                        const float4 r = C.rec[a];
                        found = in_window(r, l0, w0b) & in_window(r, l1, w1b) & in_window(r, l2, w2b);
                    }
                    if (c1 & !found) {
                        const float4 r = C.rec[b];
                        found = in_window(r, l0, w0b) & in_window(r, l1, w1b) & in_window(r, l2, w2b);
                    }
                }
#else
                for (; a < a_end && !found; ++a) {  // (a dozen atoms at most; the name leaves one of them, usually)
                    const uint32_t m = __float_as_uint(C.rec[a].w);
                    if ((((nm >> meta_name(m)) & 1ull) != 0ull) & (any | (((ra.z >> meta_rtype(m)) & 1u) != 0u)) & (a != px0) & (a != px1) & (a != pc)) {
                        const float4 r = C.rec[a];
                        found = in_window(r, l0, w0b) & in_window(r, l1, w1b) & in_window(r, l2, w2b);
                    }
                }
#endif
                TM_MARK("D0_resloop_end");
                const unsigned long long missing = ballot64(item && !found);
                TM_STAT(63, 1);
                // does any of MY triple's levels of this pass lack a candidate?  (items [max(excl, w0), min(incl, w0 + 64)) - w0)
                if (n_rl != 0u && incl > w0 && excl < w0 + 64u) {
                    const uint32_t lo = max(excl, w0) - w0, hi = min(incl, w0 + 64u) - w0;  // 0 <= lo < hi <= 64
                    const unsigned long long span = (hi == 64u ? ~0ull : ((1ull << hi) - 1ull)) & ~((1ull << lo) - 1ull);
                    fail = fail | ((missing & span) != 0ull);
                }
            }
        }
    }
    TM_MARK("D0_end");
    TM_T1(3, t_d0);
    TM_T0(t_d1);
    const bool alive = vt && !fail;
    {
        const uint32_t n_failed = (uint32_t)__popcll(ballot64(vt && fail));
        (void)n_failed;
        TM_STAT(69, n_failed);
    }

    // ---- D1: the lead of a fourth residue.  Its kind list is long (a hundred atoms is common), so these triples go one at
    // a time with their atoms and windows as wave-uniform values and the LANES over the candidates -- a candidate then
    // costs its record and nine multiply-adds, nothing is exchanged between lanes.
    const LookHdr H = A.Q.look[ti];
    const float4 q0 = C.rec[x0], q1 = C.rec[x1], q2 = C.rec[c];
    const bool lead3 = H.li3 != 0xffffu;
    uint32_t first = 0, n_c = 0;
    if (lead3 && alive) {
        first = kl[H.li3];
        n_c = kl[H.li3 + 1u] - first;
    }
    // (no level 3 at all, or level 3 is another atom of a placed residue, which D0 has looked at: nothing left to ask here)
    bool pass = alive && !lead3;
    unsigned long long found_lead = 0ull;
#ifdef TM_STATS
    {
        const uint32_t n_lead3 = (uint32_t)__popcll(ballot64(lead3 && n_c != 0u));
        TM_STAT(60, n_lead3);
    }
#endif
    // Consecutive triples mostly belong to one template and so share the kind list of their fourth lead: a list's first
    // 64 records stay in registers from one triple to the next, and the first chunk -- the only one for nine lists in ten --
    // is tested without a loop around it.
    uint32_t held_lo = 0xffffffffu, held_cnt = 0u;
    float4 held = make_float4(0.f, 0.f, 0.f, 0.f);
    struct Uniform {
        float ax, ay, az, bx, by, bz, cx, cy, cz;
        float2 wa, wb, wc;
        uint32_t lo, cnt;
    };
    // (a v_readlane per wave-uniform value is what a triple costs here: the windows and the list belong to the template,
    // which most consecutive triples share, so those eight values are fetched again only when the slot changes)
    uint32_t seen_slot = 0xffffffffu;
    Uniform tpl = {};
    auto uniform_of = [&](int q) {
        const uint32_t sl = RL(slot, q);
        if (sl != seen_slot) {
            tpl.wa = make_float2(RLf(H.w30.x, q), RLf(H.w30.y, q));
            tpl.wb = make_float2(RLf(H.w31.x, q), RLf(H.w31.y, q));
            tpl.wc = make_float2(RLf(H.w32.x, q), RLf(H.w32.y, q));
            tpl.lo = RL(first, q); tpl.cnt = RL(n_c, q);
            seen_slot = sl;
        }
        Uniform u = tpl;
        u.ax = RLf(q0.x, q); u.ay = RLf(q0.y, q); u.az = RLf(q0.z, q);
        u.bx = RLf(q1.x, q); u.by = RLf(q1.y, q); u.bz = RLf(q1.z, q);
        u.cx = RLf(q2.x, q); u.cy = RLf(q2.y, q); u.cz = RLf(q2.z, q);
        return u;
    };
    auto fits = [&](const Uniform &u, const float4 &r, uint32_t idx) {
        float dx = r.x - u.ax, dy = r.y - u.ay, dz = r.z - u.az;
        const float da = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        dx = r.x - u.bx; dy = r.y - u.by; dz = r.z - u.bz;
        const float db = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        dx = r.x - u.cx; dy = r.y - u.cy; dz = r.z - u.cz;
        const float dc = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
        return (idx < u.cnt) & (da >= u.wa.x) & (da <= u.wa.y) & (db >= u.wb.x) & (db <= u.wb.y) & (dc >= u.wc.x) & (dc <= u.wc.y);
    };
    for (unsigned long long todo = ballot64(lead3 && n_c != 0u); todo;) {
        const int q = __ffsll((long long)todo) - 1;
        todo &= todo - 1ull;
        const Uniform u = uniform_of(q);
        if (u.lo != held_lo || u.cnt != held_cnt) {
            held = C.rec[C.pool_at(u.lo + min(lane, u.cnt - 1u))];
            held_lo = u.lo;
            held_cnt = u.cnt;
        }
        bool f = ballot64(fits(u, held, lane)) != 0ull;
        TM_STAT(61, 1);
        // lists of more than 64 atoms: the chunks beyond the first
        for (uint32_t base = 64u; base < u.cnt && !f; base += 64u)
            f = ballot64(fits(u, C.rec[C.pool_at(u.lo + min(base + lane, u.cnt - 1u))], base + lane)) != 0ull;
        if (f) found_lead |= 1ull << q;
    }
    TM_MARK("D1_end");
    pass = pass | (((found_lead >> lane) & 1ull) != 0ull);
    unsigned long long keep = ballot64(pass);
    {
        const uint32_t n_keep = (uint32_t)__popcll(keep);
        (void)n_keep;
        TM_STAT(64, 1);
        TM_STAT(65, n_keep);
        TM_STAT(66, nt);
    }
#if defined(TM_STATS) || defined(TM_ABLATE)
    if (A.Q.flags & 0x800u) keep = 0ull;  // phase timing: the look at level 3, nothing below it
#endif
    if (keep) {
        // into the workgroup's queue (searched by whichever wave comes to the end of a block next: drain_shared); what does
        // not fit goes to the structure's pages, where every wave working on the structure takes a share (k_deep)
        if (PW.shared != nullptr) {
            SharedTriples &Q2 = *PW.shared;
            const uint32_t r = rank_in(keep), n_keep = (uint32_t)__popcll(keep);
            uint32_t base = 0;
            if (lane == 0u) base = atomicAdd(&Q2.tail, n_keep);
            base = uni(base);
            const bool mine = ((keep >> lane) & 1ull) != 0ull;
            const bool here = mine && base + r < kSharedTriples;
            if (here) {
                SharedEntry &e = Q2.entry[base + r];
                uint32_t *a = reinterpret_cast<uint32_t *>(e.atoms);
                a[0] = t.x; a[1] = c & 0xffffu;
                e.depth = 3u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __hip_atomic_store(&e.ti, ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            keep &= ~ballot64(here);
        }
        if (keep) page_append(PW, s, lane, keep, t.x, c, ti);
    }
    TM_T1(4, t_d1);
}

// Moves queue entries [from, from + n) to the front (n < 64).
__device__ __forceinline__ void queue_shift(uint2 *queue, uint32_t from, uint32_t n, uint32_t lane) {
    uint2 v = make_uint2(0u, 0u);
    if (lane < n) v = queue[from + lane];
    wave_sync();
    if (lane < n) queue[lane] = v;
    wave_sync();
}

// The pair queue of the block search: {list positions i | j << 16, template slot} -- 8 bytes, or packed into one word when
// the structure is in LDS (Ctx::kRecInLds).
template <class Ctx, class Scratch>
__device__ __forceinline__ void pair_put(Scratch &S, uint32_t at, uint32_t pos, uint32_t slot) {
    if constexpr (Ctx::kRecInLds)
        reinterpret_cast<uint32_t *>(S.pairs)[at] = (pos & kPackedPosMask) | ((pos >> 16) << kPackedPos) | (slot << (2u * kPackedPos));
    else S.pairs[at] = make_uint2(pos, slot);
}
template <class Ctx, class Scratch>
__device__ __forceinline__ uint2 pair_get(const Scratch &S, uint32_t at) {
    if constexpr (Ctx::kRecInLds) {
        const uint32_t w = reinterpret_cast<const uint32_t *>(S.pairs)[at];
        return make_uint2((w & kPackedPosMask) | (((w >> kPackedPos) & kPackedPosMask) << 16), w >> (2u * kPackedPos));
    } else return S.pairs[at];
}
template <class Ctx, class Scratch>
__device__ __forceinline__ void pair_shift(Scratch &S, uint32_t from, uint32_t n, uint32_t lane) {
    if constexpr (Ctx::kRecInLds) {
        uint32_t *q = reinterpret_cast<uint32_t *>(S.pairs);
        uint32_t v = 0u;
        if (lane < n) v = q[from + lane];
        wave_sync();
        if (lane < n) q[lane] = v;
        wave_sync();
    } else queue_shift(S.pairs, from, n, lane);
}

// Stage C over the first `np` queued pairs; survivors go to the triple queue (nq3 entries in it), which is worked
// off (stage D1) whenever it holds 64.
template <class Ctx>
__device__ __forceinline__ uint32_t block_pairs(Ctx &C, const SearchArgs &A, SlotRegs &R, PageWriter &PW, uint32_t s,
                                                const uint32_t *kl, uint32_t np, uint32_t &nq3, uint32_t from = 0u,
                                                uint32_t w_start = 0u) {
    auto &S = *C.S;
    const uint32_t lane = (uint32_t)C.lane;
    TM_MARK("C_setup_begin");
    const bool vp = lane < np;
    uint2 pq = make_uint2(0u, 0u);
    if (vp) pq = pair_get<Ctx>(S, from + lane);
    const uint32_t slot = pq.y, pos = pq.x;
    const uint32_t flags = lane_get(R.flags, slot);
    const bool key1 = (flags & 1u) != 0u;
    const uint32_t key = key1 ? pos >> 16 : pos & 0xffffu;
    const uint32_t trows = lane_get(R.trows, slot);
    // the row of the key atom (lanes without a pair read slot 0's first row: a valid address)
    const uint32_t r0 = C.tab_rows[trows + key], r1 = C.tab_rows[trows + key + 1u];
    // (A lane without a pair reads through slot 0: position 0 of that template's lead lists.  Should the template in slot 0 have
    // no second lead, or an empty list, that position is the first entry BEHIND a list -- for the last list of the pool
    // whatever the allocation held before --, and the atom record it names is read: harmless while the records are in LDS
    // (a read beyond LDS returns zero), a memory fault in waiting when they are in global memory (round 5: a 25 k-atom
    // complex against one- and two-atom templates, depending on what ran before).  With the kind lists in global memory
    // such a lane reads the structure's first list entry instead.)
    uint32_t i0 = lane_get(R.lo0, slot) + (pos & 0xffffu), i1 = lane_get(R.lo1, slot) + (pos >> 16);
    if constexpr (!Ctx::kPoolInLds) {
        const uint32_t safe = kl[0];
        i0 = vp ? i0 : safe;
        i1 = vp ? i1 : safe;
    }
    const uint32_t a0 = C.pool_at(i0), a1 = C.pool_at(i1);
    const float4 ro = C.rec[key1 ? a0 : a1];  // the lead that is NOT the key: its distance to lead 2 is computed
    const uint32_t a01 = a0 | (a1 << 16);
    const uint32_t lo2 = lane_get(R.lo2, slot);
    const float wklo = lane_getf(R.wklo, slot), wkhi = lane_getf(R.wkhi, slot);
    const float wolo = lane_getf(R.wolo, slot), wohi = lane_getf(R.wohi, slot);
    const uint32_t len = vp ? r1 - r0 : 0u;
    const uint32_t incl = wave_scan_add(len), excl = incl - len;
    const uint32_t total = RL(incl, 63);
    const uint32_t row_base = lane_get(R.tbegin, slot) + r0 - excl;  // entry of item w of this pair's row: row_base + w
    TM_MARK("C_setup_end");
    for (uint32_t w0 = w_start; w0 < total; w0 += 64u * kCBatch) {
        TM_MARK("C_group_begin");
#if TM_C_RESTART
        // The triple queue is worked off by the CALLER (pairs_group), not from inside this loop: stages D0/D1 inlined here
        // had this stage's two dozen live values on top of their own (the kernel sits at the 128 registers of four waves
        // per SIMD).  When the queue could not take another group of passes -- rare: a dozen triples come out of 64 pairs
        // -- the stage returns where it stands and is called again for the same pairs once the queue has been worked off.
        if (nq3 + 64u * kCBatch > (uint32_t)kBlockTripleQueue) return w0;
#endif
        // owner of item w: each pair writes its lane where its row starts (or at item w0), a running maximum spreads it;
        // the entries of the group's passes are asked for before the first of them is looked at
        uint32_t owner_of[kCBatch];
        uint2 ent_of[kCBatch];
#pragma unroll
        for (uint32_t k = 0; k < kCBatch; ++k) {
            const uint32_t wk = w0 + 64u * k;
            owner_of[k] = 0u; ent_of[k] = make_uint2(0u, 0u);
            if (k != 0u && wk >= total) continue;
            S.owner[lane] = 0u;
            {
                const uint32_t start = max(excl, wk);
                if (len != 0u && start < wk + 64u && incl > wk) S.owner[start - wk] = lane;
            }
            wave_sync();
            owner_of[k] = wave_scan_max(S.owner[lane]);
            wave_sync();
            const uint32_t w = min(wk + lane, total - 1u);  // lanes past the last item repeat it and are masked out of the vote
            ent_of[k] = C.tab_ent[lane_get(row_base, owner_of[k]) + w];
        }
#pragma unroll
        for (uint32_t k = 0; k < kCBatch; ++k) {
            const uint32_t wk = w0 + 64u * k;
            if (k != 0u && wk >= total) continue;
            const uint32_t p = owner_of[k];
            const uint2 ent = ent_of[k];
            const float dk2 = __uint_as_float(ent.y);
            bool ok = (wk + lane < total) & (dk2 >= lane_getf(wklo, p)) & (dk2 <= lane_getf(wkhi, p));
            const uint32_t c2 = C.pool_at(lane_get(lo2, p) + (ent.x >> 16));
            const float4 rc = C.rec[c2];
            const float ox = lane_getf(ro.x, p), oy = lane_getf(ro.y, p), oz = lane_getf(ro.z, p);
            const uint32_t ores = meta_res(lane_get(__float_as_uint(ro.w), p));
            const float dx = rc.x - ox, dy = rc.y - oy, dz = rc.z - oz;
            const float d2 = __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
            ok = ok & (d2 >= lane_getf(wolo, p)) & (d2 <= lane_getf(wohi, p)) & (meta_res(__float_as_uint(rc.w)) != ores);
            const uint32_t pa01 = lane_get(a01, p), pslot = lane_get(slot, p);
            if (!C.ignore_chain) {
                const uint32_t pf = lane_get(flags, p);  // (every lane takes part in the exchange: not under `if (ok)`)
                if (ok) {
                    const uint8_t qc = C.qchain[c2];
                    ok = ((qc == C.qchain[pa01 & 0xffffu]) == (((pf >> 2) & 1u) != 0u)) &
                         ((qc == C.qchain[pa01 >> 16]) == (((pf >> 3) & 1u) != 0u));
                }
            }
            unsigned long long m = ballot64(ok);
            {
                const uint32_t n_m = (uint32_t)__popcll(m);
                (void)n_m;
                TM_STAT(67, 1);
                TM_STAT(68, n_m);
            }
#if defined(TM_STATS) || defined(TM_ABLATE)
            if (A.Q.flags & 0x400u) m = 0ull;
#endif
            if (m == 0ull) continue;
            if (ok) S.triples[nq3 + rank_in(m)] = make_uint2(pa01, c2 | (pslot << 16));
            nq3 += (uint32_t)__popcll(m);
        }
        TM_MARK("C_group_end");
#if !TM_C_RESTART
        if (nq3 >= 64u) {  // (at most 63 + 64 x kCBatch queued: worked off 64 at a time, the rest moved to the front)
            wave_sync();
            uint32_t qh = 0;
            for (; nq3 - qh >= 64u; qh += 64u) block_look(C, A, R, PW, s, kl, 64u, qh);
            queue_shift(S.triples, qh, nq3 - qh, lane);
            nq3 -= qh;
        }
#endif
    }
    return 0xffffffffu;
}

// Stage C over `n` queued pairs from queue position `qh`, and stages D0/D1 over the triple queue whenever it holds 64.
template <class Ctx>
__device__ __forceinline__ void pairs_group(Ctx &C, const SearchArgs &A, SlotRegs &R, PageWriter &PW, uint32_t s,
                                            const uint32_t *kl, uint32_t n, uint32_t &nq3, uint32_t qh) {
    auto &S = *C.S;
    (void)S;
    for (uint32_t w = 0;;) {
        TM_T0(t_c);
        w = block_pairs(C, A, R, PW, s, kl, n, nq3, qh, w);
        TM_T1(2, t_c);
        TM_TN(10, 1);
#if TM_C_RESTART
        if (nq3 >= 64u) {  // (at most 63 + 64 x kCBatch queued: worked off 64 at a time, the rest moved to the front)
            wave_sync();
            uint32_t th = 0;
            for (; nq3 - th >= 64u; th += 64u) block_look(C, A, R, PW, s, kl, 64u, th);
            queue_shift(S.triples, th, nq3 - th, (uint32_t)C.lane);
            nq3 -= th;
        }
#endif
        if (w == 0xffffffffu) break;
    }
}

// The depth-first search below triples of the workgroup's queue (wave-cooperative: dfs), one triple at a time until the
// queue is empty.  Called when the wave's stage queues are empty: the template's pair bounds and lead-list chunks take
// their place in LDS.
template <class Ctx>
__device__ __forceinline__ void drain_shared(Ctx &C, const SearchArgs &A, PageWriter &PW, uint32_t s, const uint32_t *kl) {
    if (PW.shared == nullptr) return;
    SharedTriples &Q2 = *PW.shared;
    const QueryView &Q = A.Q;
    const uint32_t lane = (uint32_t)C.lane;
    WaveRegs W = {};
    uint32_t cur_ti = 0xffffffffu;
    bool feasible = false;
    PairState P;
    P.s = s; P.ti = 0; P.truncated = false;
    for (;;) {
        // take the next entry, if there is one (a compare-and-swap: the head never runs past what has been reserved)
        uint32_t idx = 0xffffffffu;
        if (lane == 0u) {
            uint32_t head = __hip_atomic_load(&Q2.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (;;) {
                const uint32_t tail = min(__hip_atomic_load(&Q2.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), kSharedTriples);
                if (head >= tail) break;
                if (__hip_atomic_compare_exchange_strong(&Q2.head, &head, head + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    idx = head;
                    break;
                }
            }
        }
        idx = uni(idx);
        if (idx == 0xffffffffu) break;
        TM_T0(t_q);
        TM_TN(12, 1);
        uint32_t ti;  // (reserved before it is written: wait for the writer, a wave of this workgroup a few instructions behind)
        do {
            ti = uni(__hip_atomic_load(&Q2.entry[idx].ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        } while (ti == 0xffffffffu);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
#ifdef TM_STATS
        const uint32_t depth_word = uni(Q2.entry[idx].depth), depth = depth_word & 0xffu;
#else
        const uint32_t depth = uni(Q2.entry[idx].depth);
#endif
        const uint32_t mine = Q2.entry[idx].atoms[min(lane, (uint32_t)kMaxK - 1u)];  // lane L < depth: the atom of level L
#ifdef TM_STATS
        if (depth > 3u) {
            uint32_t sum = 0;
            for (uint32_t j = 0; j < depth; ++j) sum += RL(mine, (int)j);
            TM_STAT(300, (sum & 0xffffffu) != (depth_word >> 8) ? 1 : 0);
            TM_STAT(301, 1);
        }
#endif
        if (ti != cur_ti) {
            TemplateRegs T;
            load_template(Q, ti, (int)lane, T);
            feasible = setup_template(C, W, T, kl, Q, ti);
            cur_ti = ti;
            P.ti = ti;
            P.truncated = false;
        }
        if (P.truncated || !feasible) continue;
        {  // the placed leads: lane L < depth fetches and keeps the entry of level L (lane = level)
            const uint32_t atom = lane < depth ? mine : 0u;
            const float4 r = C.rec[atom];
            if (lane < depth) {
                W.aidx = atom;
                W.ax = r.x; W.ay = r.y; W.az = r.z;
                W.ameta = __float_as_uint(r.w);
                if (!C.ignore_chain) W.achain = (uint32_t)C.qchain[atom];
            }
        }
        TM_STAT(84 + min(C.n_groups, 8), 1);
#ifdef TM_STATS
        const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
        dfs(C, W, P, (int)depth, &Q2);
        TM_T1(5, t_q);
#ifdef TM_STATS
        {
            const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t_entry;
            const uint32_t bl = min(23u, 64u - (uint32_t)__clzll((long long)dt));
            TM_STAT(192 + bl, 1);
            TM_STAT(216 + bl, dt);
            TM_STAT(240 + min(depth, 15u), 1);
        }
#endif
    }
    wave_sync();  // (the next block's stages write the queues again)
}

// One block: templates chunk + n_chunks * (m0 .. m0 + count - 1) of the staged selection against structure s.
template <class Ctx>
__device__ __forceinline__ void search_block(Ctx &C, const SearchArgs &A, PageWriter &PW, uint32_t s, uint32_t m0, uint32_t count,
                                             uint32_t chunk, uint32_t n_chunks, const uint32_t *kl, const uint4 *tab_ref) {
    const QueryView &Q = A.Q;
    auto &S = *C.S;
    const uint32_t lane = (uint32_t)C.lane;
    const unsigned long long t_block = Q.cost ? __builtin_amdgcn_s_memtime() : 0ull;

    // ---- A: lane = template
    TM_MARK("A_begin");
    TM_T0(t_a);
    TM_TN(9, 1);
    const bool valid = lane < count;
    SlotRegs R;
    R.ti = chunk + n_chunks * (m0 + (valid ? lane : 0u));
    const uint4 ref = tab_ref[R.ti];
    const BlockHdr H = Q.hdr[R.ti];
    // (templates marked kRefSingly are searched by the k_search<..., kSingly = true> launch)
    const uint32_t len = valid && (ref.y & kRefSingly) == 0u ? ref.y & kRefLenMask : 0u;
    R.trows = ref.z; R.tbegin = ref.w;
    R.flags = H.flags;
    R.w10lo = H.w10.x; R.w10hi = H.w10.y;
    R.wklo = H.wkey.x; R.wkhi = H.wkey.y;
    R.wolo = H.wother.x; R.wohi = H.wother.y;
    R.lo0 = kl[min((uint32_t)H.li[0], A.L.n_lead)];
    R.lo1 = kl[min((uint32_t)H.li[1], A.L.n_lead)];
    R.lo2 = kl[min((uint32_t)H.li[2], A.L.n_lead)];
    const uint32_t incl = wave_scan_add(len), excl = incl - len;
    const uint32_t total = RL(incl, 63);
    R.base = ref.x - excl;
    TM_STAT(49, count);

    TM_MARK("A_end");
    TM_T1(0, t_a);
    // ---- B: lane = entry of a pair segment
    uint32_t nq = 0, nq3 = 0;
    bool skip = false;
#if defined(TM_STATS) || defined(TM_ABLATE)
    skip = (Q.flags & 0x200u) != 0u;
#endif
    constexpr uint32_t kBB = Ctx::kRecInLds ? kBBatchLds : kBBatch;  // passes whose table entries are asked for together
    for (uint32_t w0 = 0; w0 < total && !skip; w0 += 64u * kBB) {
        TM_MARK("B_group_begin");
        TM_T0(t_b);
        uint32_t owner_of[kBB];
        uint2 ent_of[kBB];
#pragma unroll
        for (uint32_t k = 0; k < kBB; ++k) {
            const uint32_t wk = w0 + 64u * k;
            owner_of[k] = 0u; ent_of[k] = make_uint2(0u, 0u);
            if (k != 0u && wk >= total) continue;
            S.owner[lane] = 0u;
            {
                const uint32_t start = max(excl, wk);
                if (len != 0u && start < wk + 64u && incl > wk) S.owner[start - wk] = lane;
            }
            wave_sync();
            owner_of[k] = wave_scan_max(S.owner[lane]);
            wave_sync();
            const uint32_t w = min(wk + lane, total - 1u);
            ent_of[k] = C.tab_ent[lane_get(R.base, owner_of[k]) + w];
        }
#pragma unroll
        for (uint32_t k = 0; k < kBB; ++k) {
            const uint32_t wk = w0 + 64u * k;
            if (k != 0u && wk >= total) continue;
            const uint32_t p = owner_of[k];
            const uint2 ent = ent_of[k];
            const float d2 = __uint_as_float(ent.y);
            bool ok = (wk + lane < total) & (d2 >= lane_getf(R.w10lo, p)) & (d2 <= lane_getf(R.w10hi, p));
            if (!C.ignore_chain) {
                const uint32_t l0 = lane_get(R.lo0, p), l1 = lane_get(R.lo1, p), pf = lane_get(R.flags, p);
                if (ok) ok = (C.qchain[C.pool_at(l0 + (ent.x & 0xffffu))] == C.qchain[C.pool_at(l1 + (ent.x >> 16))]) == (((pf >> 1) & 1u) != 0u);
            }
            unsigned long long m = ballot64(ok);
            {
                const uint32_t n_m = (uint32_t)__popcll(m);
                (void)n_m;
                TM_STAT(52, 1);
                TM_STAT(53, n_m);
            }
#if defined(TM_STATS) || defined(TM_ABLATE)
            if (Q.flags & 0x100u) m = 0ull;
#endif
            if (m == 0ull) continue;
            if (ok) pair_put<Ctx>(S, nq + rank_in(m), ent.x, p);
            nq += (uint32_t)__popcll(m);
        }
        TM_MARK("B_group_end");
        TM_T1(1, t_b);
        if (nq >= 64u) {  // (at most 63 + 64 x kBB queued: worked off 64 at a time, the rest moved to the front)
            wave_sync();
            uint32_t qh = 0;
            for (; nq - qh >= 64u; qh += 64u) pairs_group(C, A, R, PW, s, kl, 64u, nq3, qh);
            pair_shift<Ctx>(S, qh, nq - qh, lane);
            nq -= qh;
        }
    }
    if (nq != 0u) {
        wave_sync();
        pairs_group(C, A, R, PW, s, kl, nq, nq3, 0u);
    }
    if (nq3 != 0u) {
        wave_sync();
        block_look(C, A, R, PW, s, kl, nq3);
    }
    drain_shared(C, A, PW, s, kl);
    if (Q.cost && valid) {  // calibration query: the table-driven stages are shared out evenly over the block's templates
        const unsigned long long spent = __builtin_amdgcn_s_memtime() - t_block;
        atomicAdd(&Q.cost[R.ti], spent / (64ull * count));
    }
    wave_sync();
}

// The template loop of one workgroup of k_search: its waves claim blocks of the workgroup's chunk of the staged
// selection and search the structure staged in LDS.  kPoolLds: the structure's kind lists are in LDS too (the rule),
// else in global memory.  Block sizes are guided: 64 templates while much is left, fewer towards the end of the chunk
// (down to 8), so that the sixteen waves run out of work at about the same time.
template <bool kLds, bool kIgnoreChain, bool kPoolLds, bool kSingly>
__device__ __forceinline__ void search_templates(const SearchArgs &A, WaveScratchT<kSingly ? kModeSingly : kModeBlocks> *scratch, const float4 *s_rec,
                                                 const uint16_t *s_res, const uint32_t *s_kl, const uint16_t *s_pool,
                                                 const uint32_t *kl_off, uint32_t *s_next_p, uint32_t s, uint32_t a0,
                                                 uint32_t r0, uint32_t chunk, uint32_t n_chunks, SharedTriples *shared = nullptr) {
    const BatchView &B = A.B;
    const QueryView &Q = A.Q;
    SearchCtx<kIgnoreChain, kPoolLds, kSingly ? kModeSingly : kModeBlocks, false, kSingly && !kLds, kLds> C;
    C.rc = nullptr;
    C.rec = kLds ? s_rec : B.q_rec + a0;
    C.res_start = kLds ? s_res : B.res_start + r0;
    C.pool_lds = (const __attribute__((address_space(3))) uint16_t *)s_pool;
    C.pool_glb = B.kl_pool;
    // (a wide structure: the upper bits of its kind-list entries and residue starts; k_search lets one through to the
    // template-at-a-time launch on global memory only)
    const bool wide = B.s_off[s + 1] - a0 > kNarrowAtoms;
    C.pool_hi = wide ? B.kl_pool_hi : nullptr;
    C.res_hi = wide && B.res_start_hi ? B.res_start_hi + r0 : nullptr;
    C.qchain = B.q_chain + a0;
    C.tab_rows = B.tab_rows;
    C.tab_ent = B.tab_ent;
    C.S = scratch + (threadIdx.x >> 6);
    C.lane = lane_id();
    C.k = 0; C.n_groups = 0; C.bnd = Q.bounds;
#ifdef TM_TIMES
    for (int i = 0; i < 13; ++i) C.tacc[i] = 0ull;
    const unsigned long long t_templates = __builtin_amdgcn_s_memtime();
#endif
    const int lane = C.lane;
    const uint32_t *kl = kLds ? s_kl : kl_off;
    const uint4 *tab_ref = B.tab_ref + (size_t)s * Q.n_sel;
    if constexpr (kSingly) {
        // the template-at-a-time launch: the waves take the listed templates of this chunk one by one and search those
        // whose table reference says so
        WaveRegs W = {};
        slab_init(C);
        for (;;) {
            uint32_t m = 0;
            if (lane == 0) m = atomicAdd(s_next_p, 1u);
            m = uni(m);
            const uint32_t at = chunk + n_chunks * m;
            if (at >= A.n_singly) break;
            const uint32_t ti = A.singly[at];
            if ((uni(tab_ref[ti].y) & kRefSingly) != 0u) search_singly(C, A, W, s, ti, kl);
        }
        slab_release(C);
    } else {
    const uint32_t n_mine = (Q.n_sel - chunk + n_chunks - 1u) / n_chunks;  // templates chunk, chunk + n_chunks, ...
    PageWriter PW;
    PW.page = kNoPage; PW.fill = 0u;
    // (a structure that is not in LDS is searched below its triples by k_deep only, which keeps the residues it touches in LDS)
    PW.shared = kLds && !(Q.flags & 0x8000u) ? shared : nullptr;  // (flag: test hook ENZYMM_AMD_LOCAL_DEEP=0)
    slab_init(C);
    for (;;) {
        uint32_t m0 = 0, count = 0;
        if (lane == 0) {
            const uint32_t seen = __atomic_load_n(s_next_p, __ATOMIC_RELAXED);
            const uint32_t left = seen < n_mine ? n_mine - seen : 0u;
            // (a structure that is not in LDS -- a large complex: hundreds of table entries per template, and a batch of few
            // structures is cut into chunks of a few dozen templates -- takes blocks from 2 templates: with the LDS path's
            // minimum one wave of sixteen took a workgroup's whole chunk: 7 % of the wave-time of the launch at work)
            count = min(64u, max(kLds ? kMinBlock : (uint32_t)TM_MIN_BLOCK_GLOBAL, (left / (2u * kSearchWaves)) & ~7u));
            m0 = atomicAdd(s_next_p, count);
        }
        m0 = uni(m0);
        count = uni(count);
        if (m0 >= n_mine) break;
        search_block(C, A, PW, s, m0, min(count, n_mine - m0), chunk, n_chunks, kl, tab_ref);
    }
    drain_shared(C, A, PW, s, kl);  // (what other waves queued since this wave's last block)
    if (PW.shared != nullptr) {
        // out of blocks: wait for what the waves still at work hand over (dfs: give_away) until none of them is left.
        // (`busy` only tells a waiting wave when to stop looking: whoever queues an entry also comes back for the queue)
        SharedTriples &Q2 = *PW.shared;
        if (lane == 0) atomicSub(&Q2.busy, 1u);
        TM_T0(t_wait);
        for (;;) {
            const uint32_t head = uni(__hip_atomic_load(&Q2.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
            const uint32_t tail = min(uni(__hip_atomic_load(&Q2.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)), kSharedTriples);
            if (head < tail) {
                if (lane == 0) atomicAdd(&Q2.busy, 1u);
                drain_shared(C, A, PW, s, kl);
                if (lane == 0) atomicSub(&Q2.busy, 1u);
                continue;
            }
            if (uni(__hip_atomic_load(&Q2.busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0u) break;
            __builtin_amdgcn_s_sleep(16);
        }
        TM_T1(6, t_wait);
    }
    page_publish(PW, s, (uint32_t)lane);
    slab_release(C);
#ifdef TM_TIMES
    C.tacc[8] += __builtin_amdgcn_s_memtime() - t_templates;
    if (lane == 0)
        for (int i = 0; i < 13; ++i) atomicAdd(&g_times[i], C.tacc[i]);
#endif
    }
}

// ------------------------------------------------------------------ depth-first part (k_deep)

// The template whose per-level constants sit in W, and the one being fetched behind it.
struct DeepState {
    uint32_t ti;          // 0xffffffff: none
    PairState P;
    bool feasible;
    TemplateRegs next;    // constants of the template of the next triple of the run that belongs to another template
    uint32_t next_ti;     // 0xffffffff: nothing fetched
    unsigned long long since;  // calibration query: shader clock when this template's turn began
};

// The triples of one structure, searched below level 2 by whichever waves have the structure in LDS: every wave of every
// workgroup of the structure claims runs of kDeepRun triples from the structure's pages through one counter.
template <bool kLds, bool kIgnoreChain, bool kPoolLds>
__device__ __forceinline__ void deep_templates(const SearchArgs &A, WaveScratchT<kModeDeep> *scratch, const float4 *s_rec,
                                               const uint16_t *s_res, const uint32_t *s_kl, const uint16_t *s_pool,
                                               const uint32_t *kl_off, uint32_t s, uint32_t a0, uint32_t r0, uint32_t n_listed,
                                               ResCache *cache, const uint32_t *s_run, uint32_t n_compact, SharedTriples *shared = nullptr) {
    const BatchView &B = A.B;
    const QueryView &Q = A.Q;
    SearchCtx<kIgnoreChain, kPoolLds, kModeDeep, !kLds> C;
    C.rc = cache ? cache + (threadIdx.x >> 6) : nullptr;
    if (!kLds) {
        if (lane_id() < kMaxG) C.rc->tag[lane_id()] = 0xffffffffu;
        wave_sync();
    }
    C.rec = kLds ? s_rec : B.q_rec + a0;
    C.res_start = kLds ? s_res : B.res_start + r0;
    C.pool_lds = (const __attribute__((address_space(3))) uint16_t *)s_pool;
    C.pool_glb = B.kl_pool;
    C.qchain = B.q_chain + a0;
    C.tab_rows = B.tab_rows;
    C.tab_ent = B.tab_ent;
    C.S = scratch + (threadIdx.x >> 6);
    C.lane = lane_id();
    C.k = 0; C.n_groups = 0; C.bnd = Q.bounds;
    const uint32_t lane = (uint32_t)C.lane;
    const uint32_t *kl = kLds ? s_kl : kl_off;
    const uint2 *dir = A.O.dir + A.O.dir_off[s];
    // Triples are claimed kDeepRun at a time: fewer claims on the structure's one counter, and the next template's
    // constants load under the current triple's search.
    // What a claim number means: with the workgroup's run list (k_search<..., kModeDeep> builds it in LDS when the
    // structure's pages are few) the n-th run that HAS triples -- most pages are the partly filled last pages of the
    // block search's waves, and a claim that lands on the empty part of one costs the round trips of a claim and gets
    // no work behind which to hide the next --; without it run n of the listed pages, kRunsPerPage to a page.
    constexpr uint32_t run_len = kDeepRun;
    constexpr uint32_t kRunsPerPage = kPageTriples / run_len;
    const bool compact = n_compact != 0xffffffffu;
    const uint32_t n_runs = compact ? n_compact : n_listed * kRunsPerPage;
    const DeepEntry *pages = A.O.pages;
    // the entries of run r (lane q < nt holds triple q) -- a load that is left in flight: callers use it a run later
    auto fetch = [&](uint32_t r, DeepEntry &t) -> uint32_t {
        t.a01 = 0u; t.c = 0u; t.ti = 0xffffffffu;
        if (r >= n_runs) return 0u;
        size_t first;
        uint32_t nt;
        if (compact) {
            const uint32_t e = uni(s_run[r]);
            first = (size_t)(e & ~(run_len - 1u));
            nt = (e & (run_len - 1u)) + 1u;
        } else {
            const uint2 pg = dir[r / kRunsPerPage];  // {page, triples in it}
            const uint32_t e0 = (r % kRunsPerPage) * run_len;
            nt = uni(pg.y) > e0 ? min(run_len, uni(pg.y) - e0) : 0u;
            first = (size_t)uni(pg.x) * kPageTriples + e0;
        }
        if (lane < nt) t = pages[first + lane];
        return nt;
    };
    WaveRegs W = {};
    DeepState D;
    D.ti = 0xffffffffu; D.next_ti = 0xffffffffu; D.feasible = false; D.since = 0ull;
    D.P.s = s; D.P.ti = 0; D.P.truncated = false;
    slab_init(C);
    // two runs are claimed ahead: while run n is searched, the entries of run n + 1 are on their way (and, towards the
    // end of run n, the constants of its first template), and the claim of run n + 2 is
    // What the waves of this workgroup hand over (a structure in global memory; dfs: give_away): partial assignments in the
    // workgroup's queue, searched by whichever wave has nothing else to do -- between two runs, and when the runs are gone.
    auto drain = [&]() {
        if constexpr (!kLds) {
            if (shared == nullptr) return;
            SharedTriples &Q2 = *shared;
            for (;;) {
                uint32_t idx = 0xffffffffu;
                if (lane == 0u) {
                    uint32_t head = __hip_atomic_load(&Q2.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    for (;;) {
                        const uint32_t tail = min(__hip_atomic_load(&Q2.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP), kSharedTriples);
                        if (head >= tail) break;
                        if (__hip_atomic_compare_exchange_strong(&Q2.head, &head, head + 1u, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                            idx = head;
                            break;
                        }
                    }
                }
                idx = uni(idx);
                if (idx == 0xffffffffu) break;
                uint32_t ti;  // (reserved before it is written: wait for the writer, a wave of this workgroup a few instructions behind)
                do {
                    ti = uni(__hip_atomic_load(&Q2.entry[idx].ti, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                } while (ti == 0xffffffffu);
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                const uint32_t depth = uni(Q2.entry[idx].depth) & 0xffu;
                const uint32_t mine = Q2.entry[idx].atoms[min(lane, (uint32_t)kMaxK - 1u)];  // lane L < depth: the atom of level L
                if (ti != D.ti) {
                    if (D.ti != 0xffffffffu && Q.cost && lane == 0) atomicAdd(&Q.cost[D.ti], __builtin_amdgcn_s_memtime() - D.since);
                    TemplateRegs T;
                    load_template(Q, ti, (int)lane, T);
                    D.feasible = setup_template(C, W, T, kl, Q, ti);
                    D.ti = ti;
                    D.P.ti = ti;
                    D.P.truncated = false;
                    D.since = Q.cost ? __builtin_amdgcn_s_memtime() : 0ull;
                    D.next_ti = 0xffffffffu;
                }
                if (D.P.truncated || !D.feasible) continue;
                {
                    const uint32_t atom = lane < depth ? mine : 0u;
                    const float4 r = C.rec[atom];
                    if (lane < depth) {
                        W.aidx = atom;
                        W.ax = r.x; W.ay = r.y; W.az = r.z;
                        W.ameta = __float_as_uint(r.w);
                        if (!C.ignore_chain) W.achain = (uint32_t)C.qchain[atom];
                    }
                }
                dfs(C, W, D.P, (int)depth, shared);
            }
        }
    };
    // (A structure in global memory -- a large complex: a search below one of its triples takes from a tenth of a
    // millisecond to several -- claims ONE run at a time, when the wave is done with the one before: with two claimed ahead
    // up to four triples waited behind a long search while other waves of the structure had nothing left,
    // tests/gpu_deep_timeline.py.  The round trips of a claim do not show beside such searches.)
    constexpr bool kAhead = kLds;
    uint32_t run = 0;
    if (lane == 0) run = atomicAdd(&A.O.dir_head[s], kAhead ? 2u : 1u);
    run = uni(run);
    uint32_t run1 = kAhead ? run + 1u : 0xffffffffu;
    DeepEntry t;
    uint32_t nt = fetch(run, t);
    while (run < n_runs) {
        uint32_t claimed = 0;
        if (kAhead && lane == 0) claimed = atomicAdd(&A.O.dir_head[s], 1u);
        DeepEntry tn;
        const uint32_t nn = fetch(run1, tn);
        for (uint32_t q = 0; q < nt; ++q) {
            const uint32_t a01 = RL(t.a01, (int)q), c = RL(t.c, (int)q), ti = RL(t.ti, (int)q);
            const uint32_t x0 = a01 & 0xffffu, x1 = a01 >> 16;
#if defined(TM_STATS) || defined(TM_ABLATE)
            if (Q.flags & 0x2000u) continue;  // phase timing: runs claimed and read, nothing else
#endif
            if (ti != D.ti) {  // a triple of another template: its constants take over W
                if (D.ti != 0xffffffffu && Q.cost && lane == 0) atomicAdd(&Q.cost[D.ti], __builtin_amdgcn_s_memtime() - D.since);
                TemplateRegs T;
                if (D.next_ti == ti) T = D.next;
                else load_template(Q, ti, (int)lane, T);
                D.feasible = setup_template(C, W, T, kl, Q, ti);
                D.ti = ti;
                D.P.ti = ti;
                // (whether another wave already met the pair's max_candidates limit is not looked up here -- two dependent
                // loads from HBM per triple; a wave learns it with its own first completion, on_complete)
                D.P.truncated = false;
                D.since = Q.cost ? __builtin_amdgcn_s_memtime() : 0ull;
                // the next template starts loading now: the one of a later triple of this run, or else of the next run
                const unsigned long long later = ballot64(lane > q && lane < nt && t.ti != ti);
                D.next_ti = 0xffffffffu;
                if (later) {
                    D.next_ti = RL(t.ti, __ffsll((long long)later) - 1);
                    load_template(Q, D.next_ti, (int)lane, D.next);
                } else {
                    const unsigned long long beyond = ballot64(lane < nn && tn.ti != ti);
                    if (beyond) {
                        D.next_ti = RL(tn.ti, __ffsll((long long)beyond) - 1);
                        load_template(Q, D.next_ti, (int)lane, D.next);
                    }
                }
            }
            if (D.P.truncated || !D.feasible) continue;
#if defined(TM_STATS) || defined(TM_ABLATE)
            if (Q.flags & 0x1000u) continue;  // phase timing: template constants set up, no search
#endif
            {  // the three placed leads: lane L < 3 fetches and keeps the entry of level L (lane = level)
                const uint32_t mine = lane == 0u ? x0 : lane == 1u ? x1 : c;
                const float4 r = C.rec[mine];
                if (lane < 3u) {
                    W.aidx = mine;
                    W.ax = r.x; W.ay = r.y; W.az = r.z;
                    W.ameta = __float_as_uint(r.w);
                    if (!C.ignore_chain) W.achain = (uint32_t)C.qchain[mine];
                }
            }
            TM_STAT(84 + min(C.n_groups, 8), 1);  // [87..92]: paged triples by residue count of their template
#ifdef TM_STATS
            const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
#endif
            dfs(C, W, D.P, 3, kLds ? nullptr : shared);  // false: cut short by max_candidates (D.P.truncated is set)
#ifdef TM_STATS
            if (!kLds) {  // (a structure in global memory: the block search queues nothing, the histogram of entry durations is free)
                const unsigned long long dt = __builtin_amdgcn_s_memrealtime() - t_entry;
                const uint32_t bl = min(23u, 64u - (uint32_t)__clzll((long long)dt));
                TM_STAT(192 + bl, 1);
                TM_STAT(216 + bl, dt);
                TM_STAT(240 + 3, 1);
            }
#endif
        }
        if constexpr (kAhead) {
            run = run1;
            run1 = uni(claimed);
            t = tn;
            nt = nn;
        } else {
            drain();
            if (lane == 0) claimed = atomicAdd(&A.O.dir_head[s], 1u);
            run = uni(claimed);
            nt = fetch(run, t);
        }
    }
    if constexpr (!kLds) {
        // out of runs: wait for what the waves still at work hand over until none of them is left (search_templates)
        if (shared != nullptr) {
            SharedTriples &Q2 = *shared;
            drain();
            if (lane == 0u) atomicSub(&Q2.busy, 1u);
            for (;;) {
                const uint32_t head = uni(__hip_atomic_load(&Q2.head, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                const uint32_t tail = min(uni(__hip_atomic_load(&Q2.tail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)), kSharedTriples);
                if (head < tail) {
                    if (lane == 0u) atomicAdd(&Q2.busy, 1u);
                    drain();
                    if (lane == 0u) atomicSub(&Q2.busy, 1u);
                    continue;
                }
                if (uni(__hip_atomic_load(&Q2.busy, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == 0u) break;
                __builtin_amdgcn_s_sleep(16);
            }
        }
    }
    if (D.ti != 0xffffffffu && Q.cost && lane == 0) atomicAdd(&Q.cost[D.ti], __builtin_amdgcn_s_memtime() - D.since);
    slab_release(C);
}

// All three launch modes (kModeBlocks, kModeSingly, kModeDeep) stage the structure of their work item in LDS the same way.
template <bool kLds, bool kIgnoreChain, int kMode>
#ifdef TM_SEARCH_MIN_WAVES
__global__ __launch_bounds__(kSearchBlock, TM_SEARCH_MIN_WAVES) void k_search(SearchArgs A) {
#else
__global__ __launch_bounds__(kSearchBlock) void k_search(SearchArgs A) {
#endif
    constexpr bool kSingly = kMode == kModeSingly;
    const LibView &L = A.L;
    const BatchView &B = A.B;
    const QueryView &Q = A.Q;
    const uint32_t max_atoms = A.max_atoms, max_res = A.max_res,
                   pool_cap = A.pool_cap;
    extern __shared__ __align__(16) unsigned char smem[];
    // [WaveScratchT<kMode> x kSearchWaves][rec float4 x max_atoms][res_start u16 x (max_res+1)]
    // [kind-list offsets u32 x (n_lead+1)][kind-list pool u16 x pool_cap]
    WaveScratchT<kMode> *scratch = reinterpret_cast<WaveScratchT<kMode> *>(smem);
#ifdef TM_POISON_LDS
    // debugging build: the waves' scratch starts from a recognisable pattern instead of what the last kernel left on this CU
    for (uint32_t i = threadIdx.x; i < sizeof(WaveScratchT<kMode>) * kSearchWaves / 4; i += blockDim.x) reinterpret_cast<uint32_t *>(smem)[i] = 0xfffefffeu;
    __syncthreads();
#endif
    float4 *s_rec = reinterpret_cast<float4 *>(smem + sizeof(WaveScratchT<kMode>) * kSearchWaves);
    uint16_t *s_res = reinterpret_cast<uint16_t *>(s_rec + (kLds ? max_atoms : 0));
    uint32_t *s_kl = reinterpret_cast<uint32_t *>(reinterpret_cast<unsigned char *>(s_res) +
                                                  (kLds ? search_res_bytes(max_res) : 0));
    uint16_t *s_pool = reinterpret_cast<uint16_t *>(s_kl + (kLds ? search_kl_words(L.n_lead) : 0));
    // k_deep on a structure that is not in LDS: the residue cache takes the place of the structure's copy
    ResCache *cache = (!kLds && kMode == kModeDeep) ? reinterpret_cast<ResCache *>(s_rec) : nullptr;
    (void)cache;
    __shared__ uint32_t s_next;

    // Workgroups b and b + 8 share an XCD.  The template selection of a work item is cut into
    // (rounds << cx) interleaved chunks -- chunk c holds the selected templates c, c + n_chunks, ..., so every
    // chunk gets the same mix of cheap and expensive templates -- and an XCD only ever works on the chunks
    // whose low cx bits equal its own: its share of the staged template table (about 1 KB per template) stays
    // resident in its own 4 MB L2.  The structure a workgroup searches is copied into LDS once, so it does not
    // matter which XCD that is.  Work items are ordered heaviest structure first, and heavy structures are cut
    // into more rounds (host side), so that a few large structures in a batch do not become the tail.
    // The QUEUED block search (structures in LDS; A.queue): the launch is one workgroup per CU and every workgroup takes
    // units -- (work item, chunk of its XCD's class) -- from its class's counter until none are left, then from the
    // other classes'.  A grid of one workgroup per unit left a CU idle between two workgroups (9 us median, 68 us at
    // the 90th percentile: a 160 KB workgroup waits for a whole CU of the shader engine whose turn it is) and the four
    // XCDs of the lighter class idle at the end: 76 % of the CU-time between the first start and the last end was held
    // by a workgroup (tests/gpu_wg_timeline.py).
    constexpr bool kQueued = kMode == kModeBlocks && kLds;
    __shared__ uint32_t s_claim;
    const uint32_t cx = A.chunk_xcd_shift;
    for (;;) {
#ifdef TM_STATS
    const unsigned long long t_wg = __builtin_amdgcn_s_memrealtime();
#endif
    uint32_t item_index, chunk_low;
    if constexpr (kQueued) {
        // (what the claim needs is read from the kernel-argument segment again: nothing of it stays in registers through
        // the search of a unit)
        __syncthreads();  // the previous unit's searches are over: its structure may leave LDS, s_claim be written again
        if (threadIdx.x == 0) {
            const ColdArgsPtr cold = cold_args();
            uint32_t *queue = cold->queue;
            const uint32_t n_items = cold->n_items, mask = (1u << cold->chunk_xcd_shift) - 1u, own = blockIdx.x & mask;
            uint32_t got = 0xffffffffu;
            for (uint32_t k = 0; k <= mask && got == 0xffffffffu; ++k) {
                const uint32_t c = (own + k) & mask;
                // (look before adding: the counter of a class that ran dry does not run away)
                if (__hip_atomic_load(&queue[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_items) continue;
                const uint32_t v = atomicAdd(&queue[c], 1u);
                if (v < n_items) got = (v << 3) | c;
            }
            s_claim = got;
        }
        __syncthreads();
        const uint32_t got = s_claim;
        if (got == 0xffffffffu) return;
        item_index = got >> 3;
        chunk_low = got & 7u;
    } else {
        // one unit per workgroup: workgroups b and b + 8 share an XCD, workgroup b takes the chunk class b & mask
        const uint32_t bid = blockIdx.x, xcd = bid & 7u;
        item_index = ((bid >> 3) << (3u - cx)) + (xcd >> cx);
        chunk_low = xcd & ((1u << cx) - 1u);
    }
    if (item_index >= A.n_items) { if constexpr (kQueued) continue; else return; }
#ifdef TM_ABLATE
    const uint32_t log_at = (item_index << cx) + chunk_low;
    WgLog *const wg_log = kMode == kModeDeep ? g_deep_log : g_wg_log;
    if (kMode != kModeSingly && threadIdx.x == 0 && log_at < kWgLog) {
        uint32_t hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        wg_log[log_at].start = __builtin_amdgcn_s_memrealtime();
        wg_log[log_at].hw_id = hw;
        wg_log[log_at].xcc_id = xcc;
    }
#endif
    const WorkItem item = A.work[item_index];
    const uint32_t s = item.s;
    const uint32_t chunk = ((uint32_t)item.round << cx) + chunk_low;
    const uint32_t n_chunks = (uint32_t)item.rounds << cx;
    if (chunk >= (kSingly ? A.n_singly : Q.n_sel)) { if constexpr (kQueued) continue; else return; }

    if (B.tab_ok[s] != 1u) { if constexpr (kQueued) continue; else return; }  // pair tables incomplete (pool overflow, reported with the result): nothing to search
    uint32_t n_listed = 0;
    if (kMode == kModeDeep) {  // pages of triples the block search left for this structure (its share of the directory)
        n_listed = min(A.O.dir_n[s], A.O.dir_off[s + 1] - A.O.dir_off[s]);
        if (n_listed == 0u) { if constexpr (kQueued) continue; else return; }
        // a structure with few triples is not worth all its workgroups (each stages the structure in LDS and queues at the
        // structure's counter): as many take part as there are 8 triples per wave for
        if ((unsigned long long)chunk * (kDeepShare * kSearchWaves) >= A.O.dir_triples[s]) { if constexpr (kQueued) continue; else return; }
    }
    const uint32_t a0 = B.s_off[s], n = B.s_off[s + 1] - a0;
    // A wide structure (more than kNarrowAtoms atoms: 24-bit atom indices) is searched by ONE launch -- the
    // template-at-a-time search on global memory over the wide structures alone (A.wide_only) -- and passed over by every other.
    if ((n > kNarrowAtoms) != (A.wide_only != 0u)) { if constexpr (kQueued) continue; else return; }
    const uint32_t r0 = B.s_roff[s], nres1 = B.s_roff[s + 1] - r0;  // n_res + 1 entries
    const uint32_t *kl_off = B.kl_off + (size_t)s * (L.n_lead + 1);
    bool pool_in_lds = false;
    if (kLds) {
        // the structure's atom records, residue starts, kind-list offsets and (when it fits) the
        // kind lists themselves: after this the search reads global memory only for template data
        // Every load of the first pass is issued before the first LDS store -- one trip to L2 for the four arrays together,
        // where a loop per array waits once per iteration (the kind lists, 2 bytes per lane, took seven) --; the kind
        // lists travel 16 bytes per lane: from the 16-byte boundary below their first entry, the offsets rebased to that.
        const uint32_t p0 = kl_off[0], plen = kl_off[L.n_lead] - p0;
        pool_in_lds = plen + 7u <= pool_cap;
        const uint32_t rebase = pool_in_lds ? p0 & ~7u : 0u;
        uint32_t tid = threadIdx.x;
        asm volatile("" : "+v"(tid));  // (per unit: see load_template)
        const uint32_t step = blockDim.x;
        const uint4 *g_rec = reinterpret_cast<const uint4 *>(B.q_rec + a0);
        const uint4 *g_pool = reinterpret_cast<const uint4 *>(B.kl_pool + rebase);
        uint4 *l_rec = reinterpret_cast<uint4 *>(s_rec), *l_pool = reinterpret_cast<uint4 *>(s_pool);
        const uint32_t n_pool = pool_in_lds ? (p0 - rebase + plen + 7u) / 8u : 0u;  // 16-byte words of kind lists
        constexpr uint32_t kRecPer = 3, kPoolPer = 1;
        // (named registers, every one initialised: an array indexed in a loop went to scratch memory -- 64 bytes per
        // lane through L2, +75 MB of writes per launch)
        const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
        uint4 v_rec0 = zero4, v_rec1 = zero4, v_rec2 = zero4, v_pool0 = zero4;
        uint16_t v_res = 0;
        uint32_t v_kl = 0;
        if (tid < n) v_rec0 = g_rec[tid];
        if (tid + step < n) v_rec1 = g_rec[tid + step];
        if (tid + 2u * step < n) v_rec2 = g_rec[tid + 2u * step];
        if (tid < n_pool) v_pool0 = g_pool[tid];
        if (tid < nres1) v_res = B.res_start[r0 + tid];
        if (tid <= L.n_lead) v_kl = kl_off[tid];
        if (tid < n) l_rec[tid] = v_rec0;
        if (tid + step < n) l_rec[tid + step] = v_rec1;
        if (tid + 2u * step < n) l_rec[tid + 2u * step] = v_rec2;
        if (tid < n_pool) l_pool[tid] = v_pool0;
        if (tid < nres1) s_res[tid] = v_res;
        if (tid <= L.n_lead) s_kl[tid] = v_kl - rebase;
        // what a structure has beyond the first pass
        for (uint32_t i = tid + kRecPer * step; i < n; i += step) l_rec[i] = g_rec[i];
        for (uint32_t i = tid + kPoolPer * step; i < n_pool; i += step) l_pool[i] = g_pool[i];
        for (uint32_t i = tid + step; i < nres1; i += step) s_res[i] = B.res_start[r0 + i];
        for (uint32_t i = tid + step; i <= L.n_lead; i += step) s_kl[i] = kl_off[i] - rebase;
    }
    if (threadIdx.x == 0) s_next = 0u;
    // the block search's queue of triples for the depth-first search (static LDS of that mode only)
    SharedTriples *shared = nullptr;
    // (... and of k_deep on a structure in global memory: a large complex's last waves sit in searches of thousands of
    // evaluations -- two interchangeable atoms per residue, eight residues -- while the other waves of their workgroups have
    // run out of triples: the same hand-over among the sixteen waves of a workgroup, deep_templates)
    if constexpr (kMode == kModeBlocks || (kMode == kModeDeep && !kLds)) {
        __shared__ SharedTriples s_shared;
        shared = &s_shared;
        for (uint32_t i = threadIdx.x; i < kSharedTriples; i += blockDim.x) s_shared.entry[i].ti = 0xffffffffu;
        // (threads 0..2, one word each: as one 12-byte store of a constant the value was kept in three registers through
        // the whole kernel -- and spilled)
        if (threadIdx.x < 3u) (&s_shared.head)[threadIdx.x] = threadIdx.x == 2u ? (uint32_t)kSearchWaves : 0u;
        static_assert(offsetof(SharedTriples, tail) == offsetof(SharedTriples, head) + 4 && offsetof(SharedTriples, busy) == offsetof(SharedTriples, head) + 8, "head, tail, busy");
    }
    TM_STAT(55, threadIdx.x == 0 ? 1 : 0);
    TM_STAT(56, threadIdx.x == 0 && pool_in_lds ? 1 : 0);
    TM_STAT(57, threadIdx.x == 0 ? kl_off[L.n_lead] - kl_off[0] : 0);
    __syncthreads();

#ifdef TM_STATS
    // how long the workgroup's waves work against how long the workgroup holds its CU: [70 + 4 mode] sum of the waves'
    // working ticks, [71 + 4 mode] sum over workgroups of (last wave's end - start) x waves, [72 + 4 mode] staging ticks x waves
    __shared__ unsigned long long s_end;
    const unsigned long long t_work = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) s_end = 0ull;
    __syncthreads();
#endif
    if constexpr (kMode == kModeDeep) {
        // The workgroup's run list: entry n = where the n-th run of triples of this structure sits (first entry among all
        // pages | triples in it - 1), over the runs that have triples.  Built when the structure lists at most one page
        // per thread and kRunLds runs; a structure beyond that (a large complex: its pages are full) is claimed by
        // page position (deep_templates).
        __shared__ uint32_t s_run[kRunLds];
        __shared__ uint32_t s_wtot[kSearchWaves], s_wide[kSearchWaves];
        uint32_t n_compact = 0xffffffffu;
        if (n_listed <= kSearchBlock && !(Q.flags & 0x4000u)) {  // (flag: test hook ENZYMM_AMD_RUN_LIST=0)
            uint2 d = make_uint2(0u, 0u);
            if (threadIdx.x < n_listed) d = A.O.dir[A.O.dir_off[s] + threadIdx.x];
            const uint32_t mine = (d.y + kDeepRun - 1u) / kDeepRun;
            const uint32_t incl = wave_scan_add(mine);
            const uint32_t wave = threadIdx.x >> 6;
            const bool wave_wide = ballot64(d.x >= (1u << 26)) != 0ull;  // (an entry position must fit 32 bits)
            if ((threadIdx.x & 63u) == 63u) { s_wtot[wave] = incl; s_wide[wave] = wave_wide ? 1u : 0u; }
            __syncthreads();
            uint32_t before = 0, total = 0, wide = 0;
            for (uint32_t w = 0; w < kSearchWaves; ++w) {
                const uint32_t v = s_wtot[w];
                total += v;
                before += w < wave ? v : 0u;
                wide |= s_wide[w];
            }
            if (total <= kRunLds && !wide) {
                const uint32_t at = before + incl - mine;
                for (uint32_t j = 0; j < mine; ++j)
                    s_run[at + j] = (d.x * kPageTriples + j * kDeepRun) | (min(kDeepRun, d.y - j * kDeepRun) - 1u);
                n_compact = total;
            }
            __syncthreads();
        }
        if (pool_in_lds) deep_templates<kLds, kIgnoreChain, true>(A, scratch, s_rec, s_res, s_kl, s_pool, kl_off, s, a0, r0, n_listed, cache, s_run, n_compact, shared);
        else deep_templates<kLds, kIgnoreChain, false>(A, scratch, s_rec, s_res, s_kl, s_pool, kl_off, s, a0, r0, n_listed, cache, s_run, n_compact, shared);
    } else {
        if (pool_in_lds)
            search_templates<kLds, kIgnoreChain, true, kSingly>(A, scratch, s_rec, s_res, s_kl, s_pool, kl_off, &s_next, s, a0, r0, chunk, n_chunks, shared);
        else
            search_templates<kLds, kIgnoreChain, false, kSingly>(A, scratch, s_rec, s_res, s_kl, s_pool, kl_off, &s_next, s, a0, r0, chunk, n_chunks, shared);
    }
#ifdef TM_ABLATE
    if (kMode != kModeSingly && (threadIdx.x & 63) == 0 && log_at < kWgLog) {
        const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
        atomicMax(&wg_log[log_at].end, t_end);
        atomicMin(&wg_log[log_at].first_end, t_end);
        atomicAdd(&wg_log[log_at].sum_end, t_end);
    }
#endif
#ifdef TM_STATS
    {
        const unsigned long long t_done = __builtin_amdgcn_s_memrealtime();
        TM_STAT(70 + 4 * kMode, t_done - t_work);
        if ((threadIdx.x & 63) == 0) atomicMax(&s_end, t_done);
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(&g_stats[71 + 4 * kMode], (s_end - t_work) * (unsigned long long)kSearchWaves);
            atomicAdd(&g_stats[72 + 4 * kMode], (t_work - t_wg) * (unsigned long long)kSearchWaves);
        }
    }
#endif
    if constexpr (!kQueued) return;
    else if (cold_args()->queue_once != 0u) return;  // (ENZYMM_AMD_QUEUED=0: one unit per workgroup, in launch order)
    }  // the next unit of the queue
}

#define TM_INSTANTIATE(mode)                                       \
    template __global__ void k_search<true, true, mode>(SearchArgs);   \
    template __global__ void k_search<true, false, mode>(SearchArgs);  \
    template __global__ void k_search<false, true, mode>(SearchArgs);  \
    template __global__ void k_search<false, false, mode>(SearchArgs);
TM_INSTANTIATE(kModeBlocks)
TM_INSTANTIATE(kModeSingly)
TM_INSTANTIATE(kModeDeep)
#undef TM_INSTANTIATE

size_t search_cache_bytes() { return sizeof(ResCache) * kSearchWaves; }
// LDS a launch mode declares statically, beyond the few words every mode has (k_deep's run list)
size_t search_static_bytes(int mode) {
    return mode == kModeDeep ? (kRunLds + 2 * kSearchWaves) * sizeof(uint32_t) : mode == kModeBlocks ? sizeof(SharedTriples) + 16 : 0;
}
size_t search_scratch_bytes(int mode) {
    return (mode == kModeBlocks ? sizeof(WaveScratchT<kModeBlocks>) : mode == kModeSingly ? sizeof(WaveScratchT<kModeSingly>)
                                                                                         : sizeof(WaveScratchT<kModeDeep>)) * kSearchWaves;
}

// ------------------------------------------------------------------ superposition (fp64)

// One Jacobi rotation on the symmetric 4x4 matrix A (rows/cols P,Q), accumulating V.
template <int P, int Q>
__device__ __forceinline__ void jacobi_rot(double (&A)[4][4], double (&V)[4][4]) {
    const double apq = A[P][Q];
    if (fabs(apq) < 1e-300) return;
    const double theta = (A[Q][Q] - A[P][P]) / (2.0 * apq);
    const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double aip = A[i][P], aiq = A[i][Q];
        A[i][P] = c * aip - s * aiq;
        A[i][Q] = s * aip + c * aiq;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double api = A[P][i], aqi = A[Q][i];
        A[P][i] = c * api - s * aqi;
        A[Q][i] = s * api + c * aqi;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double vip = V[i][P], viq = V[i][Q];
        V[i][P] = c * vip - s * viq;
        V[i][Q] = s * vip + c * viq;
    }
}

// Horn (1987) closed-form absolute orientation: the optimal proper rotation taking the
// centred target atoms onto the centred template atoms is the unit quaternion that is the
// dominant eigenvector of the 4x4 matrix N built from the cross-covariance S.
// In: S[a][b] = sum q'_a p'_b.  Out: row-major rotation.
__device__ void horn_rotation(const double S[3][3], double R[9]) {
    double A[4][4], V[4][4];
    const double Sxx = S[0][0], Sxy = S[0][1], Sxz = S[0][2];
    const double Syx = S[1][0], Syy = S[1][1], Syz = S[1][2];
    const double Szx = S[2][0], Szy = S[2][1], Szz = S[2][2];
    A[0][0] = Sxx + Syy + Szz; A[0][1] = Syz - Szy;       A[0][2] = Szx - Sxz;        A[0][3] = Sxy - Syx;
    A[1][0] = A[0][1];         A[1][1] = Sxx - Syy - Szz; A[1][2] = Sxy + Syx;        A[1][3] = Szx + Sxz;
    A[2][0] = A[0][2];         A[2][1] = A[1][2];         A[2][2] = -Sxx + Syy - Szz; A[2][3] = Syz + Szy;
    A[3][0] = A[0][3];         A[3][1] = A[1][3];         A[3][2] = A[2][3];          A[3][3] = -Sxx - Syy + Szz;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[0][3] * A[0][3] + A[1][2] * A[1][2] +
                           A[1][3] * A[1][3] + A[2][3] * A[2][3];
        const double dia = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2] + A[3][3] * A[3][3];
        if (off <= 1e-32 * dia || off < 1e-300) break;
        jacobi_rot<0, 1>(A, V); jacobi_rot<0, 2>(A, V); jacobi_rot<0, 3>(A, V);
        jacobi_rot<1, 2>(A, V); jacobi_rot<1, 3>(A, V); jacobi_rot<2, 3>(A, V);
    }
    int m = 0;
    double best = A[0][0];
    if (A[1][1] > best) { best = A[1][1]; m = 1; }
    if (A[2][2] > best) { best = A[2][2]; m = 2; }
    if (A[3][3] > best) { best = A[3][3]; m = 3; }
    double w = m == 0 ? V[0][0] : m == 1 ? V[0][1] : m == 2 ? V[0][2] : V[0][3];
    double x = m == 0 ? V[1][0] : m == 1 ? V[1][1] : m == 2 ? V[1][2] : V[1][3];
    double y = m == 0 ? V[2][0] : m == 1 ? V[2][1] : m == 2 ? V[2][2] : V[2][3];
    double z = m == 0 ? V[3][0] : m == 1 ? V[3][1] : m == 2 ? V[3][2] : V[3][3];
    const double nrm = sqrt(w * w + x * x + y * y + z * z);
    w /= nrm; x /= nrm; y /= nrm; z /= nrm;
    R[0] = w * w + x * x - y * y - z * z; R[1] = 2.0 * (x * y - w * z);         R[2] = 2.0 * (x * z + w * y);
    R[3] = 2.0 * (x * y + w * z);         R[4] = w * w - x * x + y * y - z * z; R[5] = 2.0 * (y * z - w * x);
    R[6] = 2.0 * (x * z - w * y);         R[7] = 2.0 * (y * z + w * x);         R[8] = w * w - x * x - y * y + z * z;
}

// Superposes candidate `c` onto its template: proper rotation R, translation T with
// x_template = R x_target + T, and RMSD = sqrt(sum |R q + T - p|^2 / k) evaluated directly.
// atom i (template order) of candidate c; hi: bits 16-23 of its atoms (a batch with a wide structure) or NULL
__device__ __forceinline__ uint32_t cand_atom(const Cand &c, const uint8_t *hi, int i) {
    return hi ? (uint32_t)c.atoms[i] | ((uint32_t)hi[i] << 16) : (uint32_t)c.atoms[i];
}
__device__ __forceinline__ const uint8_t *cand_hi(const OutView &O, const Cand &c) {
    return O.cands_hi ? O.cands_hi + (size_t)(&c - O.cands) * kMaxK : nullptr;
}

__device__ double superpose_candidate(const LibView &L, const BatchView &B, const QueryView &Q, const Cand &c,
                                      const uint8_t *hi, double R[9], double T[3]) {
    const uint32_t off = L.t_off[Q.sel[c.ti]];
    const Staged &St = Q.staged[c.ti];
    const int k = (int)St.k();
    const double *qx = B.q_xyz + 3 * (size_t)B.s_off[c.s];
    // candidate atoms are stored in TEMPLATE order, library coordinates in SEARCH order
    double qc[3] = {0, 0, 0}, pc[3] = {0, 0, 0};
    for (int pos = 0; pos < k; ++pos) {
        const double *q = qx + 3 * (size_t)cand_atom(c, hi, St.orig[pos]);
        const double *p = L.a_xyz + 3 * (size_t)(off + pos);
        for (int a = 0; a < 3; ++a) { qc[a] += q[a]; pc[a] += p[a]; }
    }
    const double inv_k = 1.0 / (double)k;
    for (int a = 0; a < 3; ++a) { qc[a] *= inv_k; pc[a] *= inv_k; }
    double Sm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int pos = 0; pos < k; ++pos) {
        const double *q = qx + 3 * (size_t)cand_atom(c, hi, St.orig[pos]);
        const double *p = L.a_xyz + 3 * (size_t)(off + pos);
        const double qv[3] = {q[0] - qc[0], q[1] - qc[1], q[2] - qc[2]};
        const double pv[3] = {p[0] - pc[0], p[1] - pc[1], p[2] - pc[2]};
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) Sm[a][b] += qv[a] * pv[b];
    }
    horn_rotation(Sm, R);
    for (int a = 0; a < 3; ++a) T[a] = pc[a] - (R[3 * a] * qc[0] + R[3 * a + 1] * qc[1] + R[3 * a + 2] * qc[2]);
    double sum = 0.0;
    for (int pos = 0; pos < k; ++pos) {
        const double *q = qx + 3 * (size_t)cand_atom(c, hi, St.orig[pos]);
        const double *p = L.a_xyz + 3 * (size_t)(off + pos);
        const double ex = R[0] * q[0] + R[1] * q[1] + R[2] * q[2] + T[0] - p[0];
        const double ey = R[3] * q[0] + R[4] * q[1] + R[5] * q[2] + T[1] - p[1];
        const double ez = R[6] * q[0] + R[7] * q[1] + R[8] * q[2] + T[2] - p[2];
        sum += ex * ex + ey * ey + ez * ez;
    }
    return sqrt(sum * inv_k);
}

__global__ __launch_bounds__(256) void k_superpose(LibView L, BatchView B, QueryView Q, OutView O) {
    unsigned long long n = *O.n_cands;
    if (n > O.cand_capacity) n = O.cand_capacity;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += stride) {
        double R[9], T[3];
        const Cand &cand = O.cands[c];
        if (cand.s == 0xffffffffu) continue;  // a reserved slot no candidate was written to
        const double r = superpose_candidate(L, B, Q, cand, cand_hi(O, cand), R, T);
        O.cand_rmsd[c] = r;
        // (a candidate that may become a hit keeps its superposition: k_select repeated the 40 us of it for every hit)
        if (c < O.cand_fit_capacity && r <= Q.rmsd_thr[cand.ti]) {
            double *fit = O.cand_fit + 12 * c;
            for (int a = 0; a < 9; ++a) fit[a] = R[a];
            for (int a = 0; a < 3; ++a) fit[9 + a] = T[a];
        }
        // the pair's minimum within the threshold (non-negative doubles order like their bit patterns)
        if (O.pair_min && r <= Q.rmsd_thr[cand.ti])
            atomicMin(&O.pair_min[(size_t)cand.s * Q.n_sel + Q.order[cand.ti]], (unsigned long long)__double_as_longlong(r));
    }
}

// best_match: the candidates within kTieEps of their pair's minimum RMSD are linked into the pair's tie list (one or two
// per pair; a homo-oligomer built by symmetry operators gives a few more).  One lane per candidate.
__global__ __launch_bounds__(256) void k_ties(QueryView Q, OutView O) {
    unsigned long long n = *O.n_cands;
    if (n > O.cand_capacity) n = O.cand_capacity;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long c = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += stride) {
        Cand &cand = O.cands[c];
        if (cand.s == 0xffffffffu) continue;
        const double r = O.cand_rmsd[c];
        const size_t pair = (size_t)cand.s * Q.n_sel + Q.order[cand.ti];
        const unsigned long long low = O.pair_min[pair];
        if (low == ~0ull || !(r <= Q.rmsd_thr[cand.ti]) || !(r <= __longlong_as_double((long long)low) + kTieEps)) continue;
        cand.tie_next = atomicExch(&O.pair_tie[pair], (uint32_t)c + 1u);
    }
}

// true when tuple a (TEMPLATE order, file-order atom indices) is lexicographically smaller than b
__device__ __forceinline__ bool tuple_less(const OutView &O, const uint32_t *orig, const Cand &a, const Cand &b, int k) {
    const uint8_t *ha = cand_hi(O, a), *hb = cand_hi(O, b);
    for (int i = 0; i < k; ++i) {
        const uint32_t x = orig[cand_atom(a, ha, i)], y = orig[cand_atom(b, hb, i)];
        if (x != y) return x < y;
    }
    return false;
}

__device__ void emit_hit(const LibView &L, const BatchView &B, const QueryView &Q, const OutView &O,
                         const PairEntry &pe, const Cand &c, int k) {
    const unsigned long long slot = atomicAdd(O.n_hits, 1ull);
    if (slot >= O.hit_capacity) {
        atomicOr(O.overflow, 4u);
        return;
    }
    tm_hit *h = O.hits + slot;
    double R[9], T[3];
    const uint8_t *hi = cand_hi(O, c);
    const unsigned long long ci = (unsigned long long)(&c - O.cands);
    if (ci < O.cand_fit_capacity) {  // (kept by k_superpose: an emitted candidate lies within its threshold)
        const double *fit = O.cand_fit + 12 * ci;
        for (int a = 0; a < 9; ++a) R[a] = fit[a];
        for (int a = 0; a < 3; ++a) T[a] = fit[9 + a];
        h->rmsd = O.cand_rmsd[ci];
    } else h->rmsd = superpose_candidate(L, B, Q, c, hi, R, T);
    for (int a = 0; a < 9; ++a) h->rot[a] = R[a];
    for (int a = 0; a < 3; ++a) h->trans[a] = T[a];
    h->structure = pe.s;
    h->template_sel = Q.order[pe.ti];
    const uint32_t count = O.pair_count[(size_t)pe.s * Q.n_sel + Q.order[pe.ti]];
    h->n_candidates = count & 0x7fffffffu;
    h->flags = (count >> 31) ? TM_HIT_TRUNCATED : 0u;
    h->n_atoms = (uint32_t)k;
    h->reserved = 0;
    const uint32_t a0 = B.s_off[pe.s];
    for (int i = 0; i < kMaxK; ++i) h->atoms[i] = i < k ? B.q_orig[a0 + cand_atom(c, hi, i)] : 0xffffffffu;

    // ---- fused epilogue: residue-orientation angle and logistic filter
    // (what enzymm computes per hit in Python: jess_run.py:425-478 and 298-346)
    double orientation = __longlong_as_double(0x7ff8000000000000ll);  // NaN
    const uint32_t t = Q.sel[pe.ti];
    if (L.r_off && L.r_off[t + 1] > L.r_off[t] && 3u * (L.r_off[t + 1] - L.r_off[t]) == (uint32_t)k) {
        const uint32_t r0 = L.r_off[t], nres = L.r_off[t + 1] - r0;
        const double *qx = B.q_xyz + 3 * (size_t)a0;
        double acc = 0.0;
        for (uint32_t r = 0; r < nres; ++r) {
            double p[3][3];  // the residue's three matched atoms in the template frame
            for (int n = 0; n < 3; ++n) {
                const double *q = qx + 3 * (size_t)cand_atom(c, hi, (int)(3 * r) + n);
                for (int a = 0; a < 3; ++a) p[n][a] = R[3 * a] * q[0] + R[3 * a + 1] * q[1] + R[3 * a + 2] * q[2] + T[a];
            }
            const int fi = L.r_first[r0 + r] < 3 ? L.r_first[r0 + r] : 0, se = L.r_second[r0 + r];
            double v[3];
            if (se == 9) {
                const int s1 = fi == 0 ? 1 : 0, s2 = fi == 2 ? 1 : 2;
                for (int a = 0; a < 3; ++a) v[a] = (p[s1][a] + p[s2][a]) / 2 - p[fi][a];
            } else {
                const int sj = se < 3 ? se : 0;
                for (int a = 0; a < 3; ++a) v[a] = p[sj][a] - p[fi][a];
            }
            const double *tv = L.r_vec + 3 * (size_t)(r0 + r);
            const double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
            const double nt = sqrt(tv[0] * tv[0] + tv[1] * tv[1] + tv[2] * tv[2]);
            double a3[3] = {tv[0], tv[1], tv[2]}, b3[3] = {v[0], v[1], v[2]};
            if (nt != 0.0) for (int a = 0; a < 3; ++a) a3[a] = tv[a] / nt;
            if (nv != 0.0) for (int a = 0; a < 3; ++a) b3[a] = v[a] / nv;
            double dot = a3[0] * b3[0] + a3[1] * b3[1] + a3[2] * b3[2];
            if (dot > 1.0 && dot <= 1.0 + 1e-5) dot = 1.0;     // Vec3.angle_to's tolerance, template.py:157-181
            if (dot < -1.0 && dot >= -1.0 - 1e-5) dot = -1.0;
            acc += acos(dot);
        }
        orientation = acc / (double)nres;
    }
    h->orientation = orientation;
    if (Q.model_set && orientation == orientation) {
        const int m = Q.model_set[pe.ti];
        bool correct = true;
        if (m >= 0) {
            uint32_t votes = 0;
            for (uint32_t i = Q.model_offsets[m]; i < Q.model_offsets[m + 1]; ++i) {
                const double *mo = Q.models + 4 * (size_t)i;
                const double z = mo[2] + mo[0] * h->rmsd + mo[1] * orientation;
                votes += (1.0 / (1.0 + exp(-z))) >= mo[3];
            }
            correct = votes >= Q.model_votes[m];
        }
        h->flags |= TM_HIT_PREDICTED | (correct ? TM_HIT_CORRECT : 0u);
    }
}

// One lane per (structure, template) pair with candidates: walk the pair's candidate chain.
// best_match: keep the minimum-RMSD candidate (ties within kTieEps: smaller atom tuple) and emit it;
// otherwise emit every candidate whose RMSD is within the threshold.
__global__ __launch_bounds__(256) void k_select(LibView L, BatchView B, QueryView Q, OutView O) {
    unsigned long long n = *O.n_pairs;
    if (n > O.pair_capacity) n = O.pair_capacity;
    const bool best_match = (Q.flags & TM_FLAG_BEST_MATCH) != 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const PairEntry pe = O.pairs[e];
        const double thr = Q.rmsd_thr[pe.ti];
        const int k = (int)Q.staged[pe.ti].k();
        const uint32_t *orig = B.q_orig + B.s_off[pe.s];
        const uint32_t last = O.pair_last[(size_t)pe.s * Q.n_sel + Q.order[pe.ti]] - 1u;  // 0 - 1: no candidate
        if (!best_match) {
            for (uint32_t c = last; c != 0xffffffffu; c = O.cands[c].prev)
                if (O.cand_rmsd[c] <= thr) emit_hit(L, B, Q, O, pe, O.cands[c], k);
            continue;
        }
        // The candidates within kTieEps of the pair's GLOBAL minimum RMSD were listed by k_ties; the smallest atom tuple
        // among them wins.  (A comparison against a moving best while walking the chain depends on the order of the chain:
        // of three candidates pairwise within eps but not all within eps of the minimum, which one survives would change
        // with it -- and a chain of max_candidates candidates is a long walk for one lane.)
        (void)last;
        uint32_t best = 0xffffffffu;
        for (uint32_t c = O.pair_tie[(size_t)pe.s * Q.n_sel + Q.order[pe.ti]]; c != 0u; c = O.cands[c - 1u].tie_next)
            if (best == 0xffffffffu || tuple_less(O, orig, O.cands[c - 1u], O.cands[best], k)) best = c - 1u;
        if (best != 0xffffffffu) emit_hit(L, B, Q, O, pe, O.cands[best], k);
    }
}


// A query's hit rows appended to a table that collects a rank's hits over several queries (tm_result_append_hits).
// ONE workgroup: one reservation, rows copied 8 bytes per lane (a hit table is a few thousand rows of 232 bytes).
static_assert(sizeof(tm_hit) % 8 == 0 && offsetof(tm_hit, structure) == 0, "k_append_hits copies rows as 64-bit words");
// Grid-wide (round 5; one 1024-thread workgroup before: on the critical path of every pass of a sharded run): workgroup b
// takes the b-th slice of the source rows and reserves as many rows of the destination with one atomic -- the rows of a
// pass table are unordered (rank 0 sorts after the gather), so the slices may land in any order.
__global__ __launch_bounds__(256) void k_append_hits(const tm_hit *__restrict__ src, const unsigned long long *__restrict__ n_src,
                                                      unsigned long long src_capacity, tm_hit *__restrict__ dst,
                                                      unsigned long long dst_capacity, unsigned long long *__restrict__ dst_count,
                                                      uint32_t structure_offset) {
    __shared__ unsigned long long s_base;
    unsigned long long n = *n_src;
    if (n > src_capacity) n = src_capacity;
    const unsigned long long per = (n + gridDim.x - 1) / gridDim.x;
    const unsigned long long row0 = per * blockIdx.x;
    if (row0 >= n) return;
    const unsigned long long rows = n - row0 < per ? n - row0 : per;
    if (threadIdx.x == 0) s_base = atomicAdd(dst_count, rows);
    __syncthreads();
    const unsigned long long base = s_base;
    constexpr uint32_t kWords = sizeof(tm_hit) / 8;
    const unsigned long long total = rows * kWords;
    for (unsigned long long w = threadIdx.x; w < total; w += blockDim.x) {
        const unsigned long long row = w / kWords;
        const uint32_t col = (uint32_t)(w % kWords);
        if (base + row >= dst_capacity) continue;
        unsigned long long v = reinterpret_cast<const unsigned long long *>(src + row0 + row)[col];
        // tm_hit.structure is the LOW word of a row's first 8 bytes: the offset is added to that word alone (a sum beyond
        // 2^32 must not carry into template_sel, the high word)
        if (col == 0) v = (v & 0xffffffff00000000ull) | (uint32_t)((uint32_t)v + structure_offset);
        reinterpret_cast<unsigned long long *>(dst + base + row)[col] = v;
    }
}

}  // namespace tmx
