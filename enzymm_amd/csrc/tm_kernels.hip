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
#include "tm_device.hpp"

namespace tmx {

#ifdef TM_STATS
// diagnostic build only: [0..23] chunk evaluations per level, [24..47] survivors per level,
// [48] complete assignments, [49] searches, [50] pair tests (j-loop trips), [51] fp64 rejects,
// [52] joint-scan tiles, [53] joint-scan survivors
__device__ unsigned long long g_stats[64];
#define TM_STAT(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
// [54] searches with >= 1 joint-scan survivor.  Ablation (diagnostic build only): query flag bit 8 skips the
// subtrees below the joint scan, bit 9 skips the joint scan too (template staging only).
#else
#define TM_STAT(i, v) do { } while (0)
#endif

// ------------------------------------------------------------------ wave helpers

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

__device__ __forceinline__ uint32_t uni(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni64(unsigned long long v) {
    uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
    uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

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

// ------------------------------------------------------------------ k_stage_templates

__global__ __launch_bounds__(64) void k_stage_templates(LibView L, uint32_t n_sel, const uint32_t *__restrict__ sel,
                                                         const double *__restrict__ cutoff,
                                                         const double *__restrict__ maxdyn,
                                                         Staged *__restrict__ staged, float2 *__restrict__ bounds) {
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
            S.group[lane] = L.a_group[off + lane];
            S.tchain[lane] = L.a_chain[off + lane];
            S.orig[lane] = L.a_orig[off + lane];
            S.any[lane] = L.k_any[kind];
            S.ref[lane] = ref;
        } else {
            S.namemask[lane] = 0; S.resmask[lane] = 0; S.lead_li[lane] = 0xffffu;
            S.group[lane] = 0xff; S.tchain[lane] = 0; S.orig[lane] = 0; S.any[lane] = 0; S.ref[lane] = 0;
        }
    }
    if (lane == 0) {
        uint8_t g = 0;
        for (int i = 0; i < k; ++i) g = max(g, (uint8_t)(L.a_group[off + i] + 1));
        S.k = (uint8_t)k;
        S.n_groups = g;
    }
    for (int p = lane; p < npairs; p += 64) {
        int i, j;
        pair_of(p, i, j);
        const double *a = L.a_xyz + 3 * (size_t)(off + i), *b = L.a_xyz + 3 * (size_t)(off + j);
        const double dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
        const double dt = sqrt(dx * dx + dy * dy + dz * dz);
        double thr = cut + L.a_w[off + i] + L.a_w[off + j];
        thr = (thr < mx ? thr : mx) + kPrefilterSlack;
        const double lo = dt - thr, hi = dt + thr;
        const float lo2 = lo > 0.0 ? (float)(lo * lo) * (1.0f - 1e-6f) : 0.0f;
        const float hi2 = (float)(hi * hi) * (1.0f + 1e-6f);
        bounds[(size_t)ti * kMaxPairs + p] = make_float2(lo2, hi2);
    }
}

// ------------------------------------------------------------------ k_kind_lists

// One workgroup per structure.  Pass A counts the admissible atoms of every lead kind, a
// bump allocation reserves the structure's slice of the pool, pass B writes the lists.
__global__ __launch_bounds__(256) void k_kind_lists(LibView L, BatchView B, uint32_t *__restrict__ kl_off,
                                                     uint16_t *__restrict__ kl_pool,
                                                     unsigned long long *__restrict__ cursor,
                                                     unsigned long long capacity, uint32_t *__restrict__ overflow) {
    extern __shared__ uint32_t s_cnt[];  // [n_lead + 1]
    __shared__ unsigned long long s_base;
    const uint32_t s = blockIdx.x;
    const uint32_t a0 = B.s_off[s], n = B.s_off[s + 1] - a0;
    const int lane = lane_id(), wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    const float4 *rec = B.q_rec + a0;

    for (uint32_t li = wave; li < L.n_lead; li += nwaves) {
        const uint32_t kind = L.lead_kind[li];
        const uint32_t rm = L.k_resmask[kind];
        const uint64_t nm = L.k_namemask[kind];
        const bool any = L.k_any[kind] != 0;
        uint32_t c = 0;
        for (uint32_t b = 0; b < n; b += 64) {
            const uint32_t i = b + lane;
            bool ok = false;
            if (i < n) {
                const uint32_t m = __float_as_uint(rec[i].w);
                ok = ((nm >> meta_name(m)) & 1ull) && (any || ((rm >> meta_rtype(m)) & 1u));
            }
            c += __popcll(__ballot(ok));
        }
        if (lane == 0) s_cnt[li] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive scan (n_lead is a few hundred)
        uint32_t run = 0;
        for (uint32_t li = 0; li < L.n_lead; ++li) {
            const uint32_t c = s_cnt[li];
            s_cnt[li] = run;
            run += c;
        }
        s_cnt[L.n_lead] = run;
        unsigned long long base = atomicAdd(cursor, (unsigned long long)run);
        if (base + run > capacity) {
            *overflow = 1u;
            base = ~0ull;
        }
        s_base = base;
    }
    __syncthreads();
    const unsigned long long base = s_base;
    uint32_t *off_out = kl_off + (size_t)s * (L.n_lead + 1);
    for (uint32_t li = threadIdx.x; li <= L.n_lead; li += blockDim.x)
        off_out[li] = base == ~0ull ? 0u : (uint32_t)(base + s_cnt[li]);
    if (base == ~0ull) return;

    for (uint32_t li = wave; li < L.n_lead; li += nwaves) {
        const uint32_t kind = L.lead_kind[li];
        const uint32_t rm = L.k_resmask[kind];
        const uint64_t nm = L.k_namemask[kind];
        const bool any = L.k_any[kind] != 0;
        unsigned long long w = base + s_cnt[li];
        for (uint32_t b = 0; b < n; b += 64) {
            const uint32_t i = b + lane;
            bool ok = false;
            if (i < n) {
                const uint32_t m = __float_as_uint(rec[i].w);
                ok = ((nm >> meta_name(m)) & 1ull) && (any || ((rm >> meta_rtype(m)) & 1u));
            }
            const unsigned long long mask = __ballot(ok);
            if (ok) kl_pool[w + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)i;
            w += __popcll(mask);
        }
    }
}

// ------------------------------------------------------------------ k_search

// Per-wave scratch in LDS.  Everything indexed by "level" = position in the search order.
struct __align__(16) WaveScratch {
    float4 arec[kMaxK];               // assigned target atom record per level
    unsigned long long pend[kMaxK];   // survivors of the current chunk not yet descended into
    unsigned long long namemask[kMaxK];
    uint32_t resmask[kMaxK];
    uint32_t aidx[kMaxK];             // assigned target atom (local index) per level
    uint32_t cur[kMaxK];              // next chunk base to evaluate
    uint32_t base[kMaxK];             // base of the chunk `pend` refers to
    uint32_t lo[kMaxK];               // lead: pool offset of the kind list; else: first atom of the residue
    uint32_t cnt[kMaxK];              // number of candidates at this level
    uint16_t surv[kMaxG][64];         // lead levels: candidate atom of every lane of the current chunk
    uint16_t list0[kJointList], list1[kJointList];  // the two top lead lists, staged for the joint scan
    uint8_t group[kMaxK], tchain[kMaxK], orig[kMaxK], any[kMaxK], lead[kMaxK], achain[kMaxK], ref[kMaxK];
};

struct SearchCtx {
    const float4 *rec;          // structure atom records (LDS or global)
    const uint16_t *res_start;  // residue -> first atom (LDS or global)
    const uint16_t *pool;       // kind-list pool
    const uint8_t *qchain;      // global, per atom of this structure
    const float2 *bnd;          // this template's pair bounds (global, read with uniform addresses)
    WaveScratch *S;
    int lane, k;
    bool ignore_chain;
};

// Tests up to 64 candidates for `level`; returns the survivor mask.
__device__ __forceinline__ unsigned long long eval_chunk(const SearchCtx &C, int level, uint32_t base) {
    WaveScratch &S = *C.S;
    const int lane = C.lane;
    const uint32_t cnt = uni(S.cnt[level]);
    const uint32_t lo = uni(S.lo[level]);
    const bool lead = uni((uint32_t)S.lead[level]) != 0;
    const uint32_t idx = base + lane;
    bool ok = idx < cnt;
    uint32_t cand = 0;
    if (lead) {
        if (ok) cand = C.pool[lo + idx];
        S.surv[uni((uint32_t)S.group[level])][lane] = (uint16_t)cand;
        wave_sync();
    } else {
        cand = ok ? lo + idx : lo;
    }
    const float4 r = C.rec[cand];
    const uint32_t m = __float_as_uint(r.w);
    if (!lead) {
        const unsigned long long nm = uni64(S.namemask[level]);
        const uint32_t rm = uni(S.resmask[level]);
        const bool any = uni((uint32_t)S.any[level]) != 0;
        ok = ok && ((nm >> meta_name(m)) & 1ull) && (any || ((rm >> meta_rtype(m)) & 1u));
    }
    const uint32_t cres = meta_res(m);
    const uint32_t mygroup = uni((uint32_t)S.group[level]);
    uint32_t cchain = 0, mychain = 0;
    if (!C.ignore_chain) {
        cchain = C.qchain[cand];
        mychain = uni((uint32_t)S.tchain[level]);
    }
    TM_STAT(level, 1);
    if (__ballot(ok) == 0ull) return 0ull;
    const float2 *brow = C.bnd + tri(level);
    for (int j = 0; j < level; ++j) {
        const float4 a = S.arec[j];
        const float2 b = brow[j];
        const float dx = r.x - a.x, dy = r.y - a.y, dz = r.z - a.z;
        const float d2 = dx * dx + dy * dy + dz * dz;
        const bool same_t = uni((uint32_t)S.group[j]) == mygroup;
        const bool same_q = meta_res(__float_as_uint(a.w)) == cres;
        ok = ok && (d2 >= b.x) && (d2 <= b.y) && (same_t == same_q) && (cand != S.aidx[j]);
        if (!C.ignore_chain) {
            const bool tc = uni((uint32_t)S.tchain[j]) == mychain;
            const bool qc = (uint32_t)S.achain[j] == cchain;
            ok = ok && (tc == qc);
        }
        TM_STAT(50, 1);
        if (__ballot(ok) == 0ull) return 0ull;
    }
    const unsigned long long out = __ballot(ok);
    TM_STAT(24 + level, __popcll(out));
    return out;
}

// Prepares `level` after levels < level were assigned.
__device__ __forceinline__ void init_level(const SearchCtx &C, int level) {
    WaveScratch &S = *C.S;
    if (C.lane == 0) {
        S.cur[level] = 0;
        S.pend[level] = 0ull;
        if (!S.lead[level]) {
            // same template residue as an already placed lead atom: scan the atoms of that target residue
            const uint32_t res = meta_res(__float_as_uint(S.arec[S.ref[level]].w));
            const uint32_t s0 = C.res_start[res], s1 = C.res_start[res + 1];
            S.lo[level] = s0;
            S.cnt[level] = s1 - s0;
        }
    }
    wave_sync();
}

__device__ __forceinline__ void assign_level(const SearchCtx &C, int level, uint32_t cand) {
    WaveScratch &S = *C.S;
    if (C.lane == 0) {
        S.aidx[level] = cand;
        S.arec[level] = C.rec[cand];
        if (!C.ignore_chain) S.achain[level] = C.qchain[cand];
    }
}

// Everything the wave needs to know about the (structure, template) pair it is searching.
struct PairState {
    uint32_t s, ti, off;      // structure, selected-template index, template atom offset in the library
    uint32_t ncand, prev;     // verified candidates so far, last emitted candidate slot
    bool truncated;
    double cut, mx;
};

// A complete assignment passed the fp32 prefilter: re-check every pair in fp64
// (| |q_i - q_j| - |t_i - t_j| | <= thr_ij exactly as the oracle does) and append it to the
// candidate buffer.  Returns false when the search must stop (max_candidates reached).
__device__ __forceinline__ bool on_complete(const SearchCtx &C, const LibView &L, const double *qx, const QueryView &Q,
                                            const OutView &O, PairState &P) {
    WaveScratch &S = *C.S;
    const int lane = C.lane, k = C.k, npairs = tri(k);
    bool bad = false;
    for (int p = lane; p < npairs; p += 64) {
        int i, j;
        pair_of(p, i, j);
        const double *qa = qx + 3 * (size_t)S.aidx[i], *qb = qx + 3 * (size_t)S.aidx[j];
        const double *ta = L.a_xyz + 3 * (size_t)(P.off + i), *tb = L.a_xyz + 3 * (size_t)(P.off + j);
        double dx = qa[0] - qb[0], dy = qa[1] - qb[1], dz = qa[2] - qb[2];
        const double dq = sqrt(dx * dx + dy * dy + dz * dz);
        dx = ta[0] - tb[0]; dy = ta[1] - tb[1]; dz = ta[2] - tb[2];
        const double dt = sqrt(dx * dx + dy * dy + dz * dz);
        double thr = P.cut + L.a_w[P.off + i] + L.a_w[P.off + j];
        thr = thr < P.mx ? thr : P.mx;
        if (fabs(dq - dt) > thr) bad = true;
    }
    TM_STAT(48, 1);
    if (__ballot(bad)) {  // fp32 prefilter false positive
        TM_STAT(51, 1);
        return true;
    }
    if (P.ncand >= Q.max_candidates) {
        P.truncated = true;
        return false;
    }
    ++P.ncand;
    unsigned long long slot = 0;
    if (lane == 0) slot = atomicAdd(O.n_cands, 1ull);
    slot = uni64(slot);
    if (slot >= O.cand_capacity) {
        if (lane == 0) atomicOr(O.overflow, 1u);
        return true;
    }
    Cand *c = O.cands + slot;
    if (lane == 0) {
        c->s = P.s;
        c->ti = P.ti;
        c->prev = P.prev;
        c->pad = 0;
    }
    if (lane < kMaxK) {
        if (lane < k) c->atoms[S.orig[lane]] = (uint16_t)S.aidx[lane];
        else c->atoms[lane] = 0xffffu;
    }
    P.prev = (uint32_t)slot;
    return true;
}

// Depth-first search over levels >= floor (levels < floor are already assigned).
// Returns false when the pair was cut short by max_candidates.
__device__ __forceinline__ bool dfs(const SearchCtx &C, const LibView &L, const double *qx, const QueryView &Q,
                                    const OutView &O, PairState &P, int floor) {
    WaveScratch &S = *C.S;
    const int lane = C.lane, k = C.k;
    if (floor >= k) return on_complete(C, L, qx, Q, O, P);
    int level = floor;
    init_level(C, level);
    for (;;) {
        unsigned long long pend = uni64(S.pend[level]);
        if (pend == 0ull) {
            // evaluate further chunks of this level until one has survivors
            uint32_t cur = uni(S.cur[level]);
            const uint32_t cnt = uni(S.cnt[level]);
            uint32_t base = 0;
            while (cur < cnt) {
                pend = eval_chunk(C, level, cur);
                base = cur;
                cur += 64;
                if (pend) break;
            }
            if (pend == 0ull) {  // level exhausted: backtrack
                if (level == floor) return true;
                --level;
                continue;
            }
            if (lane == 0) {
                S.cur[level] = cur;
                S.base[level] = base;
            }
        }
        const int b = __ffsll((long long)pend) - 1;
        pend &= pend - 1ull;
        uint32_t cand;
        if (uni((uint32_t)S.lead[level])) cand = S.surv[uni((uint32_t)S.group[level])][b];
        else cand = uni(S.lo[level]) + uni(S.base[level]) + (uint32_t)b;
        cand = uni(cand);
        if (lane == 0) S.pend[level] = pend;
        assign_level(C, level, cand);
        wave_sync();
        if (level < k - 1) {
            ++level;
            init_level(C, level);
            continue;
        }
        if (!on_complete(C, L, qx, Q, O, P)) return false;
    }
}

// Levels 0 and 1 are the lead atoms of two different template residues.  Instead of scanning
// list1 once per element of list0 (a handful of busy lanes each time), the wave tiles the
// cross product list0 x list1 over its 64 lanes and tests the single pair constraint (1, 0).
__device__ __forceinline__ bool joint_top_scan(const SearchCtx &C, const LibView &L, const double *qx,
                                               const QueryView &Q, const OutView &O, PairState &P) {
    WaveScratch &S = *C.S;
    const int lane = C.lane;
    const uint32_t n0 = uni(S.cnt[0]), n1 = uni(S.cnt[1]);
    const uint32_t lo0 = uni(S.lo[0]), lo1 = uni(S.lo[1]);
    const bool in_lds = n0 <= (uint32_t)kJointList && n1 <= (uint32_t)kJointList;
    if (in_lds) {
        for (uint32_t i = lane; i < n0; i += 64) S.list0[i] = C.pool[lo0 + i];
        for (uint32_t i = lane; i < n1; i += 64) S.list1[i] = C.pool[lo1 + i];
        wave_sync();
    }
    const uint32_t tj_shift = n1 <= 8 ? 3 : n1 <= 16 ? 4 : n1 <= 32 ? 5 : 6;
    const uint32_t TJ = 1u << tj_shift, TI = 64u >> tj_shift;
    const uint32_t li = (uint32_t)lane >> tj_shift, lj = (uint32_t)lane & (TJ - 1);
    const float2 b10 = C.bnd[0];  // pair (1, 0)
#ifdef TM_STATS
    bool any_surv = false;
#endif
    const bool tc = uni((uint32_t)S.tchain[0]) == uni((uint32_t)S.tchain[1]);
    for (uint32_t i0 = 0; i0 < n0; i0 += TI) {
        const uint32_t i = i0 + li;
        const bool vi = i < n0;
        uint32_t a = 0;
        if (vi) a = in_lds ? (uint32_t)S.list0[i] : (uint32_t)C.pool[lo0 + i];
        const float4 ra = C.rec[a];
        for (uint32_t j0 = 0; j0 < n1; j0 += TJ) {
            const uint32_t j = j0 + lj;
            const bool vj = j < n1;
            uint32_t b = 0;
            if (vj) b = in_lds ? (uint32_t)S.list1[j] : (uint32_t)C.pool[lo1 + j];
            const float4 rb = C.rec[b];
            const float dx = ra.x - rb.x, dy = ra.y - rb.y, dz = ra.z - rb.z;
            const float d2 = dx * dx + dy * dy + dz * dz;
            bool ok = vi && vj && d2 >= b10.x && d2 <= b10.y &&
                      meta_res(__float_as_uint(ra.w)) != meta_res(__float_as_uint(rb.w));
            if (!C.ignore_chain) ok = ok && ((C.qchain[a] == C.qchain[b]) == tc);
            unsigned long long m = __ballot(ok);
            TM_STAT(52, 1);
            TM_STAT(53, __popcll(m));
#ifdef TM_STATS
            if (m) any_surv = true;
#endif
            while (m) {
                const int bit = __ffsll((long long)m) - 1;
                m &= m - 1ull;
                const uint32_t sa = __builtin_amdgcn_readlane(a, bit), sb = __builtin_amdgcn_readlane(b, bit);
                assign_level(C, 0, sa);
                assign_level(C, 1, sb);
                wave_sync();
#if defined(TM_STATS) || defined(TM_ABLATE)
                if (Q.flags & 0x100u) continue;
#endif
                if (!dfs(C, L, qx, Q, O, P, 2)) return false;
            }
        }
    }
#ifdef TM_STATS
    if (any_surv) TM_STAT(54, 1);
#endif
    return true;
}

template <bool kLds>
__global__ __launch_bounds__(kSearchBlock) void k_search(LibView L, BatchView B, QueryView Q, OutView O,
                                                          uint32_t n_chunks, uint32_t chunk_size,
                                                          uint32_t max_atoms, uint32_t max_res) {
    extern __shared__ __align__(16) unsigned char smem[];
    // [WaveScratch x kSearchWaves][rec float4 x max_atoms][res_start u16 x (max_res+1)]
    WaveScratch *scratch = reinterpret_cast<WaveScratch *>(smem);
    float4 *s_rec = reinterpret_cast<float4 *>(smem + sizeof(WaveScratch) * kSearchWaves);
    uint16_t *s_res = reinterpret_cast<uint16_t *>(s_rec + (kLds ? max_atoms : 0));
    __shared__ uint32_t s_next;

    // all chunks of a structure land on one XCD (blocks b and b+8 share an XCD): its kind
    // lists and records are then served by a single L2.
    const uint32_t bid = blockIdx.x;
    const uint32_t xcd = bid & 7u, q = bid >> 3;
    const uint32_t chunk = q % n_chunks;
    const uint32_t s = (q / n_chunks) * 8u + xcd;
    if (s >= B.n_structs) return;

    const uint32_t a0 = B.s_off[s], n = B.s_off[s + 1] - a0;
    const uint32_t r0 = B.s_roff[s], nres1 = B.s_roff[s + 1] - r0;  // n_res + 1 entries
    if (kLds) {
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) s_rec[i] = B.q_rec[a0 + i];
        for (uint32_t i = threadIdx.x; i < nres1; i += blockDim.x) s_res[i] = B.res_start[r0 + i];
    }
    if (threadIdx.x == 0) s_next = chunk * chunk_size;
    __syncthreads();
    const uint32_t t_end = min((chunk + 1) * chunk_size, Q.n_sel);

    SearchCtx C;
    C.rec = kLds ? s_rec : B.q_rec + a0;
    C.res_start = kLds ? s_res : B.res_start + r0;
    C.pool = B.kl_pool;
    C.qchain = B.q_chain + a0;
    C.S = scratch + (threadIdx.x >> 6);
    C.lane = lane_id();
    C.ignore_chain = (Q.flags & TM_FLAG_IGNORE_CHAIN) != 0;
    WaveScratch &S = *C.S;
    const int lane = C.lane;
    const uint32_t *kl_off = B.kl_off + (size_t)s * (L.n_lead + 1);
    const double *qx = B.q_xyz + 3 * (size_t)a0;

    for (;;) {
        uint32_t ti = 0;
        if (lane == 0) ti = atomicAdd(&s_next, 1u);
        ti = uni(ti);
        if (ti >= t_end) break;
        const Staged &T = Q.staged[ti];
        const int k = (int)T.k;
        C.k = k;
        C.bnd = Q.bounds + (size_t)ti * kMaxPairs;
        TM_STAT(49, 1);

        // ---- stage the template: per-level admissibility and candidate source
        bool empty = false;
        if (lane < k) {
            const uint16_t li = T.lead_li[lane];
            const bool lead = li != 0xffffu;
            S.group[lane] = T.group[lane];
            S.ref[lane] = T.ref[lane];
            S.tchain[lane] = T.tchain[lane];
            S.orig[lane] = T.orig[lane];
            S.resmask[lane] = T.resmask[lane];
            S.namemask[lane] = T.namemask[lane];
            S.any[lane] = T.any[lane];
            S.lead[lane] = lead ? 1 : 0;
            if (lead) {
                const uint32_t o0 = kl_off[li], o1 = kl_off[li + 1];
                S.lo[lane] = o0;
                S.cnt[lane] = o1 - o0;
                empty = o1 == o0;
            }
        }
        const bool feasible = __ballot(empty) == 0ull;  // an empty lead list: no assignment exists
        bool feasible_override = true;
        wave_sync();

        PairState P;
        P.s = s; P.ti = ti; P.off = L.t_off[Q.sel[ti]];
        P.ncand = 0; P.prev = 0xffffffffu; P.truncated = false;
        P.cut = Q.cutoff[ti]; P.mx = Q.maxdyn[ti];

#if defined(TM_STATS) || defined(TM_ABLATE)
        if (Q.flags & 0x200u) feasible_override = false;
#endif
        if (feasible && feasible_override) {
            if (T.n_groups >= 2) joint_top_scan(C, L, qx, Q, O, P);
            else dfs(C, L, qx, Q, O, P, 0);
        }

        // ---- the pair's outcome
        const uint32_t packed = P.ncand | (P.truncated ? 0x80000000u : 0u);
        if (O.candidates && lane == 0) O.candidates[(size_t)s * Q.n_sel + ti] = packed;
        if (packed != 0u && lane == 0) {
            const unsigned long long slot = atomicAdd(O.n_pairs, 1ull);
            if (slot >= O.pair_capacity) atomicOr(O.overflow, 2u);
            else O.pairs[slot] = PairEntry{s, ti, P.prev, packed};
        }
        wave_sync();
    }
}

template __global__ void k_search<true>(LibView, BatchView, QueryView, OutView, uint32_t, uint32_t, uint32_t, uint32_t);
template __global__ void k_search<false>(LibView, BatchView, QueryView, OutView, uint32_t, uint32_t, uint32_t, uint32_t);

size_t search_scratch_bytes() { return sizeof(WaveScratch) * kSearchWaves; }

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
__device__ double superpose_candidate(const LibView &L, const BatchView &B, const QueryView &Q, const Cand &c,
                                      double R[9], double T[3]) {
    const uint32_t off = L.t_off[Q.sel[c.ti]];
    const Staged &St = Q.staged[c.ti];
    const int k = (int)St.k;
    const double *qx = B.q_xyz + 3 * (size_t)B.s_off[c.s];
    // candidate atoms are stored in TEMPLATE order, library coordinates in SEARCH order
    double qc[3] = {0, 0, 0}, pc[3] = {0, 0, 0};
    for (int pos = 0; pos < k; ++pos) {
        const double *q = qx + 3 * (size_t)c.atoms[St.orig[pos]];
        const double *p = L.a_xyz + 3 * (size_t)(off + pos);
        for (int a = 0; a < 3; ++a) { qc[a] += q[a]; pc[a] += p[a]; }
    }
    const double inv_k = 1.0 / (double)k;
    for (int a = 0; a < 3; ++a) { qc[a] *= inv_k; pc[a] *= inv_k; }
    double Sm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    for (int pos = 0; pos < k; ++pos) {
        const double *q = qx + 3 * (size_t)c.atoms[St.orig[pos]];
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
        const double *q = qx + 3 * (size_t)c.atoms[St.orig[pos]];
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
        O.cand_rmsd[c] = superpose_candidate(L, B, Q, O.cands[c], R, T);
    }
}

// true when tuple a (TEMPLATE order) is lexicographically smaller than b
__device__ __forceinline__ bool tuple_less(const Cand &a, const Cand &b, int k) {
    for (int i = 0; i < k; ++i)
        if (a.atoms[i] != b.atoms[i]) return a.atoms[i] < b.atoms[i];
    return false;
}

// One lane per (structure, template) pair with candidates: walk the pair's candidate chain,
// keep the minimum-RMSD one (ties within kTieEps: smaller atom tuple), emit the hit.
__global__ __launch_bounds__(256) void k_select(LibView L, BatchView B, QueryView Q, OutView O) {
    unsigned long long n = *O.n_pairs;
    if (n > O.pair_capacity) n = O.pair_capacity;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long e = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const PairEntry pe = O.pairs[e];
        const double thr = Q.rmsd_thr[pe.ti];
        const int k = (int)Q.staged[pe.ti].k;
        uint32_t best = 0xffffffffu;
        double best_rmsd = 0.0;
        for (uint32_t c = pe.last; c != 0xffffffffu; c = O.cands[c].prev) {
            if (c >= O.cand_capacity) break;  // the chain ran into the overflowed part of the buffer
            const double r = O.cand_rmsd[c];
            if (!(r <= thr)) continue;
            bool better = best == 0xffffffffu || r < best_rmsd - kTieEps;
            if (!better && r <= best_rmsd + kTieEps) better = tuple_less(O.cands[c], O.cands[best], k);
            if (better) { best = c; best_rmsd = r; }
        }
        if (best == 0xffffffffu) continue;
        const unsigned long long slot = atomicAdd(O.n_hits, 1ull);
        if (slot >= O.hit_capacity) { atomicOr(O.overflow, 4u); continue; }
        tm_hit *h = O.hits + slot;
        const Cand &c = O.cands[best];
        double R[9], T[3];
        h->rmsd = superpose_candidate(L, B, Q, c, R, T);
        for (int a = 0; a < 9; ++a) h->rot[a] = R[a];
        for (int a = 0; a < 3; ++a) h->trans[a] = T[a];
        h->structure = pe.s;
        h->template_sel = pe.ti;
        h->n_candidates = pe.count & 0x7fffffffu;
        h->flags = (pe.count >> 31) ? TM_HIT_TRUNCATED : 0u;
        h->n_atoms = (uint32_t)k;
        h->reserved = 0;
        const uint32_t a0 = B.s_off[pe.s];
        for (int i = 0; i < kMaxK; ++i) h->atoms[i] = i < k ? B.q_orig[a0 + c.atoms[i]] : 0xffffffffu;
    }
}

}  // namespace tmx
