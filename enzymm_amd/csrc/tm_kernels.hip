// gfx950 kernels of the template matcher.
//
//   k_pair_bounds   per selected template: fp32 (lo^2, hi^2) bounds of every pair constraint
//   k_kind_lists    per structure: for every "lead kind" the ascending list of admissible atoms
//   k_search        wave-per-(structure, template) backtracking search + fp64 re-check +
//                   quaternion superposition, best hit per pair appended to the hit table
//
// The search restates what Jess does for `Jess.query(..., best_match=True)`
// (/root/reference/enzymm/jess_run.py:803-811, semantics in SURVEY.md 8c-2): enumerate every
// assignment template atom -> target atom that is type-admissible, residue-consistent and
// satisfies all pair-distance constraints, superpose each, keep the minimum-RMSD one.
// The enumeration ORDER is this kernel's own (most selective template residue first); the
// SET of complete assignments does not depend on it.
#include "tm_device.hpp"

namespace tmx {

#ifdef TM_STATS
// diagnostic build only: [0..23] chunk evaluations per level, [24..47] survivors per level,
// [48] complete assignments, [49] searches, [50] pair tests (j-loop trips), [51] fp64 rejects
__device__ unsigned long long g_stats[64];
#define TM_STAT(i, v) do { if (lane_id() == 0) atomicAdd(&g_stats[i], (unsigned long long)(v)); } while (0)
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
__device__ __forceinline__ float unif(float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); }

// Orders this wave's LDS traffic: lanes exchange data through LDS without a workgroup barrier.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int tri(int i) { return i * (i - 1) / 2; }  // pair(i, j<i) = tri(i) + j

// ------------------------------------------------------------------ k_pair_bounds

__global__ __launch_bounds__(64) void k_pair_bounds(LibView L, uint32_t n_sel, const uint32_t *__restrict__ sel,
                                                     const double *__restrict__ cutoff,
                                                     const double *__restrict__ maxdyn, float2 *__restrict__ bounds) {
    const uint32_t ti = blockIdx.x;
    if (ti >= n_sel) return;
    const uint32_t t = sel[ti];
    const uint32_t off = L.t_off[t];
    const int k = (int)(L.t_off[t + 1] - off);
    const int npairs = k * (k - 1) / 2;
    const double cut = cutoff[ti], mx = maxdyn[ti];
    for (int p = threadIdx.x; p < npairs; p += 64) {
        int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
        while (tri(i) > p) --i;
        while (tri(i + 1) <= p) ++i;
        const int j = p - tri(i);
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

__device__ __forceinline__ double shfl_d(double v, int src) {
    const unsigned long long u = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)u, src);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(u >> 32), src);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ------------------------------------------------------------------ k_search

// Per-wave scratch in LDS.  Everything indexed by "level" = position in the search order.
struct __align__(16) WaveScratch {
    float4 arec[kMaxK];               // assigned target atom record per level
    float2 bnd[kMaxPairs];            // (lo^2, hi^2) per pair, search order
    unsigned long long pend[kMaxK];   // survivors of the current chunk not yet descended into
    unsigned long long namemask[kMaxK];
    uint32_t resmask[kMaxK];
    uint32_t aidx[kMaxK];             // assigned target atom (local index) per level
    uint32_t cur[kMaxK];              // next chunk base to evaluate
    uint32_t base[kMaxK];             // base of the chunk `pend` refers to
    uint32_t lo[kMaxK];               // lead: pool offset of the kind list; else: first atom of the residue
    uint32_t cnt[kMaxK];              // number of candidates at this level
    uint32_t best[kMaxK];             // best assignment so far (search order)
    uint16_t surv[kMaxG][64];         // lead levels: candidate atom of every lane of the current chunk
    uint8_t group[kMaxK], tchain[kMaxK], orig[kMaxK], any[kMaxK], lead[kMaxK], achain[kMaxK], ref[kMaxK];
};

struct SearchCtx {
    const float4 *rec;          // structure atom records (LDS or global)
    const uint16_t *res_start;  // residue -> first atom (LDS or global)
    const uint16_t *pool;       // kind-list pool
    const uint8_t *qchain;      // global, per atom of this structure
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
    const int pbase = tri(level);
    for (int j = 0; j < level; ++j) {
        const float4 a = S.arec[j];
        const float2 b = S.bnd[pbase + j];
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
    TM_STAT(24 + level, __popcll(__ballot(ok)));
    return __ballot(ok);
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
        const uint32_t t = Q.sel[ti];
        const uint32_t off = L.t_off[t];
        const int k = (int)(L.t_off[t + 1] - off);
        const int npairs = tri(k);
        C.k = k;
        TM_STAT(49, 1);

        // ---- stage the template: per-level admissibility, candidate source, pair bounds
        if (lane < k) {
            const uint32_t kind = L.a_kind[off + lane];
            const uint8_t g = L.a_group[off + lane];
            const uint8_t ref = L.a_ref[off + lane];
            const bool lead = ref == lane;
            S.group[lane] = g;
            S.ref[lane] = ref;
            S.tchain[lane] = L.a_chain[off + lane];
            S.orig[lane] = L.a_orig[off + lane];
            S.resmask[lane] = L.k_resmask[kind];
            S.namemask[lane] = L.k_namemask[kind];
            S.any[lane] = L.k_any[kind];
            S.lead[lane] = lead ? 1 : 0;
            if (lead) {
                const int li = L.k_lead[kind];
                const uint32_t o0 = kl_off[li], o1 = kl_off[li + 1];
                S.lo[lane] = o0;
                S.cnt[lane] = o1 - o0;
            }
        }
        for (int p = lane; p < npairs; p += 64) S.bnd[p] = Q.bounds[(size_t)ti * kMaxPairs + p];
        wave_sync();

        const double rmsd_thr = Q.rmsd_thr[ti];
        double best_rmsd = 1e300, best_R[9], best_T[3];
        uint32_t ncand = 0;
        bool truncated = false, found = false;

        // an empty candidate list for any lead level means no assignment exists
        bool feasible = true;
        {
            bool empty = lane < k && S.lead[lane] && S.cnt[lane] == 0;
            if (__ballot(empty)) feasible = false;
        }

        int level = 0;
        if (feasible) init_level(C, 0);
        while (feasible) {
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
                    if (level == 0) break;
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
            if (lane == 0) {
                S.pend[level] = pend;
                S.aidx[level] = cand;
                S.arec[level] = C.rec[cand];
                if (!C.ignore_chain) S.achain[level] = C.qchain[cand];
            }
            wave_sync();
            if (level < k - 1) {
                ++level;
                init_level(C, level);
                continue;
            }

            // ---- complete assignment: fp64 re-check of every pair, then superposition
            bool bad = false;
            {
                const double cut = Q.cutoff[ti], mx = Q.maxdyn[ti];
                for (int p = lane; p < npairs; p += 64) {
                    int i = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
                    while (tri(i) > p) --i;
                    while (tri(i + 1) <= p) ++i;
                    const int j = p - tri(i);
                    const double *qa = qx + 3 * (size_t)S.aidx[i], *qb = qx + 3 * (size_t)S.aidx[j];
                    const double *ta = L.a_xyz + 3 * (size_t)(off + i), *tb = L.a_xyz + 3 * (size_t)(off + j);
                    double dx = qa[0] - qb[0], dy = qa[1] - qb[1], dz = qa[2] - qb[2];
                    const double dq = sqrt(dx * dx + dy * dy + dz * dz);
                    dx = ta[0] - tb[0]; dy = ta[1] - tb[1]; dz = ta[2] - tb[2];
                    const double dt = sqrt(dx * dx + dy * dy + dz * dz);
                    double thr = cut + L.a_w[off + i] + L.a_w[off + j];
                    thr = thr < mx ? thr : mx;
                    if (fabs(dq - dt) > thr) bad = true;
                }
            }
            TM_STAT(48, 1);
            if (__ballot(bad)) { TM_STAT(51, 1); continue; }  // fp32 prefilter false positive
            if (ncand >= Q.max_candidates) {
                truncated = true;
                break;
            }
            ++ncand;

            // lanes < k hold one atom pair each; sums are formed redundantly by every lane
            double mqx = 0, mqy = 0, mqz = 0, mpx = 0, mpy = 0, mpz = 0;
            if (lane < k) {
                const double *qa = qx + 3 * (size_t)S.aidx[lane];
                const double *ta = L.a_xyz + 3 * (size_t)(off + lane);
                mqx = qa[0]; mqy = qa[1]; mqz = qa[2];
                mpx = ta[0]; mpy = ta[1]; mpz = ta[2];
            }
            double qc[3] = {0, 0, 0}, pc[3] = {0, 0, 0};
            for (int i = 0; i < k; ++i) {
                qc[0] += shfl_d(mqx, i); qc[1] += shfl_d(mqy, i); qc[2] += shfl_d(mqz, i);
                pc[0] += shfl_d(mpx, i); pc[1] += shfl_d(mpy, i); pc[2] += shfl_d(mpz, i);
            }
            const double inv_k = 1.0 / (double)k;
            for (int a = 0; a < 3; ++a) { qc[a] *= inv_k; pc[a] *= inv_k; }
            double Sm[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
            for (int i = 0; i < k; ++i) {
                const double qv[3] = {shfl_d(mqx, i) - qc[0], shfl_d(mqy, i) - qc[1], shfl_d(mqz, i) - qc[2]};
                const double pv[3] = {shfl_d(mpx, i) - pc[0], shfl_d(mpy, i) - pc[1], shfl_d(mpz, i) - pc[2]};
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int c = 0; c < 3; ++c) Sm[a][c] += qv[a] * pv[c];
            }
            double R[9], T[3];
            horn_rotation(Sm, R);
            for (int a = 0; a < 3; ++a) T[a] = pc[a] - (R[3 * a] * qc[0] + R[3 * a + 1] * qc[1] + R[3 * a + 2] * qc[2]);
            double sum = 0.0;
            for (int i = 0; i < k; ++i) {
                const double x = shfl_d(mqx, i), y = shfl_d(mqy, i), z = shfl_d(mqz, i);
                const double ex = R[0] * x + R[1] * y + R[2] * z + T[0] - shfl_d(mpx, i);
                const double ey = R[3] * x + R[4] * y + R[5] * z + T[1] - shfl_d(mpy, i);
                const double ez = R[6] * x + R[7] * y + R[8] * z + T[2] - shfl_d(mpz, i);
                sum += ex * ex + ey * ey + ez * ez;
            }
            const double rmsd = sqrt(sum * inv_k);
            if (!(rmsd <= rmsd_thr)) continue;
            bool better = !found || rmsd < best_rmsd - kTieEps;
            if (found && !better && rmsd <= best_rmsd + kTieEps) {
                // tie (symmetric copies of a site): the lexicographically smaller tuple in TEMPLATE
                // order wins, whatever order the two were found in
                // lane l < k compares the atom at template position orig[l]
                uint32_t mine = 0, theirs = 0, pos = 0xffffffffu;
                if (lane < k) { mine = S.aidx[lane]; theirs = S.best[lane]; pos = S.orig[lane]; }
                // smallest template position where they differ
                uint32_t key = (lane < k && mine != theirs) ? pos : 0xffffffffu;
                for (int d = 32; d >= 1; d >>= 1) key = min(key, (uint32_t)__shfl_xor((int)key, d));
                const unsigned long long who = __ballot(lane < k && pos == key);
                if (key != 0xffffffffu && who) {
                    const int wl = __ffsll((long long)who) - 1;
                    better = (uint32_t)__shfl((int)mine, wl) < (uint32_t)__shfl((int)theirs, wl);
                }
            }
            if (better) {
                found = true;
                best_rmsd = rmsd;
                for (int a = 0; a < 9; ++a) best_R[a] = R[a];
                for (int a = 0; a < 3; ++a) best_T[a] = T[a];
                if (lane < k) S.best[lane] = S.aidx[lane];
                wave_sync();
            }
        }

        // ---- emit
        if (O.candidates && lane == 0)
            O.candidates[(size_t)s * Q.n_sel + ti] = ncand | (truncated ? 0x80000000u : 0u);
        if (found) {
            unsigned long long slot = 0;
            if (lane == 0) slot = atomicAdd(O.n_hits, 1ull);
            slot = uni64(slot);
            if (slot >= O.capacity) {
                if (lane == 0) atomicExch(O.overflow, 1u);
            } else {
                tm_hit *h = O.hits + slot;
                if (lane == 0) {
                    h->structure = s;
                    h->template_sel = ti;
                    h->n_candidates = ncand;
                    h->flags = truncated ? TM_HIT_TRUNCATED : 0u;
                    h->rmsd = best_rmsd;
                    for (int a = 0; a < 9; ++a) h->rot[a] = best_R[a];
                    for (int a = 0; a < 3; ++a) h->trans[a] = best_T[a];
                    h->n_atoms = (uint32_t)k;
                    h->reserved = 0;
                }
                if (lane < kMaxK) {
                    if (lane < k) h->atoms[S.orig[lane]] = B.q_orig[a0 + S.best[lane]];
                    else h->atoms[lane] = 0xffffffffu;
                }
            }
        }
        wave_sync();
    }
}

template __global__ void k_search<true>(LibView, BatchView, QueryView, OutView, uint32_t, uint32_t, uint32_t, uint32_t);
template __global__ void k_search<false>(LibView, BatchView, QueryView, OutView, uint32_t, uint32_t, uint32_t, uint32_t);

size_t search_scratch_bytes() { return sizeof(WaveScratch) * kSearchWaves; }

}  // namespace tmx
