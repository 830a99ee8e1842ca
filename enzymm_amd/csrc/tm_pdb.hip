// Device-side PDB ingestion: raw PDB text in HBM -> the tm_batch structure-of-arrays layout.
//
// Replaces, for the batch path, what enzymm obtains from `pyjess.Molecule.load`
// (/root/reference/enzymm/jess_run.py:538-540: every ATOM/HETATM record up to the first ENDMDL) and
// `Molecule.conserved(cutoff)` (jess_run.py:541-545: keep atoms whose temperature factor >= cutoff).
// It is byte work bounded by HBM bandwidth: every input byte is read twice (newline census, then
// the fixed-column parse of the atom lines), nothing is staged on the host.
//
//   k_pdb_count_newlines / k_pdb_mark_lines   one 16-byte vector load per lane, ballot-free popcounts
//   k_scan_*                                  exclusive prefix sums (lines, kept atoms)
//   k_pdb_classify                            ATOM/HETATM/ENDMDL per line, first ENDMDL per file
//   k_pdb_parse                               fixed-column parse of kept atom lines (exact decimal -> double)
//   k_pdb_residues / k_pdb_finalize           residue runs, duplicate-key detection, chain ordinals,
//                                             centroid, packed records
#include "tm_device.hpp"

namespace tmx {

constexpr int kScanBlock = 256;
constexpr int kScanItems = 8;  // items per thread in the scan kernels
constexpr int kMaxSortedRuns = 16384;  // residue runs per file checked on the device (128 KB of LDS); above: host path

struct VocabView {
    uint32_t n_names, n_elements, n_res;
    const uint32_t *names;      // raw 4-char fields as little-endian u32
    const uint8_t *name_ids;
    const uint16_t *elements;   // raw 2-char element columns as u16
    const uint8_t *element_ids;
    const uint32_t *res;        // raw 3-char residue names in the low 24 bits
    const uint8_t *res_ids;
    uint32_t n_counted;         // residue names whose residue numbers are counted per structure
    const uint32_t *counted;
};

constexpr uint32_t kMetaCounted = 1u << 27;  // ParsedAtoms::meta flag: residue name is one of the counted ones

// ------------------------------------------------------------------ generic exclusive scan (u32)

__global__ __launch_bounds__(kScanBlock) void k_scan_block_sums(const uint32_t *__restrict__ in, uint64_t n,
                                                                 uint32_t *__restrict__ block_sums) {
    __shared__ uint32_t s_wave[kScanBlock / 64];
    const uint64_t base = ((uint64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanItems;
    uint32_t sum = 0;
    for (int i = 0; i < kScanItems; ++i)
        if (base + i < n) sum += in[base + i];
    for (int d = 32; d >= 1; d >>= 1) sum += __shfl_xor((int)sum, d);
    if ((threadIdx.x & 63) == 0) s_wave[threadIdx.x >> 6] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < kScanBlock / 64; ++w) t += s_wave[w];
        block_sums[blockIdx.x] = t;
    }
}

// single workgroup: exclusive scan of the block sums in place (u64 running total kept in u32: the
// callers bound their totals below 2^32)
__global__ __launch_bounds__(1024) void k_scan_sums(uint32_t *__restrict__ sums, uint32_t n, uint32_t *__restrict__ total) {
    __shared__ uint32_t s_part[1024];
    const uint32_t per = (n + 1023) / 1024;
    const uint32_t lo = threadIdx.x * per, hi = min(lo + per, n);
    uint32_t t = 0;
    for (uint32_t i = lo; i < hi; ++i) t += sums[i];
    s_part[threadIdx.x] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < 1024; ++i) { const uint32_t v = s_part[i]; s_part[i] = run; run += v; }
        *total = run;
    }
    __syncthreads();
    uint32_t run = s_part[threadIdx.x];
    for (uint32_t i = lo; i < hi; ++i) { const uint32_t v = sums[i]; sums[i] = run; run += v; }
}

__global__ __launch_bounds__(kScanBlock) void k_scan_apply(const uint32_t *__restrict__ in, uint64_t n,
                                                            const uint32_t *__restrict__ block_offsets,
                                                            uint32_t *__restrict__ out) {
    __shared__ uint32_t s_wave[kScanBlock / 64];
    const uint64_t base = ((uint64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanItems;
    uint32_t v[kScanItems], sum = 0;
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = base + i < n ? in[base + i] : 0u;
        sum += v[i];
    }
    // inclusive scan of the per-thread sums across the workgroup
    uint32_t inc = sum;
    const int lane = threadIdx.x & 63;
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t up = (uint32_t)__shfl_up((int)inc, d);
        if (lane >= d) inc += up;
    }
    if (lane == 63) s_wave[threadIdx.x >> 6] = inc;
    __syncthreads();
    uint32_t wave_off = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) wave_off += s_wave[w];
    uint32_t run = block_offsets[blockIdx.x] + wave_off + inc - sum;
    for (int i = 0; i < kScanItems; ++i) {
        if (base + i < n) out[base + i] = run;
        run += v[i];
    }
}

// ------------------------------------------------------------------ newlines

// 16 bytes per lane; flags[g] = number of '\n' in the 16-byte granule g
__global__ __launch_bounds__(256) void k_pdb_count_newlines(const uint4 *__restrict__ text, uint64_t n_granules,
                                                             uint32_t *__restrict__ counts) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_granules; g += stride) {
        const uint4 v = text[g];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t c = 0;
        for (int i = 0; i < 4; ++i) {
            const uint32_t x = w[i] ^ 0x0a0a0a0au;  // zero byte where '\n'
            c += __popc(((x - 0x01010101u) & ~x) & 0x80808080u);
        }
        counts[g] = c;
    }
}

// writes the byte position of every '\n' at its rank (scan of the granule counts)
__global__ __launch_bounds__(256) void k_pdb_mark_lines(const uint8_t *__restrict__ text, uint64_t n_granules,
                                                         const uint32_t *__restrict__ rank,
                                                         uint64_t *__restrict__ newline_pos) {
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; g < n_granules; g += stride) {
        const uint4 v = reinterpret_cast<const uint4 *>(text)[g];
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t r = rank[g];
        for (int i = 0; i < 4; ++i)
            for (int b = 0; b < 4; ++b)
                if (((w[i] >> (8 * b)) & 0xffu) == 0x0au) newline_pos[r++] = g * 16 + (uint64_t)(4 * i + b);
    }
}

// ------------------------------------------------------------------ classify

__device__ __forceinline__ uint32_t file_of(const uint64_t *file_off, uint32_t n_files, uint64_t pos) {
    uint32_t lo = 0, hi = n_files;  // largest f with file_off[f] <= pos
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (file_off[mid] <= pos) lo = mid; else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ double parse_fixed(const uint8_t *p, int n, bool &ok) {
    // decimal field without exponent: [spaces][sign]digits[.digits][spaces]; value = integer / 10^frac is
    // the correctly rounded double of the decimal (both operands exact), i.e. what strtod returns
    long long mant = 0;
    int frac = -1, digits = 0;
    bool neg = false, seen = false, tail = false;
    for (int i = 0; i < n; ++i) {
        const uint8_t c = p[i];
        if (c == ' ') { if (seen) tail = true; continue; }
        if (tail) { ok = false; return 0.0; }
        if ((c == '-' || c == '+') && !seen) { neg = c == '-'; seen = true; continue; }
        if (c == '.' && frac < 0) { frac = 0; seen = true; continue; }
        if (c < '0' || c > '9') { ok = false; return 0.0; }
        mant = mant * 10 + (c - '0');
        ++digits;
        if (frac >= 0) ++frac;
        seen = true;
    }
    if (digits == 0 || digits > 15) { ok = false; return 0.0; }
    double scale = 1.0;
    for (int i = 0; i < frac; ++i) scale *= 10.0;
    const double v = (double)mant / scale;
    return neg ? -v : v;
}

__device__ __forceinline__ int parse_int(const uint8_t *p, int n, bool &ok) {
    int v = 0, digits = 0;
    bool neg = false;
    for (int i = 0; i < n; ++i) {
        const uint8_t c = p[i];
        if (c == ' ') continue;
        if (c == '-' && digits == 0) { neg = true; continue; }
        if (c < '0' || c > '9') { ok = false; return 0; }
        v = v * 10 + (c - '0');
        ++digits;
    }
    if (digits == 0) ok = false;
    return neg ? -v : v;
}

// one lane per line: is it an atom record, which file, first ENDMDL per file
__global__ __launch_bounds__(256) void k_pdb_classify(const uint8_t *__restrict__ text,
                                                       const uint64_t *__restrict__ newline_pos, uint32_t n_lines,
                                                       const uint64_t *__restrict__ file_off, uint32_t n_files,
                                                       uint32_t ignore_endmdl, uint32_t *__restrict__ line_file,
                                                       uint8_t *__restrict__ line_kind, uint32_t *__restrict__ first_end) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    const uint64_t start = i == 0 ? 0 : newline_pos[i - 1] + 1, end = newline_pos[i];
    const uint32_t f = file_of(file_off, n_files, start);
    line_file[i] = f;
    uint64_t len = end - start;
    const uint8_t *p = text + start;
    uint8_t kind = 0;
    if (len >= 6) {
        const bool atom = p[0] == 'A' && p[1] == 'T' && p[2] == 'O' && p[3] == 'M' && p[4] == ' ' && p[5] == ' ';
        const bool het = p[0] == 'H' && p[1] == 'E' && p[2] == 'T' && p[3] == 'A' && p[4] == 'T' && p[5] == 'M';
        const bool endm = p[0] == 'E' && p[1] == 'N' && p[2] == 'D' && p[3] == 'M' && p[4] == 'D' && p[5] == 'L';
        if (atom || het) kind = 1;
        if (endm && !ignore_endmdl) atomicMin(&first_end[f], i);
    }
    line_kind[i] = kind;
}

// keep[i] = atom record before its file's first ENDMDL (and at or above the temperature-factor cutoff)
__global__ __launch_bounds__(256) void k_pdb_keep(const uint8_t *__restrict__ text, const uint64_t *__restrict__ newline_pos,
                                                   uint32_t n_lines, const uint32_t *__restrict__ line_file,
                                                   const uint8_t *__restrict__ line_kind,
                                                   const uint32_t *__restrict__ first_end, double min_bfactor,
                                                   uint32_t *__restrict__ keep, uint32_t *__restrict__ bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines) return;
    uint32_t k = 0;
    if (line_kind[i] && i < first_end[line_file[i]]) {
        k = 1;
        const uint64_t start = i == 0 ? 0 : newline_pos[i - 1] + 1, end = newline_pos[i];
        uint64_t len = end - start;
        if (len && text[end - 1] == '\r') --len;
        if (len < 54) {
            atomicOr(bad, 1u);  // malformed record: the Python loader raises ValueError for it
            k = 0;
        } else if (min_bfactor > 0.0) {
            bool ok = true;
            double b = 0.0;
            if (len >= 66) b = parse_fixed(text + start + 60, 6, ok);
            else if (len > 60) b = parse_fixed(text + start + 60, (int)(len - 60), ok);
            if (!ok) b = 0.0;
            if (!(b >= min_bfactor)) k = 0;
        }
    }
    keep[i] = k;
}

// s_off[f] = kept atoms before file f (scan value at the file's first line); s_off[n_files] = total
__global__ void k_pdb_file_offsets(const uint64_t *__restrict__ newline_pos, uint32_t n_lines,
                                   const uint64_t *__restrict__ file_off, uint32_t n_files,
                                   const uint32_t *__restrict__ atom_rank, const uint32_t *__restrict__ total,
                                   uint32_t *__restrict__ s_off) {
    const uint32_t f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f > n_files) return;
    if (f == n_files) { s_off[f] = *total; return; }
    // first line whose start >= file_off[f]: number of newlines at positions < file_off[f]
    uint32_t lo = 0, hi = n_lines;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (newline_pos[mid] < file_off[f]) lo = mid + 1; else hi = mid;
    }
    s_off[f] = lo < n_lines ? atom_rank[lo] : *total;
}

// ------------------------------------------------------------------ parse

struct ParsedAtoms {
    double *xyz;          // [n*3]
    uint32_t *meta;       // name id << 16 | residue type << 22 (residue ordinal filled in later)
    uint64_t *key;        // chain (2 bytes) << 40 | (resSeq + 2^24) << 8 | iCode
    uint32_t *line_off;   // byte offset of the record inside its file
};

__global__ __launch_bounds__(256) void k_pdb_parse(const uint8_t *__restrict__ text, const uint64_t *__restrict__ newline_pos,
                                                    uint32_t n_lines, const uint32_t *__restrict__ keep,
                                                    const uint32_t *__restrict__ atom_rank,
                                                    const uint32_t *__restrict__ line_file,
                                                    const uint64_t *__restrict__ file_off, VocabView V,
                                                    ParsedAtoms out, uint32_t *__restrict__ bad) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_lines || !keep[i]) return;
    const uint64_t start = i == 0 ? 0 : newline_pos[i - 1] + 1;
    const uint8_t *p = text + start;
    const uint32_t a = atom_rank[i];
    bool ok = true;
    out.xyz[3 * (size_t)a + 0] = parse_fixed(p + 30, 8, ok);
    out.xyz[3 * (size_t)a + 1] = parse_fixed(p + 38, 8, ok);
    out.xyz[3 * (size_t)a + 2] = parse_fixed(p + 46, 8, ok);
    const int resseq = parse_int(p + 22, 4, ok);
    if (!ok) atomicOr(bad, 1u);
    const uint32_t name = (uint32_t)p[12] | ((uint32_t)p[13] << 8) | ((uint32_t)p[14] << 16) | ((uint32_t)p[15] << 24);
    uint32_t name_id = TM_NAME_OTHER;
    for (uint32_t n = 0; n < V.n_names; ++n)
        if (V.names[n] == name) { name_id = V.name_ids[n]; break; }
    if (name_id == TM_NAME_OTHER) {
        const uint16_t el = (uint16_t)(name & 0xffffu);
        for (uint32_t n = 0; n < V.n_elements; ++n)
            if (V.elements[n] == el) { name_id = V.element_ids[n]; break; }
    }
    const uint32_t res = (uint32_t)p[17] | ((uint32_t)p[18] << 8) | ((uint32_t)p[19] << 16);
    uint32_t rtype = TM_RES_OTHER;
    for (uint32_t n = 0; n < V.n_res; ++n)
        if (V.res[n] == res) { rtype = V.res_ids[n]; break; }
    uint32_t counted = 0;
    for (uint32_t n = 0; n < V.n_counted; ++n)
        if (V.counted[n] == res) { counted = kMetaCounted; break; }
    out.meta[a] = (name_id << 16) | (rtype << 22) | counted;
    out.key[a] = ((uint64_t)p[20] << 48) | ((uint64_t)p[21] << 40) | ((uint64_t)(uint32_t)(resseq + (1 << 24)) << 8) | p[26];
    out.line_off[a] = (uint32_t)(start - file_off[line_file[i]]);
}

// ------------------------------------------------------------------ residues

// per structure: number of residue runs (maximal stretches of equal key) and whether two runs share a
// key (then the file is not residue-contiguous and the caller must take the host path)
__global__ __launch_bounds__(256) void k_pdb_residues(const uint32_t *__restrict__ s_off, const uint64_t *__restrict__ key,
                                                       uint32_t *__restrict__ n_runs, uint32_t *__restrict__ needs_host) {
    extern __shared__ uint64_t s_keys[];  // [kMaxSortedRuns]
    __shared__ uint32_t s_count;
    const uint32_t s = blockIdx.x;
    const uint32_t a0 = s_off[s], n = s_off[s + 1] - a0;
    if (threadIdx.x == 0) s_count = 0;
    __syncthreads();
    // pass 1: count runs; collect run keys while they fit
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        if (i == 0 || key[a0 + i] != key[a0 + i - 1]) {
            const uint32_t slot = atomicAdd(&s_count, 1u);
            if (slot < (uint32_t)kMaxSortedRuns) s_keys[slot] = key[a0 + i];
        }
    }
    __syncthreads();
    const uint32_t runs = s_count;
    if (threadIdx.x == 0) n_runs[s] = runs;
    if (runs > (uint32_t)kMaxSortedRuns || n > TM_NARROW_STRUCTURE_ATOMS) {
        if (threadIdx.x == 0) needs_host[s] = 1u;  // too many residues to check here (or too many atoms)
        return;
    }
    // bitonic sort of the run keys in LDS, then look for equal neighbours
    uint32_t m = 1;
    while (m < runs) m <<= 1;
    for (uint32_t i = runs + threadIdx.x; i < m; i += blockDim.x) s_keys[i] = ~0ull - i;  // distinct padding
    __syncthreads();
    for (uint32_t k2 = 2; k2 <= m; k2 <<= 1)
        for (uint32_t j = k2 >> 1; j > 0; j >>= 1) {
            for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
                const uint32_t ixj = i ^ j;
                if (ixj > i) {
                    const bool up = (i & k2) == 0;
                    const uint64_t x = s_keys[i], y = s_keys[ixj];
                    if ((x > y) == up) { s_keys[i] = y; s_keys[ixj] = x; }
                }
            }
            __syncthreads();
        }
    bool dup = false;
    for (uint32_t i = threadIdx.x; i + 1 < runs; i += blockDim.x) dup |= s_keys[i] == s_keys[i + 1];
    if (dup) needs_host[s] = 1u;
}

// per structure: residue ordinals (= run index), res_start, chain ordinals, centroid, packed records
__global__ __launch_bounds__(256) void k_pdb_finalize(const uint32_t *__restrict__ s_off, const uint32_t *__restrict__ s_roff,
                                                       ParsedAtoms P, float4 *__restrict__ rec, uint8_t *__restrict__ chain,
                                                       uint32_t *__restrict__ orig, uint16_t *__restrict__ res_start,
                                                       uint32_t *__restrict__ residue_count,
                                                       const uint32_t *__restrict__ needs_host) {
    if (needs_host[blockIdx.x]) return;  // flattened on the host (tm_batch_create_from_pdb), its residue table differs
    __shared__ uint32_t s_wave[4];
    // distinct residue NUMBERS among atoms whose residue name is counted (enzymm's query_residue_count ignores
    // chains and insertion codes): one bit per value of the 4-column resSeq field, -999 .. 9999
    __shared__ uint32_t s_seen[(11000 + 31) / 32];
    __shared__ uint32_t s_distinct;
    for (uint32_t i = threadIdx.x; i < (11000 + 31) / 32; i += blockDim.x) s_seen[i] = 0;
    if (threadIdx.x == 0) s_distinct = 0;
    __shared__ uint32_t s_carry;
    __shared__ double s_sum[3][4];
    __shared__ uint16_t s_chains[256];
    const uint32_t s = blockIdx.x;
    const uint32_t a0 = s_off[s], n = s_off[s + 1] - a0;
    const uint32_t r0 = s_roff[s];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_carry = 0;
    double sum[3] = {0, 0, 0};
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x)
        for (int a = 0; a < 3; ++a) sum[a] += P.xyz[3 * (size_t)(a0 + i) + a];
    for (int a = 0; a < 3; ++a) {
        for (int d = 32; d >= 1; d >>= 1) sum[a] += __shfl_xor(sum[a], d);
        if (lane == 0) s_sum[a][wave] = sum[a];
    }
    __syncthreads();
    double c[3];
    for (int a = 0; a < 3; ++a) c[a] = n ? (s_sum[a][0] + s_sum[a][1] + s_sum[a][2] + s_sum[a][3]) / (double)n : 0.0;
    // chain ordinals by first appearance (single lane: chains are few)
    if (threadIdx.x == 0) {
        uint32_t nc = 0;
        uint16_t last = 0xffffu;
        uint8_t last_ord = 0;
        for (uint32_t i = 0; i < n; ++i) {
            const uint16_t code = (uint16_t)(P.key[a0 + i] >> 40);
            if (code != last) {
                uint32_t j = 0;
                while (j < nc && s_chains[j] != code) ++j;
                if (j == nc && nc < 256) s_chains[nc++] = code;
                last = code;
                last_ord = (uint8_t)(j & 0xffu);
            }
            chain[a0 + i] = last_ord;
        }
    }
    // residue ordinal = number of run starts at or before the atom, minus one (blocked scan, file order)
    for (uint32_t base = 0; base < n; base += blockDim.x) {
        const uint32_t i = base + threadIdx.x;
        const bool valid = i < n;
        const uint32_t flag = valid && (i == 0 || P.key[a0 + i] != P.key[a0 + i - 1]) ? 1u : 0u;
        uint32_t inc = flag;
        for (int d = 1; d < 64; d <<= 1) {
            const uint32_t up = (uint32_t)__shfl_up((int)inc, d);
            if (lane >= d) inc += up;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        uint32_t off = s_carry;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        const uint32_t ord = off + inc - 1;  // 0-based residue ordinal
        if (valid) {
            if (flag) res_start[r0 + ord] = (uint16_t)i;
            float4 r;
            r.x = (float)(P.xyz[3 * (size_t)(a0 + i) + 0] - c[0]);
            r.y = (float)(P.xyz[3 * (size_t)(a0 + i) + 1] - c[1]);
            r.z = (float)(P.xyz[3 * (size_t)(a0 + i) + 2] - c[2]);
            const uint32_t m = P.meta[a0 + i];
            r.w = __uint_as_float((ord & 0xffffu) | (m & ~kMetaCounted));
            rec[a0 + i] = r;
            orig[a0 + i] = i;
            if (m & kMetaCounted) {
                const uint32_t v = (uint32_t)((P.key[a0 + i] >> 8) & 0xffffffffull) - (1u << 24) + 999u;  // resSeq + 999
                if (v < 11000u) atomicOr(&s_seen[v >> 5], 1u << (v & 31));
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) res_start[r0 + s_carry] = (uint16_t)n;
    __syncthreads();
    uint32_t bits = 0;
    for (uint32_t i = threadIdx.x; i < (11000 + 31) / 32; i += blockDim.x) bits += __popc(s_seen[i]);
    if (bits) atomicAdd(&s_distinct, bits);
    __syncthreads();
    if (threadIdx.x == 0) residue_count[s] = s_distinct;
}

}  // namespace tmx
