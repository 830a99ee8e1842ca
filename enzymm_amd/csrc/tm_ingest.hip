// Device side of the compact hand-over of host atom tables (round 5): the host sends 17 bytes per atom -- one packed
// word (residue ordinal | atom-name id | residue-type id), a chain ordinal and the coordinates as int32 thousandths of
// an Angstrom (exact for the 3-decimal fields of a PDB file; a structure with any other coordinate sends its fp64
// values instead) -- and k_flatten lays the batch out in HBM: the fp64 coordinates the exact re-check reads, the
// centred fp32 records, the file-order index and the residue table.  Before: the host flattened (2.1 ms per 1000
// structures on 16 threads) and uploaded 46 bytes per atom.
// Replaces the atom tables pyjess.Molecule.load hands to Jess.query (/root/reference/enzymm/jess_run.py:538, 800-811).
#pragma once
#include "tm_device.hpp"

namespace tmx {

constexpr uint32_t kFlattenChunk = 4096;  // atoms per workgroup of k_flatten
constexpr uint32_t kNoOffset = 0xffffffffu;

// What the host knows of a structure when it hands it over (32 bytes).
struct __align__(8) FlatDesc {
    double c[3];        // centroid the records are centred on: ((double)sum of the int32 values / 1000.0) / n for an
                        // exact structure (the sum is an integer: no rounding, no order), the sequential fp64 mean otherwise
    uint32_t xyz_off;   // kNoOffset: coordinates = milli / 1000.0; else first atom of the structure in the fp64 side array
    uint32_t orig_off;  // kNoOffset: the atoms arrive in file order (orig = identity); else first atom in the side array
                        // of file-order indices (a structure whose residues were not contiguous: sorted on the host)
};
static_assert(sizeof(FlatDesc) == 32, "FlatDesc layout");

// grid (structures, chunks of kFlattenChunk atoms); purely per atom -- the centroid comes with the descriptor.
// The atoms arrive residue-contiguous with ordinals ascending from 0 (by first appearance), so a residue starts where
// the ordinal changes.
__global__ __launch_bounds__(256) void k_flatten(const uint32_t *__restrict__ s_off, const uint32_t *__restrict__ s_roff,
                                                  const FlatDesc *__restrict__ desc, const uint32_t *__restrict__ meta,
                                                  const int32_t *__restrict__ milli, const double *__restrict__ xyz_side,
                                                  const uint32_t *__restrict__ orig_side, double *__restrict__ q_xyz,
                                                  float4 *__restrict__ q_rec, uint32_t *__restrict__ q_orig,
                                                  uint16_t *__restrict__ res_start, uint8_t *__restrict__ res_start_hi) {
    const uint32_t s = blockIdx.x;
    const uint32_t a0 = s_off[s], n = s_off[s + 1] - a0;
    const uint32_t begin = blockIdx.y * kFlattenChunk;
    if (begin >= n && !(n == 0 && blockIdx.y == 0)) return;
    const uint32_t r0 = s_roff[s];
    if (n == 0) {
        if (threadIdx.x == 0) {
            res_start[r0] = 0;
            if (res_start_hi) res_start_hi[r0] = 0;
        }
        return;
    }
    const FlatDesc d = desc[s];
    const uint32_t end = min(n, begin + kFlattenChunk);
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const size_t a = (size_t)a0 + i;
        double x, y, z;
        if (d.xyz_off == kNoOffset) {
            // the correctly rounded quotient of two exactly represented integers: the double strtod gives for the field
            x = (double)milli[3 * a] / 1000.0;
            y = (double)milli[3 * a + 1] / 1000.0;
            z = (double)milli[3 * a + 2] / 1000.0;
        } else {
            const double *p = xyz_side + 3 * ((size_t)d.xyz_off + i);
            x = p[0]; y = p[1]; z = p[2];
        }
        q_xyz[3 * a] = x; q_xyz[3 * a + 1] = y; q_xyz[3 * a + 2] = z;
        const uint32_t m = meta[a];
        q_rec[a] = make_float4((float)(x - d.c[0]), (float)(y - d.c[1]), (float)(z - d.c[2]), __uint_as_float(m));
        q_orig[a] = d.orig_off == kNoOffset ? i : orig_side[(size_t)d.orig_off + i];
        const uint32_t ord = meta_res(m);
        // (res_start_hi: bits 16-23, for batches that hold a structure of more than 65535 atoms)
        if (i == 0 || meta_res(meta[a - 1]) != ord) {
            res_start[r0 + ord] = (uint16_t)i;
            if (res_start_hi) res_start_hi[r0 + ord] = (uint8_t)(i >> 16);
        }
        if (i + 1 == n) {
            res_start[r0 + ord + 1] = (uint16_t)n;
            if (res_start_hi) res_start_hi[r0 + ord + 1] = (uint8_t)(n >> 16);
        }
    }
}

// out[i] = table[index[i]] (tm_batch_line_offsets_at: the byte offsets of the matched atoms' records only)
__global__ __launch_bounds__(256) void k_gather_u32(const uint32_t *__restrict__ table, const uint32_t *__restrict__ index,
                                                     uint32_t n, uint32_t *__restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = table[index[i]];
}

}  // namespace tmx
