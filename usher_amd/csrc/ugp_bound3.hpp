// ugp_bound3.hpp -- the per-batch tables of the third pruning bound (ugp_flatten.hpp "B3"; kernels in ugp_bound3.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugp_flatten.hpp"

namespace ugp {

// What k_best8 reads at a record whose first two tests failed (one small struct on the device: the walk's kernel arguments stay
// what they were -- the kernel sits at its scalar-register limit).  All arrays are per 512-sample tile, tile-major.
struct B3Dev {
    // (round 6: one BYTE per tile and block -- 255 = "255 or more: no bound".  cum values are counts of useful mutations on one root
    // path, a few dozen at most in any tree the bound matters for; 16-bit tables were 191 MB written per batch of 32 tiles at 10 M nodes.)
    const uint8_t *over;     // [n_tiles][n_blocks]   cum_over, saturated
    const uint8_t *under;    // [n_tiles][n_blocks]   cum_under, saturated
    const uint8_t *l1;       // [n_tiles][n_l1]       maxima of cum_over over 64 blocks
    const uint8_t *l2;       // [n_tiles][n_l2]       ... over 64 x 64
    const uint8_t *l3;       // [n_tiles][n_l3]       ... over 64^3
    uint32_t n_blocks, n_l1, n_l2, n_l3;
};

// useful[tile][useful_words]: one nibble per site, bit a = allele a is in some sample's set that excludes the reference base.
// group_off / events: FlatMat::b3_group_off / b3_events on the device; n_blocks = b3_blocks(packed-stream words) (whole groups).
// Scratch: pairmask [ceil(n_tiles / 32)][4 * n_sites] words.
// Fills over / under ([n_tiles][n_blocks]) and l1 / l2 / l3.
hipError_t launch_b3_tables(const uint32_t *useful, uint32_t useful_words, uint32_t n_sites, uint32_t n_tiles, const uint32_t *group_off, const uint32_t *events,
                            uint32_t n_blocks, uint32_t *pairmask, uint8_t *over, uint8_t *under, uint8_t *l1, uint8_t *l2, uint8_t *l3, hipStream_t s);
inline uint32_t b3_div64(uint32_t n) { return (n + 63u) / 64u; }

}  // namespace ugp
