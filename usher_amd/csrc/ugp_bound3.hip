// ugp_bound3.hip -- per-batch tables of the third pruning bound (ugp_flatten.hpp "B3", ugp_bound3.hpp), gfx950.
//
// Input: the tiles' "useful" nibbles (which (site, allele) pairs match a variant of some sample of the tile -- written by the tile
// builders) and the tree's static event lists in block order (FlatMat::b3_events: per group of B3_GROUP_BLOCKS blocks, the events
// inside one block, the range starts and the range ends of the others, and the events open at the group's first block).
// Output, per tile and block of B3_BLOCK_WORDS packed-stream words:
//     cum_over[b]  = #events whose block range [b0, b1] contains b                 (upper bound of the useful events on the root
//                                                                                   path of any node with a word in b)
//     cum_under[b] = #events with b0 < b < b1                                       (lower bound of the same)
// plus three 64-ary levels of maxima of cum_over, so that the walk reads the maximum over any block range with one 64-lane load.
// With S(b) / E(b) = inclusive prefix counts of the range starts / ends of the events that span more than one block and same(b) =
// events inside block b:   cum_over[b] = S(b) - E(b - 1) + same(b),   cum_under[b] = S(b - 1) - E(b).
//
// Two steps, neither with a global atomic, and nothing touched but the tables themselves (4 bytes per tile and block):
//   k_b3_pairmask     nibbles of 32 tiles -> one 32-bit tile mask per (site, allele) pair
//   k_b3_group_tables per group of 256 blocks: counts the events per (tile, block) in LDS -- and the events open at the group's first
//                     block, a static list of their own: no scan across groups --, scans them, writes cum_over / cum_under / level 1
//   k_b3_level2 / 3   the two upper levels of maxima
// (first version of the round: one global atomic per event and tile into an 8-byte-per-block work array, zeroed per batch: 1.15 ms
// for the atomics, 0.5 ms for the scan over 370 MB, per batch of 32 tiles -- five times what the bound saves the walk.  Second: group
// sums + a scan along each tile's groups in front of the tables kernel, 70 us of the 257.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugp_bound3.hpp"

namespace ugp {

// pairmask[y][4 * site + allele]: bit t = the pair is useful for tile 32 * y + t
__global__ void __launch_bounds__(256) k_b3_pairmask(const uint32_t *__restrict__ useful, uint32_t useful_words, uint32_t n_sites, uint32_t n_tiles,
                                                     uint4 *__restrict__ pairmask) {
    const uint32_t site = blockIdx.x * 256u + threadIdx.x, y = blockIdx.y;
    if (site >= n_sites) return;
    uint32_t m0 = 0, m1 = 0, m2 = 0, m3 = 0;
    const uint32_t nt = min(32u, n_tiles - y * 32u);
    for (uint32_t t = 0; t < nt; t++) {
        const uint32_t nib = (useful[(uint64_t)(y * 32u + t) * useful_words + (site >> 3)] >> ((site & 7u) * 4u)) & 15u;
        m0 |= (nib & 1u) << t; m1 |= ((nib >> 1) & 1u) << t; m2 |= ((nib >> 2) & 1u) << t; m3 |= (nib >> 3) << t;
    }
    pairmask[(uint64_t)y * n_sites + site] = make_uint4(m0, m1, m2, m3);
}

// the tables of one group of B3_GROUP_BLOCKS blocks for 32 tiles: per (tile, block) one LDS word counts range starts (bits 7:0), events
// inside the block (15:8; both at most B3_BLOCK_WORDS) and range ends (31:16); thread (tile, eighth of the group) then turns 32
// of them into cum_over | cum_under << 16 in place, and the block writes the rows out four bytes (two blocks) at a time.
constexpr uint32_t B3_ROW = B3_GROUP_BLOCKS + 1;   // (padded: a wave's 32 rows fall into 32 different LDS banks)
__global__ void __launch_bounds__(256) k_b3_group_tables(const uint32_t *__restrict__ pairmask, uint32_t n_pairs, uint32_t n_tiles, const uint32_t *__restrict__ group_off,
                                                         const uint32_t *__restrict__ events, uint32_t n_groups, uint32_t n_blocks,
                                                         uint16_t *__restrict__ over, uint16_t *__restrict__ under, uint16_t *__restrict__ l1, uint32_t n_l1) {
    __shared__ uint32_t cnt[32 * B3_ROW];
    __shared__ uint32_t pa[256], pb[256], mx[256], open0[32];
    const uint32_t g = blockIdx.x, y = blockIdx.y, tid = threadIdx.x;
    const uint32_t *pm = pairmask + (uint64_t)y * n_pairs;
    for (uint32_t i = tid; i < 32u * B3_ROW; i += 256u) cnt[i] = 0;
    if (tid < 32u) open0[tid] = 0;
    __syncthreads();
    // list 3 -> open0[tile] = the tile's useful events open at the group's first block (S - E in front of the group: all a group needs
    // of the groups before it -- no scan across groups); lists 0..2 -> the counters
    for (uint32_t k = 0; k < 4u; k++) {
        const uint32_t i0 = group_off[(uint64_t)k * (n_groups + 1) + g], i1 = group_off[(uint64_t)k * (n_groups + 1) + g + 1];
        const uint32_t inc = k == 0 ? 1u << 8 : k == 1 ? 1u : 1u << 16;
        for (uint32_t i = i0 + tid; i < i1; i += 1024u) {   // (four events in flight per thread)
            uint32_t w[4], mk[4];
#pragma unroll
            for (int u = 0; u < 4; u++) w[u] = i + 256u * u < i1 ? events[i + 256u * u] : 0xFFFFFFFFu;
#pragma unroll
            for (int u = 0; u < 4; u++) mk[u] = w[u] != 0xFFFFFFFFu ? pm[w[u] & 0xFFFFFFu] : 0u;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                uint32_t mask = mk[u];
                while (mask) {
                    const uint32_t t = (uint32_t)__builtin_ctz(mask);
                    mask &= mask - 1u;
                    if (k == 3u) atomicAdd(&open0[t], 1u); else atomicAdd(&cnt[t * B3_ROW + (w[u] >> 24)], inc);
                }
            }
        }
    }
    __syncthreads();
    const uint32_t t = tid & 31u, c = tid >> 5, tile = y * 32u + t;
    uint32_t *row = cnt + t * B3_ROW + c * 32u;
    uint32_t a = 0, b = 0;
#pragma unroll 8
    for (int k = 0; k < 32; k++) { const uint32_t w = row[k]; a += w & 0xFFu; b += w >> 16; }
    pa[t * 8u + c] = a; pb[t * 8u + c] = b;
    __syncthreads();
    uint32_t S = open0[t], E = 0;   // (only S - E matters)
    for (uint32_t k = 0; k < c; k++) { S += pa[t * 8u + k]; E += pb[t * 8u + k]; }   // inclusive prefixes in front of the thread's first block
    uint32_t m = 0;
#pragma unroll 8
    for (int k = 0; k < 32; k++) {
        const uint32_t w = row[k];
        const uint32_t st = w & 0xFFu, same = (w >> 8) & 0xFFu, en = w >> 16;
        const uint32_t ov = min(S + st - E + same, 65535u);   // S(b) - E(b - 1) + same(b)   (65535: "no bound" for the walk -- cannot be reached on the packed path)
        const uint32_t un = min(S - (E + en), 65535u);        // S(b - 1) - E(b)   (never negative: an event ends where or behind it starts)
        S += st; E += en;
        row[k] = ov | (un << 16);
        m = max(m, ov);
    }
    mx[t * 8u + c] = m;
    __syncthreads();
    const uint64_t g0 = (uint64_t)g * B3_GROUP_BLOCKS;
    for (uint32_t it = 0; it < 16u; it++) {   // two tile rows per pass: 128 threads x two blocks
        const uint32_t tt = it * 2u + (tid >> 7), i = tid & 127u, tl = y * 32u + tt;
        if (tl >= n_tiles) continue;
        const uint32_t r0 = cnt[tt * B3_ROW + 2u * i], r1 = cnt[tt * B3_ROW + 2u * i + 1u];
        const uint64_t at = (uint64_t)tl * n_blocks + g0 + 2u * i;
        *(uint32_t *)(over + at) = (r0 & 0xFFFFu) | (r1 << 16);
        *(uint32_t *)(under + at) = (r0 >> 16) | (r1 & 0xFFFF0000u);
    }
    if (tid < 128u) {   // level 1: 64 blocks = two threads' maxima
        const uint32_t tt = tid >> 2, q = tid & 3u, tl = y * 32u + tt, i = g * (B3_GROUP_BLOCKS / 64u) + q;
        if (tl < n_tiles && i < n_l1) l1[(uint64_t)tl * n_l1 + i] = (uint16_t)max(mx[tt * 8u + 2u * q], mx[tt * 8u + 2u * q + 1u]);
    }
}

// level 2 (grid: 64 entries per block, tile) and level 3 (one block per tile): a thread reads its 64 entries as sixteen 8-byte loads
// (n_l1 is a multiple of four: whole groups of 256 blocks)
__device__ inline uint32_t b3_max64(const uint16_t *A, uint32_t i, uint32_t n) {
    uint32_t m = 0;
    if (i * 64u + 64u <= n && (n & 3u) == 0) {
        const uint2 *p = (const uint2 *)(A + i * 64u);
        uint2 v[16];
#pragma unroll
        for (int k = 0; k < 16; k++) v[k] = p[k];
#pragma unroll
        for (int k = 0; k < 16; k++) m = max(max(m, max(v[k].x & 0xFFFFu, v[k].x >> 16)), max(v[k].y & 0xFFFFu, v[k].y >> 16));
    } else {
        for (uint32_t k = 0; k < 64u && i * 64u + k < n; k++) m = max(m, (uint32_t)A[i * 64u + k]);
    }
    return m;
}
__global__ void __launch_bounds__(64) k_b3_level2(const uint16_t *__restrict__ l1, uint32_t n_l1, uint16_t *__restrict__ l2, uint32_t n_l2) {
    const uint32_t tile = blockIdx.y, i = blockIdx.x * 64u + threadIdx.x;
    if (i < n_l2) l2[(uint64_t)tile * n_l2 + i] = (uint16_t)b3_max64(l1 + (uint64_t)tile * n_l1, i, n_l1);
}
__global__ void __launch_bounds__(64) k_b3_level3(const uint16_t *__restrict__ l2, uint32_t n_l2, uint16_t *__restrict__ l3, uint32_t n_l3) {
    const uint32_t tile = blockIdx.x;
    for (uint32_t i = threadIdx.x; i < n_l3; i += 64u) l3[(uint64_t)tile * n_l3 + i] = (uint16_t)b3_max64(l2 + (uint64_t)tile * n_l2, i, n_l2);
}

hipError_t launch_b3_tables(const uint32_t *useful, uint32_t useful_words, uint32_t n_sites, uint32_t n_tiles, const uint32_t *group_off, const uint32_t *events,
                            uint32_t n_blocks, uint32_t *pairmask, uint16_t *over, uint16_t *under, uint16_t *l1, uint16_t *l2, uint16_t *l3, hipStream_t s) {
    if (!n_sites || !n_tiles || !n_blocks) return hipSuccess;
    const uint32_t ng = n_blocks >> B3_GROUP_SHIFT, ty = (n_tiles + 31u) / 32u, n_pairs = n_sites * 4u;
    const uint32_t n_l1 = b3_div64(n_blocks), n_l2 = b3_div64(n_l1), n_l3 = b3_div64(n_l2);
    hipLaunchKernelGGL(k_b3_pairmask, dim3((n_sites + 255u) / 256u, ty), dim3(256), 0, s, useful, useful_words, n_sites, n_tiles, (uint4 *)pairmask);
    hipLaunchKernelGGL(k_b3_group_tables, dim3(ng, ty), dim3(256), 0, s, pairmask, n_pairs, n_tiles, group_off, events, ng, n_blocks, over, under, l1, n_l1);
    hipLaunchKernelGGL(k_b3_level2, dim3((n_l2 + 63u) / 64u, n_tiles), dim3(64), 0, s, l1, n_l1, l2, n_l2);
    hipLaunchKernelGGL(k_b3_level3, dim3(n_tiles), dim3(64), 0, s, l2, n_l2, l3, n_l3);
    return hipGetLastError();
}

}  // namespace ugp
