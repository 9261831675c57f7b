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
//   k_b3_levels       the two upper levels of maxima
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
// inside the block (15:8; both at most B3_BLOCK_WORDS: an event is listed under the block of its own mutation word, and a block has 16
// words) and range ends (31:16; at most the mutation words of one root path, < 0x7F7F where the lists exist: ugp_flatten.cpp); thread (tile, sixteenth of the group) then turns 16
// of them into cum_over | cum_under << 16 in place (each saturated at 255), and the block writes the rows out four bytes (four blocks) at a time.
constexpr uint32_t B3_ROW = B3_GROUP_BLOCKS + 1;   // (padded: a wave's 32 rows fall into 32 different LDS banks)
#ifndef UGP_B3_TB
#define UGP_B3_TB 512
#endif
constexpr uint32_t B3_TB = UGP_B3_TB, B3_CH = B3_TB / 32, B3_CB = B3_GROUP_BLOCKS / B3_CH;   // threads (128, 256 or 512); scan pieces per tile row; blocks per piece
static_assert(B3_TB == 128 || B3_TB == 256 || B3_TB == 512, "k_b3_group_tables: 128, 256 or 512 threads");
__global__ void __launch_bounds__(B3_TB) k_b3_group_tables(const uint32_t *__restrict__ pairmask, uint32_t n_pairs, uint32_t n_tiles, const uint32_t *__restrict__ group_off,
                                                           const uint32_t *__restrict__ events, uint32_t n_groups, uint32_t n_blocks,
                                                           uint8_t *__restrict__ over, uint8_t *__restrict__ under, uint8_t *__restrict__ l1, uint32_t n_l1) {
    __shared__ uint32_t cnt[32 * B3_ROW];
    __shared__ uint32_t pa[B3_TB], pb[B3_TB], mx[B3_TB], open0[32];
    const uint32_t g = blockIdx.x, y = blockIdx.y, tid = threadIdx.x;
    const uint32_t *pm = pairmask + (uint64_t)y * n_pairs;
    for (uint32_t i = tid; i < 32u * B3_ROW; i += B3_TB) cnt[i] = 0;
    if (tid < 32u) open0[tid] = 0;
    // the group's four lists as one index space (each pass: event word -> tile mask, a dependent pair of loads, four per thread in
    // flight -- the whole group in two passes).  List 3 -> open0[tile] = the tile's useful events open at the group's first block
    // (S - E in front of the group: all a group needs of the groups before it -- no scan across groups); lists 0..2 -> the counters.
    uint32_t lb[4], cum[5];
    cum[0] = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        lb[k] = group_off[(uint64_t)k * (n_groups + 1) + g];
        cum[k + 1] = cum[k] + (group_off[(uint64_t)k * (n_groups + 1) + g + 1] - lb[k]);
    }
    __syncthreads();
    for (uint32_t base = 0; base < cum[4]; base += 4u * B3_TB) {
        uint32_t w[4], mk[4], kk[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const uint32_t idx = base + tid + B3_TB * u;
            const uint32_t k = (idx >= cum[1] ? 1u : 0u) + (idx >= cum[2] ? 1u : 0u) + (idx >= cum[3] ? 1u : 0u);
            kk[u] = k;
            const uint32_t src = k == 0 ? lb[0] + idx : k == 1 ? lb[1] + (idx - cum[1]) : k == 2 ? lb[2] + (idx - cum[2]) : lb[3] + (idx - cum[3]);
            w[u] = idx < cum[4] ? events[src] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < 4; u++) mk[u] = w[u] != 0xFFFFFFFFu ? pm[w[u] & 0xFFFFFFu] : 0u;
#pragma unroll
        for (int u = 0; u < 4; u++) {
            uint32_t mask = mk[u];
            const uint32_t inc = kk[u] == 0 ? 1u << 8 : kk[u] == 1 ? 1u : 1u << 16;
            while (mask) {
                const uint32_t t = (uint32_t)__builtin_ctz(mask);
                mask &= mask - 1u;
                if (kk[u] == 3u) atomicAdd(&open0[t], 1u); else atomicAdd(&cnt[t * B3_ROW + (w[u] >> 24)], inc);
            }
        }
    }
    __syncthreads();
    const uint32_t t = tid & 31u, c = tid >> 5;
    uint32_t *row = cnt + t * B3_ROW + c * B3_CB;
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < (int)B3_CB; k++) { const uint32_t w = row[k]; a += w & 0xFFu; b += w >> 16; }
    pa[t * B3_CH + c] = a; pb[t * B3_CH + c] = b;
    __syncthreads();
    uint32_t S = open0[t], E = 0;   // (only S - E matters)
    for (uint32_t k = 0; k < c; k++) { S += pa[t * B3_CH + k]; E += pb[t * B3_CH + k]; }   // inclusive prefixes in front of the thread's first block
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < (int)B3_CB; k++) {
        const uint32_t w = row[k];
        const uint32_t st = w & 0xFFu, same = (w >> 8) & 0xFFu, en = w >> 16;
        const uint32_t ov = min(S + st - E + same, 255u);   // S(b) - E(b - 1) + same(b)   (255: "no bound" for the walk)
        const uint32_t un = min(S - (E + en), 255u);        // S(b - 1) - E(b)   (never negative: an event ends where or behind it starts; saturated like ov: a node whose
                                                            //  count is 255 or more has only descendants whose blocks read 255 -- no bound, the safe answer)
        S += st; E += en;
        row[k] = ov | (un << 16);
        m = max(m, ov);
    }
    mx[t * B3_CH + c] = m;
    __syncthreads();
    const uint64_t g0 = (uint64_t)g * B3_GROUP_BLOCKS;
    for (uint32_t it = 0; it < 32u / (B3_TB / 64u); it++) {   // B3_TB / 64 tile rows per pass: 64 threads x four blocks (one dword of bytes) each
        const uint32_t tt = it * (B3_TB / 64u) + (tid >> 6), i = tid & 63u, tl = y * 32u + tt;
        if (tl >= n_tiles) continue;
        const uint32_t *r = cnt + tt * B3_ROW + 4u * i;
        const uint32_t r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
        const uint64_t at = (uint64_t)tl * n_blocks + g0 + 4u * i;
        *(uint32_t *)(over + at) = (r0 & 0xFFu) | ((r1 & 0xFFu) << 8) | ((r2 & 0xFFu) << 16) | ((r3 & 0xFFu) << 24);
        *(uint32_t *)(under + at) = ((r0 >> 16) & 0xFFu) | (((r1 >> 16) & 0xFFu) << 8) | (((r2 >> 16) & 0xFFu) << 16) | (((r3 >> 16) & 0xFFu) << 24);
    }
    if (tid < 128u) {   // level 1: 64 blocks = 64 / B3_CB pieces
        constexpr uint32_t PER = 64u / B3_CB;
        const uint32_t tt = tid >> 2, q = tid & 3u, tl = y * 32u + tt, i = g * (B3_GROUP_BLOCKS / 64u) + q;
        uint32_t v = 0;
#pragma unroll
        for (uint32_t k = 0; k < PER; k++) v = max(v, mx[tt * B3_CH + PER * q + k]);
        if (tl < n_tiles && i < n_l1) l1[(uint64_t)tl * n_l1 + i] = (uint8_t)v;
    }
}

// levels 2 and 3 in one launch: block (i, tile) owns level-3 entry i = 64 level-2 entries = 4096 level-1 entries; a wave reduces 64
// level-1 entries at a time (one per lane) to a level-2 entry, the block's maximum is the level-3 entry
__global__ void __launch_bounds__(256) k_b3_levels(const uint8_t *__restrict__ l1, uint32_t n_l1, uint8_t *__restrict__ l2, uint32_t n_l2, uint8_t *__restrict__ l3,
                                                   uint32_t n_l3) {
    __shared__ uint32_t wm[4];
    const uint32_t i3 = blockIdx.x, tile = blockIdx.y, lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    const uint8_t *A = l1 + (uint64_t)tile * n_l1;
    uint32_t m3 = 0;
    for (uint32_t e = wv; e < 64u; e += 4u) {
        const uint32_t i2 = i3 * 64u + e;
        if (i2 >= n_l2) break;
        const uint32_t at = i2 * 64u + lane;
        uint32_t v = at < n_l1 ? (uint32_t)A[at] : 0u;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
        if (lane == 0) l2[(uint64_t)tile * n_l2 + i2] = (uint8_t)v;
        m3 = max(m3, v);
    }
    if (lane == 0) wm[wv] = m3;
    __syncthreads();
    if (threadIdx.x == 0 && i3 < n_l3) l3[(uint64_t)tile * n_l3 + i3] = (uint8_t)max(max(wm[0], wm[1]), max(wm[2], wm[3]));
}

hipError_t launch_b3_tables(const uint32_t *useful, uint32_t useful_words, uint32_t n_sites, uint32_t n_tiles, const uint32_t *group_off, const uint32_t *events,
                            uint32_t n_blocks, uint32_t *pairmask, uint8_t *over, uint8_t *under, uint8_t *l1, uint8_t *l2, uint8_t *l3, hipStream_t s) {
    if (!n_sites || !n_tiles || !n_blocks) return hipSuccess;
    const uint32_t ng = n_blocks >> B3_GROUP_SHIFT, ty = (n_tiles + 31u) / 32u, n_pairs = n_sites * 4u;
    const uint32_t n_l1 = b3_div64(n_blocks), n_l2 = b3_div64(n_l1), n_l3 = b3_div64(n_l2);
    hipLaunchKernelGGL(k_b3_pairmask, dim3((n_sites + 255u) / 256u, ty), dim3(256), 0, s, useful, useful_words, n_sites, n_tiles, (uint4 *)pairmask);
    hipLaunchKernelGGL(k_b3_group_tables, dim3(ng, ty), dim3(B3_TB), 0, s, pairmask, n_pairs, n_tiles, group_off, events, ng, n_blocks, over, under, l1, n_l1);
    hipLaunchKernelGGL(k_b3_levels, dim3(n_l3, n_tiles), dim3(256), 0, s, l1, n_l1, l2, n_l2, l3, n_l3);
    return hipGetLastError();
}

}  // namespace ugp
