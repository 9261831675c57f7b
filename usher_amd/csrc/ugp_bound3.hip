// ugp_bound3.hip -- per-batch tables of the third pruning bound (ugp_flatten.hpp "B3", ugp_bound3.hpp), gfx950.
//
// Input: the tile's "useful" nibbles (which (site, allele) pairs match a variant of some sample of the tile -- written by the tile
// builders) and the tree's static posting lists (FlatMat::b3_events: per pair, the block range of every event's subtree).
// Output, per tile and block of B3_BLOCK_WORDS packed-stream words:
//     cum_over[b]  = #events whose block range [b0, b1] contains b                 (upper bound of the useful events on the root
//                                                                                   path of any node with a word in b)
//     cum_under[b] = #events with b0 < b < b1                                       (lower bound of the same)
// plus three 64-ary levels of maxima of cum_over, so that the walk reads the maximum over any block range with one 64-lane load.
// With S(b) / E(b) = inclusive prefix counts of the range starts / ends of the events that span more than one block and same(b) =
// events inside block b:   cum_over[b] = S(b) - E(b - 1) + same(b),   cum_under[b] = S(b - 1) - E(b).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugp_bound3.hpp"

namespace ugp {

// work[tile][block] = {start count | same count << 16, end count}.  One wave per 64 (site, allele) pairs: the lanes look their
// pairs up in the tile's useful nibbles, then all 64 walk the posting list of each useful pair together.
__global__ void __launch_bounds__(64) k_b3_events(const uint32_t *__restrict__ useful, uint32_t useful_words, uint32_t n_sites, const uint32_t *__restrict__ pair_off,
                                                  const uint2 *__restrict__ events, uint2 *__restrict__ work, uint32_t n_blocks) {
    const uint32_t tile = blockIdx.y, lane = threadIdx.x;
    const uint32_t pair = blockIdx.x * 64u + lane, site = pair >> 2, al = pair & 3u;
    const bool u = site < n_sites && ((useful[(uint64_t)tile * useful_words + (site >> 3)] >> ((site & 7u) * 4u + al)) & 1u);
    unsigned long long mask = __builtin_amdgcn_ballot_w64(u);
    uint32_t *W = (uint32_t *)(work + (uint64_t)tile * n_blocks);
    while (mask) {
        const uint32_t k = (uint32_t)__builtin_ctzll(mask);
        mask &= mask - 1ull;
        const uint32_t q = blockIdx.x * 64u + k;
        const uint32_t i0 = pair_off[q], i1 = pair_off[q + 1];
        for (uint32_t i = i0 + lane; i < i1; i += 64u) {
            const uint2 e = events[i];
            if (e.x == e.y) atomicAdd(&W[2u * e.x], 1u << 16);
            else { atomicAdd(&W[2u * e.x], 1u); atomicAdd(&W[2u * e.y + 1u], 1u); }
        }
    }
}

// sums of the start and end counts of one segment of B3_SEG blocks
__global__ void __launch_bounds__(256) k_b3_seg_sums(const uint2 *__restrict__ work, uint32_t n_blocks, uint32_t n_seg, uint2 *__restrict__ seg) {
    __shared__ uint32_t ss[256], se[256];
    const uint32_t tile = blockIdx.y, sg = blockIdx.x, t = threadIdx.x;
    const uint2 *W = work + (uint64_t)tile * n_blocks;
    uint32_t a = 0, b = 0;
    for (uint32_t i = t; i < B3_SEG; i += 256u) {   // (strided: coalesced 8-byte loads)
        const uint32_t blk = sg * B3_SEG + i;
        if (blk < n_blocks) { const uint2 w = W[blk]; a += w.x & 0xFFFFu; b += w.y; }
    }
    ss[t] = a; se[t] = b;
    __syncthreads();
    for (uint32_t o = 128; o > 0; o >>= 1) { if (t < o) { ss[t] += ss[t + o]; se[t] += se[t + o]; } __syncthreads(); }
    if (t == 0) seg[(uint64_t)tile * n_seg + sg] = make_uint2(ss[0], se[0]);
}

// exclusive scan of a tile's segment sums (one block per tile; a few hundred segments)
__global__ void __launch_bounds__(256) k_b3_seg_scan(uint2 *__restrict__ seg, uint32_t n_seg) {
    __shared__ uint32_t pa[256], pb[256];
    const uint32_t tile = blockIdx.x, t = threadIdx.x;
    uint2 *S = seg + (uint64_t)tile * n_seg;
    uint32_t base_a = 0, base_b = 0;
    for (uint32_t s0 = 0; s0 < n_seg; s0 += 256u) {
        const uint32_t i = s0 + t;
        const uint2 v = i < n_seg ? S[i] : make_uint2(0u, 0u);
        pa[t] = v.x; pb[t] = v.y;
        __syncthreads();
        for (uint32_t o = 1; o < 256u; o <<= 1) {   // (Hillis-Steele)
            const uint32_t xa = t >= o ? pa[t - o] : 0u, xb = t >= o ? pb[t - o] : 0u;
            __syncthreads();
            pa[t] += xa; pb[t] += xb;
            __syncthreads();
        }
        if (i < n_seg) S[i] = make_uint2(base_a + pa[t] - v.x, base_b + pb[t] - v.y);
        const uint32_t ta = pa[255], tb = pb[255];
        __syncthreads();
        base_a += ta; base_b += tb;
    }
}

// the tables of one segment of B3_SEG blocks: the counters are staged through LDS (coalesced loads and stores; a thread then owns 16
// consecutive blocks of the staged copy for the prefix sums)
__global__ void __launch_bounds__(256) k_b3_tables(const uint2 *__restrict__ work, uint32_t n_blocks, uint32_t n_seg, const uint2 *__restrict__ seg,
                                                   uint16_t *__restrict__ over, uint16_t *__restrict__ under, uint16_t *__restrict__ l1, uint32_t n_l1) {
    __shared__ uint2 wk[B3_SEG];          // 32 KB
    __shared__ uint32_t res[B3_SEG];      // 16 KB: over | under << 16
    __shared__ uint32_t pa[256], pb[256], mx[256];
    const uint32_t tile = blockIdx.y, sg = blockIdx.x, t = threadIdx.x;
    const uint2 *W = work + (uint64_t)tile * n_blocks;
    const uint32_t g0 = sg * B3_SEG;
    for (uint32_t i = t; i < B3_SEG; i += 256u) wk[i] = g0 + i < n_blocks ? W[g0 + i] : make_uint2(0u, 0u);
    __syncthreads();
    uint32_t a = 0, b = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) { const uint2 w = wk[t * 16 + k]; a += w.x & 0xFFFFu; b += w.y; }
    pa[t] = a; pb[t] = b;
    __syncthreads();
    for (uint32_t o = 1; o < 256u; o <<= 1) {
        const uint32_t xa = t >= o ? pa[t - o] : 0u, xb = t >= o ? pb[t - o] : 0u;
        __syncthreads();
        pa[t] += xa; pb[t] += xb;
        __syncthreads();
    }
    const uint2 base = seg[(uint64_t)tile * n_seg + sg];
    uint32_t S = base.x + pa[t] - a, E = base.y + pb[t] - b;   // inclusive prefixes in front of the thread's first block
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const uint2 w = wk[t * 16 + k];
        const uint32_t st = w.x & 0xFFFFu, same = w.x >> 16, en = w.y;
        const uint32_t ov = min(S + st - E + same, 65535u);   // S(b) - E(b - 1) + same(b)   (65535: "no bound" for the walk -- cannot be reached on the packed path)
        const uint32_t un = min(S - (E + en), 65535u);        // S(b - 1) - E(b)   (never negative: an event ends where or behind it starts)
        S += st; E += en;
        res[t * 16 + k] = ov | (un << 16);
        if (g0 + t * 16 + k < n_blocks) m = max(m, ov);
    }
    mx[t] = m;
    __syncthreads();
    uint16_t *O = over + (uint64_t)tile * n_blocks, *U = under + (uint64_t)tile * n_blocks;
    for (uint32_t i = t; i < B3_SEG / 2; i += 256u) {   // two blocks per thread and store: 4-byte stores, coalesced
        const uint32_t blk = g0 + 2u * i;
        const uint32_t r0 = res[2 * i], r1 = res[2 * i + 1];
        if (blk + 1 < n_blocks) {
            *(uint32_t *)(O + blk) = (r0 & 0xFFFFu) | (r1 << 16);
            *(uint32_t *)(U + blk) = (r0 >> 16) | (r1 & 0xFFFF0000u);
        } else if (blk < n_blocks) { O[blk] = (uint16_t)r0; U[blk] = (uint16_t)(r0 >> 16); }
    }
    if (t < 64) {   // level 1: 64 blocks = four threads
        const uint32_t v = max(max(mx[4 * t], mx[4 * t + 1]), max(mx[4 * t + 2], mx[4 * t + 3]));
        const uint32_t i = sg * 64u + t;
        if (i < n_l1) l1[(uint64_t)tile * n_l1 + i] = (uint16_t)v;
    }
}

// levels 2 and 3 of one tile (one block)
__global__ void __launch_bounds__(256) k_b3_levels(const uint16_t *__restrict__ l1, uint32_t n_l1, uint16_t *__restrict__ l2, uint32_t n_l2, uint16_t *__restrict__ l3,
                                                   uint32_t n_l3) {
    const uint32_t tile = blockIdx.x, t = threadIdx.x;
    const uint16_t *A = l1 + (uint64_t)tile * n_l1;
    uint16_t *B = l2 + (uint64_t)tile * n_l2, *C = l3 + (uint64_t)tile * n_l3;
    for (uint32_t i = t; i < n_l2; i += 256u) {
        uint32_t m = 0;
        for (uint32_t k = 0; k < 64u && i * 64u + k < n_l1; k++) m = max(m, (uint32_t)A[i * 64u + k]);
        B[i] = (uint16_t)m;
    }
    __syncthreads();
    __threadfence_block();
    for (uint32_t i = t; i < n_l3; i += 256u) {
        uint32_t m = 0;
        for (uint32_t k = 0; k < 64u && i * 64u + k < n_l2; k++) m = max(m, (uint32_t)B[i * 64u + k]);
        C[i] = (uint16_t)m;
    }
}

hipError_t launch_b3_events(const uint32_t *useful, uint32_t useful_words, uint32_t n_sites, uint32_t n_tiles, const uint32_t *pair_off, const uint32_t *events,
                            uint32_t *work, uint32_t n_blocks, hipStream_t s) {
    if (!n_sites || !n_tiles) return hipSuccess;
    hipLaunchKernelGGL(k_b3_events, dim3((n_sites * 4u + 63u) / 64u, n_tiles), dim3(64), 0, s, useful, useful_words, n_sites, pair_off, (const uint2 *)events, (uint2 *)work,
                       n_blocks);
    return hipGetLastError();
}

hipError_t launch_b3_tables(const uint32_t *work, uint32_t n_tiles, uint32_t n_blocks, uint32_t *seg, uint16_t *over, uint16_t *under, uint16_t *l1, uint16_t *l2,
                            uint16_t *l3, hipStream_t s) {
    if (!n_tiles || !n_blocks) return hipSuccess;
    const uint32_t n_seg = (n_blocks + B3_SEG - 1) / B3_SEG, n_l1 = b3_div64(n_blocks), n_l2 = b3_div64(n_l1), n_l3 = b3_div64(n_l2);
    hipLaunchKernelGGL(k_b3_seg_sums, dim3(n_seg, n_tiles), dim3(256), 0, s, (const uint2 *)work, n_blocks, n_seg, (uint2 *)seg);
    hipLaunchKernelGGL(k_b3_seg_scan, dim3(n_tiles), dim3(256), 0, s, (uint2 *)seg, n_seg);
    hipLaunchKernelGGL(k_b3_tables, dim3(n_seg, n_tiles), dim3(256), 0, s, (const uint2 *)work, n_blocks, n_seg, (const uint2 *)seg, over, under, l1, n_l1);
    hipLaunchKernelGGL(k_b3_levels, dim3(n_tiles), dim3(256), 0, s, l1, n_l1, l2, n_l2, l3, n_l3);
    return hipGetLastError();
}

}  // namespace ugp
