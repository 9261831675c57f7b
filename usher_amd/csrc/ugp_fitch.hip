// ugp_fitch.hip -- Fitch-Sankoff small-parsimony assignment of a batch of VCF sites
// onto a fixed tree, for gfx950.  Replaces mapper_body::operator()
// (src/usher_mapper.cpp:6-161) as driven by the VCF reader
// (src/mutation_annotated_tree.cpp:2108-2179).
//
// Formulation.  With unit substitution costs the Sankoff recurrence of :86-111
//     sc[p][b] += min_k( sc[c][k] + [k != b] )  =  m_c + [b not in F_c],
// m_c = min_k sc[c][k], F_c = argmin set of child c.  So the forward pass only needs the
// 4-bit set F per (node, site):  F_p = argmin over the bases allowed at p (all four; the
// genotype mask when the VCF has a column for p; {REF} for a leaf without a variant -- the
// "large value" num_nodes of :35-62 always exceeds the number of children) of
// cnt_p[b] = #{children c : b not in F_c}.  The backward pass (:114-157) keeps the parent's
// state when it is in F_n and otherwise takes the lowest base of F_n (the first strictly
// smaller score in the j = 0..3 scan).
//
// Layout.  F[node][W] u32, eight sites per word (one nibble each), nodes in breadth-first
// order so the children of a node -- and of the next internal node, and the next -- are W-word
// rows that lie back to back.  The unit of work is a (node, tile) item, a tile = 64 words = 512
// sites: lanes read consecutive words, so a row of a tile is one 256 B coalesced load.  Child
// counts are kept bit-sliced (plane k = bit k of all 32 (site, base) counters of the lane), so
// a child costs ~3*log2(c) VALU ops at a node with c children and the argmin is a bit-sliced
// tournament.  Levels are processed bottom-up then top-down, one launch per level; states
// overwrite F in place and the top-down sweep lists the (site, node) pairs whose state differs
// from the parent's.
//
// Rows that are never stored (round 6).  A leaf without a genotype cell among the 512 sites of a tile is {REF} there, and an
// internal node without one allows all four bases: neither needs a row of its own.  A byte map `mark8` (the item holds a cell of
// the VCF; plain stores) and a bitmap `stored` (marked, or the node is internal: the forward sweep writes its row), both
// tile-major, tell every kernel which rows exist; the others are read from row N of the table -- a copy of the reference word --
// or are all-ones in registers.  On the 10 M-node bench tree 55 % of the leaf items have no cell: the initialisation writes
// 2.8 GB instead of 10.2, the two sweeps read 7 of 10.2 GB of child rows.  HBM traffic of a call by the counters: 48 -> 29 GB.
//
// Where the time of a call goes at 10 M nodes x 2 048 sites x 13.7 M cells (round 6, profiles/r06_fitch_*): 10.9 ms = the
// caller's parent array and cells over PCIe 2.3 (pageable memory: the boundary), mark / initialise / scatter the cells 1.7,
// forward sweep 2.9 (3.9 TB/s on the wide levels), backward sweep 3.2 (4.8 TB/s), sorted result back to the host 0.7.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "usher_amd.h"

namespace ugp {
int set_error(int code, const std::string &msg);   // ugp_capi.cpp
}

namespace {

__device__ __forceinline__ uint32_t nib_any(uint32_t x) {   // 0xF in every nibble of x that is non-zero
    uint32_t t = (x | (x >> 1) | (x >> 2) | (x >> 3)) & 0x11111111u;
    return t * 15u;
}

__device__ __forceinline__ uint32_t nib_lowbit(uint32_t f) {   // lowest set bit of every nibble
    const uint32_t up = ((f << 1) & 0xEEEEEEEEu) | ((f << 2) & 0xCCCCCCCCu) | ((f << 3) & 0x88888888u);
    return f & ~up;
}

// Work items are (node, 64-word tile) pairs.  A block of four waves takes one tile of 4 x FS_* consecutive nodes (or internal
// nodes, in the forward sweep) and the next block the next tile of the same nodes: block b -> tile b % gy, node group b / gy,
// so neighbouring blocks read neighbouring 256-byte pieces of the same rows.  FS_* items per wave put that many independent
// row loads in flight behind ONE round of scalar look-ups.
constexpr int FS_NB = 8;

// the two bitmaps are tile-major: bit tile * npad + node (npad = the node count rounded up to 64), so the items of a wave --
// consecutive nodes of one tile -- are consecutive bits
__device__ __forceinline__ bool fs_bit(const uint32_t *__restrict__ bm, uint64_t b) { return (bm[b >> 5] >> (b & 31)) & 1u; }
__device__ __forceinline__ uint32_t fs_bits32(const uint32_t *__restrict__ bm, uint64_t b) {   // bits b .. b+31 (the bitmaps end in two spare words)
    const uint64_t two = (uint64_t)bm[b >> 5] | ((uint64_t)bm[(b >> 5) + 1] << 32);
    return (uint32_t)(two >> (b & 31));
}

// which (node, tile) items hold a cell: one BYTE per item (`mark8`, zeroed by the caller), set with a plain store -- 13.7 M
// atomicOr on a bitmap ran at 17 per nanosecond (0.8 ms of the call); k_fs_init folds the bytes into the `stored` bitmap.
// Cells are grouped by site; v_off[s] is the first cell of site s of this pass.
__global__ void k_fs_mark(const uint64_t *__restrict__ v_off, uint32_t n_sites, const uint32_t *__restrict__ v_node, uint64_t c0, uint64_t c1, uint64_t npad,
                          uint32_t n_nodes, uint8_t *__restrict__ mark8) {
    const uint64_t i = c0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // (cells [c0, c1): one piece of the upload)
    if (i >= c1) return;
    uint32_t lo = 0, hi = n_sites;   // last s with v_off[s] <= i
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (v_off[mid] <= i) lo = mid; else hi = mid;
    }
    const uint32_t n = v_node[i];
    if (n >= n_nodes) return;   // (k_fs_scatter reports it)
    mark8[(uint64_t)(lo >> 9) * npad + n] = 1;
}

// rows of the marked items start as {REF} (leaves) or "any base" (internal nodes); `stored` = marked | internal.  A wave takes 64
// consecutive nodes of one tile -- one 64-bit word of the bitmap: every lane looks up whether its node is marked / internal, then
// the wave writes the rows of the marked ones, one row per step.
__global__ __launch_bounds__(256) void k_fs_init(uint32_t *__restrict__ F, const uint32_t *__restrict__ refw,
                                                 const uint32_t *__restrict__ n_children, uint32_t n_nodes, uint32_t W, uint32_t y0, uint32_t ny, uint64_t npad,
                                                 const uint8_t *__restrict__ mark8, uint32_t *__restrict__ stored) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t y = y0 + blockIdx.x % ny, n0 = ((blockIdx.x / ny) * 4 + wave) * 64;   // (tiles [y0, y0 + ny): one piece of the upload)
    if (n0 >= n_nodes) return;
    const bool mine = n0 + lane < n_nodes;
    uint64_t mb = __builtin_amdgcn_ballot_w64(mine && mark8[(uint64_t)y * npad + n0 + lane] != 0);
    const uint64_t ib = __builtin_amdgcn_ballot_w64(mine && n_children[n0 + lane] != 0);
    if (lane == 0) ((uint64_t *)stored)[((uint64_t)y * npad + n0) >> 6] = mb | ib;
    const uint32_t w = y * 64 + lane;
    if (w >= W) return;
    const uint32_t r = refw[w];
    uint32_t *row = F + (uint64_t)n0 * W + w;
    while (mb) {
        const uint32_t u = __builtin_ctzll(mb);
        mb &= mb - 1;
        row[(uint64_t)u * W] = ((ib >> u) & 1) ? 0xFFFFFFFFu : r;
    }
}

// which of the forward sweep's nodes hold a cell of their own (their row exists and restricts the bases they may take): one dword
// per wave -- (tile, group of FS_FN internal nodes of a level; `goff` = the first group of every level), bit 4 * u for node u
constexpr int FS_FN = 8;
__global__ void k_fs_desc(const uint32_t *__restrict__ inodes, const uint32_t *__restrict__ ilvl_off, const uint32_t *__restrict__ goff, uint32_t n_levels,
                          uint32_t n_groups, uint32_t y0, uint32_t ny, uint64_t npad, const uint8_t *__restrict__ mark8, uint32_t *__restrict__ desc) {
    static_assert(FS_FN == 8, "eight lanes fill one descriptor");
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;   // lane = (tile, group, node of the group)
    const uint64_t ig = i >> 3;
    const uint32_t u = (uint32_t)i & 7;
    uint32_t x = 0;
    if (ig < (uint64_t)n_groups * ny) {
        const uint32_t y = y0 + (uint32_t)(ig / n_groups), g = (uint32_t)(ig % n_groups);
        uint32_t lo = 0, hi = n_levels;   // last level L with goff[L] <= g
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (goff[mid] <= g) lo = mid; else hi = mid;
        }
        const uint32_t k = ilvl_off[lo] + (g - goff[lo]) * FS_FN + u;
        if (k < ilvl_off[lo + 1] && mark8[(uint64_t)y * npad + inodes[k]]) x = 1u << (4 * u);
    }
    x |= __shfl_xor(x, 1); x |= __shfl_xor(x, 2); x |= __shfl_xor(x, 4);
    if (u == 0 && ig < (uint64_t)n_groups * ny) desc[(uint64_t)y0 * n_groups + ig] = x;
}

// genotype cells of tree nodes: replace the initial nibble by the allele mask (:47-62).
// Cells are grouped by site; v_off[s] is the first cell of site s of this pass.
__global__ void k_fs_scatter(uint32_t *__restrict__ F, const uint32_t *__restrict__ refw, const uint32_t *__restrict__ n_children,
                             const uint64_t *__restrict__ v_off, uint32_t n_sites, const uint32_t *__restrict__ v_node,
                             const uint8_t *__restrict__ v_nuc, uint64_t c0, uint64_t c1, uint32_t W, uint32_t n_nodes, uint32_t *__restrict__ flags) {
    const uint64_t i = c0 + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c1) return;
    uint32_t lo = 0, hi = n_sites;   // last s with v_off[s] <= i
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (v_off[mid] <= i) lo = mid; else hi = mid;
    }
    const uint32_t s = lo, n = v_node[i];
    // the checks the host used to run over every cell: flags bit 0 = a node index out of range, bit 1 = an allele mask outside
    // 1..15, bit 2 = the node indices of a site do not ascend (a node may then be named twice: the caller filters and comes again)
    const uint32_t a = v_nuc[i];
    if (n >= n_nodes) { atomicOr(flags, 1u); return; }
    if ((a & 0xFu) == 0 || a > 15u) { atomicOr(flags, 2u); return; }
    if (i > v_off[s] && n <= v_node[i - 1]) atomicOr(flags, 4u);
    const uint32_t w = s >> 3, sh = (s & 7) * 4;
    const uint32_t old = n_children[n] ? 0xFu : ((refw[w] >> sh) & 0xFu);
    const uint32_t x = (old ^ (a & 0xFu)) << sh;
    if (x) atomicXor(&F[(uint64_t)n * W + w], x);
}

// forward pass for the internal nodes of one level (:86-111).  A wave takes one tile of a GROUP of FS_FN internal nodes in a row.
// In breadth-first order the children of consecutive internal nodes are consecutive rows, so the group's children are ONE run of
// rows: the wave streams it, eight row loads per batch with the next batch requested before the current one is counted, and the
// nodes are segments of the stream -- a node is finished (the argmin over its allowed bases, one store) when its last child has
// been counted.  Nodes with 2 children and nodes with 200 go through the same code at the same depth of loads in flight.
// What the wave has to know first is independent SCALAR loads -- three s_load_dwordx8 from arrays indexed by the node's rank among
// the internal nodes, one descriptor dword -- and a row that does not exist is read from row n_nodes of the table, a copy of the
// reference word, in 32-bit address arithmetic (the caller keeps (n_nodes + 1) * W below 2^32): "no row" is a select between two
// row indices.  (History of this kernel, round 6: items with a byte-load descriptor each ran one after the other, 15 us per wave
// -- a vector load's wait drains the rows already requested; then up to three children per node in straight-line code and a
// one-row-ahead loop for the rest -- which is a fifth of the nodes and a third of the rows of the bench tree, half / three
// quarters of a SARS-CoV-2-shaped one.)
//
// Child counts are bit-sliced (plane k = bit k of the lane's 32 (site, base) counters); KP planes count to 2^KP - 1, and a group is
// run with 3, 5 or 8 of them by its largest node (up to FS_WIDE = 255 children; larger nodes: "Polytomies" below).
__device__ __forceinline__ uint32_t fs_pick(uint32_t cand, uint32_t plane) {   // keep the candidates whose counter has a 0 in this plane, if any
    const uint32_t z = cand & ~plane;
    const uint32_t m = nib_any(z);
    return (z & m) | (cand & ~m);
}

// Polytomies.  A node with more than FS_WIDE children is not part of any stream -- one wave counting 10 000 rows eight at a time
// was the whole duration of the upper levels of a SARS-CoV-2-shaped tree (the seven topmost launches 2.5 of the sweep's 6.2 ms) --
// its children are cut into chunks of FS_CHUNK rows, k_fs_wide_count counts each chunk in a wave of its own (8 planes to a scratch
// buffer), k_fs_wide_final adds the chunks' counters bit-sliced and finishes the node.  The list of such nodes is made by
// k_fs_topo, the chunk tables by the host (they are few).
constexpr uint32_t FS_WIDE = 255, FS_CHUNK = 64;   // (a chunk: 8 batches of loads, ~12 us of one wave)

// one run of rows [row0, row0 + n_rows) = the children of the nodes at the head of the queue (pp / ncs / dd: node, child count,
// own-row bit; a node is popped when its last child has been counted)
#ifndef UGP_FS_B
#define UGP_FS_B 8
#endif
constexpr int FS_B = UGP_FS_B;   // rows per batch of the forward stream (8 or 16: the bitmap window is 32 bits; 16 measured: both sweeps' kernels 6.0 -> 6.5 ms)
template <int KP>
__device__ __forceinline__ void fs_stream(uint32_t *__restrict__ F, uint32_t w, uint32_t W, uint32_t ref_row, uint32_t (&pp)[FS_FN],
                                          uint32_t (&ncs)[FS_FN], uint32_t &dd, uint32_t row0, uint32_t n_rows,
                                          const uint32_t *__restrict__ stored, uint64_t bbase) {
    uint32_t plane[KP];
#pragma unroll
    for (int k = 0; k < KP; k++) plane[k] = 0;
    uint32_t xa[FS_B], xb[FS_B];
#pragma unroll
    for (int j = 0; j < FS_B; j++) xb[j] = 0;
    uint32_t rem = ncs[0];
    uint32_t have = fs_bits32(stored, bbase), have_next = fs_bits32(stored, bbase + FS_B);
#pragma unroll
    for (int j = 0; j < FS_B; j++) {
        const uint32_t r = ((uint32_t)j < n_rows && ((have >> j) & 1u)) ? row0 + j : ref_row;
        xa[j] = F[r * W + w];
    }
#pragma nounroll
    for (uint32_t i = 0; i < n_rows; i += FS_B) {
        const uint32_t have_after = fs_bits32(stored, bbase + i + 2 * FS_B);   // (for the batch after the next: a round trip ahead of its use)
        if (i + FS_B < n_rows) {
#pragma unroll
            for (int j = 0; j < FS_B; j++) {
                const uint32_t r = (i + FS_B + j < n_rows && ((have_next >> j) & 1u)) ? row0 + i + FS_B + j : ref_row;
                xb[j] = F[r * W + w];
            }
        }
#pragma unroll
        for (int j = 0; j < FS_B; j++) {
            if (i + j >= n_rows) break;
            uint32_t carry = ~xa[j];   // +1 for every (site, base) with base not in F_c
#pragma unroll
            for (int k = 0; k < KP; k++) {
                const uint32_t t = plane[k] & carry;
                plane[k] ^= carry;
                carry = t;
            }
            if (--rem == 0) {   // the node's last child: argmin over the allowed bases (all four, or the node's own genotype mask)
                uint32_t cand = 0xFFFFFFFFu;
                if (dd & 1u) cand = F[pp[0] * W + w];
#pragma unroll
                for (int k = KP - 1; k >= 0; k--) {
                    cand = fs_pick(cand, plane[k]);
                    plane[k] = 0;
                }
                F[pp[0] * W + w] = cand;
#pragma unroll
                for (int q = 0; q + 1 < FS_FN; q++) { pp[q] = pp[q + 1]; ncs[q] = ncs[q + 1]; }
                ncs[FS_FN - 1] = 0;
                dd >>= 4;
                rem = ncs[0];
            }
        }
#pragma unroll
        for (int j = 0; j < FS_B; j++) xa[j] = xb[j];
        have_next = have_after;
    }
}

__global__ __launch_bounds__(256) void k_fs_forward(uint32_t *__restrict__ F, const uint32_t *__restrict__ nodes, const uint32_t *__restrict__ ifirst,
                                                    const uint32_t *__restrict__ inch, const uint32_t *__restrict__ desc, uint32_t n_groups,
                                                    uint32_t n_level, uint32_t W, uint32_t gy, uint64_t npad, const uint32_t *__restrict__ stored,
                                                    uint32_t n_nodes) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t y = blockIdx.x % gy, grp = (blockIdx.x / gy) * 4 + wave, idx0 = grp * FS_FN;
    if (idx0 >= n_level) return;
    const uint32_t w = y * 64 + lane;
    if (w >= W) return;
    uint32_t dd = desc[(uint64_t)y * n_groups + grp];
    uint32_t row0 = ifirst[idx0];
    uint32_t pp[FS_FN], ncs[FS_FN];
#pragma unroll
    for (int u = 0; u < FS_FN; u++) {   // (past the end of a level lie the next level's entries, and eight spare ones at the very end)
        pp[u] = nodes[idx0 + u];
        ncs[u] = inch[idx0 + u];
    }
#pragma unroll
    for (int u = 0; u < FS_FN; u++)
        if (idx0 + u >= n_level) ncs[u] = 0;
    // the group's children are one run of rows; a polytomy cuts it (its rows belong to k_fs_wide_*): usually one turn of this loop
#pragma nounroll
    while (ncs[0] != 0) {
        if (ncs[0] > FS_WIDE) {
            row0 += ncs[0];
#pragma unroll
            for (int q = 0; q + 1 < FS_FN; q++) { pp[q] = pp[q + 1]; ncs[q] = ncs[q + 1]; }
            ncs[FS_FN - 1] = 0;
            dd >>= 4;
            continue;
        }
        uint32_t n_rows = 0, widest = 0;
        bool open = true;
#pragma unroll
        for (int u = 0; u < FS_FN; u++) {
            open = open && ncs[u] != 0 && ncs[u] <= FS_WIDE;
            if (open) { n_rows += ncs[u]; widest = max(widest, ncs[u]); }
        }
        const uint64_t bbase = (uint64_t)y * npad + row0;
        if (widest <= 7) fs_stream<3>(F, w, W, n_nodes, pp, ncs, dd, row0, n_rows, stored, bbase);
        else if (widest <= 31) fs_stream<5>(F, w, W, n_nodes, pp, ncs, dd, row0, n_rows, stored, bbase);
        else fs_stream<8>(F, w, W, n_nodes, pp, ncs, dd, row0, n_rows, stored, bbase);
        row0 += n_rows;
    }
}

// one chunk of a polytomy's children (at most FS_CHUNK rows) for one tile: counts in 8 planes -> part[(chunk, tile)][plane][lane]
__global__ __launch_bounds__(256) void k_fs_wide_count(const uint32_t *__restrict__ F, const uint32_t *__restrict__ chunk_first, const uint32_t *__restrict__ chunk_rows,
                                                       uint32_t n_chunks, uint32_t W, uint32_t gy, uint64_t npad, const uint32_t *__restrict__ stored,
                                                       uint32_t n_nodes, uint32_t *__restrict__ part) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t y = blockIdx.x % gy, ch = (blockIdx.x / gy) * 4 + wave;
    if (ch >= n_chunks) return;
    const uint32_t w = y * 64 + lane;
    if (w >= W) return;
    const uint32_t row0 = chunk_first[ch], n_rows = chunk_rows[ch];
    const uint64_t bbase = (uint64_t)y * npad + row0;
    uint32_t plane[8];
#pragma unroll
    for (int k = 0; k < 8; k++) plane[k] = 0;
    uint32_t xa[8], xb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint32_t have = fs_bits32(stored, bbase), have_next = fs_bits32(stored, bbase + 8);
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const uint32_t r = ((uint32_t)j < n_rows && ((have >> j) & 1u)) ? row0 + j : n_nodes;
        xa[j] = F[r * W + w];
    }
#pragma nounroll
    for (uint32_t i = 0; i < n_rows; i += 8) {
        const uint32_t have_after = fs_bits32(stored, bbase + i + 16);
        if (i + 8 < n_rows) {
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t r = (i + 8 + j < n_rows && ((have_next >> j) & 1u)) ? row0 + i + 8 + j : n_nodes;
                xb[j] = F[r * W + w];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) {
            uint32_t carry = i + j < n_rows ? ~xa[j] : 0u;
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const uint32_t t = plane[k] & carry;
                plane[k] ^= carry;
                carry = t;
            }
        }
#pragma unroll
        for (int j = 0; j < 8; j++) xa[j] = xb[j];
        have_next = have_after;
    }
    uint32_t *out = part + ((uint64_t)ch * gy + y) * 8 * 64 + lane;
#pragma unroll
    for (int k = 0; k < 8; k++) out[k * 64] = plane[k];
}

// a polytomy for one tile: the sum of its chunks' counters (bit-sliced addition, 8 planes into 32), then the argmin as everywhere
__global__ __launch_bounds__(256) void k_fs_wide_final(uint32_t *__restrict__ F, const uint32_t *__restrict__ wide_node, const uint32_t *__restrict__ wide_nc,
                                                       const uint32_t *__restrict__ wide_chunk0, uint32_t n_wide, uint32_t W, uint32_t gy, uint64_t npad,
                                                       const uint8_t *__restrict__ mark8, const uint32_t *__restrict__ part) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t y = blockIdx.x % gy, i = (blockIdx.x / gy) * 4 + wave;
    if (i >= n_wide) return;
    const uint32_t w = y * 64 + lane;
    if (w >= W) return;
    const uint32_t p = wide_node[i], nc = wide_nc[i], c0 = wide_chunk0[i], n_ch = (nc + FS_CHUNK - 1) / FS_CHUNK;
    const int K = 32 - __builtin_clz(nc);
    uint32_t acc[32];
#pragma unroll
    for (int k = 0; k < 32; k++) acc[k] = 0;
    auto add = [&](const uint32_t (&x)[8]) {
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            const uint32_t h = acc[k] ^ x[k];
            const uint32_t cy = (acc[k] & x[k]) | (carry & h);
            acc[k] = h ^ carry;
            carry = cy;
        }
#pragma unroll
        for (int k = 8; k < 32; k++) {
            if (k >= K) break;
            const uint32_t t = acc[k] & carry;
            acc[k] ^= carry;
            carry = t;
        }
    };
    for (uint32_t c = 0; c < n_ch; c += 2) {   // two chunks' planes in flight
        const bool two = c + 1 < n_ch;
        const uint32_t *in = part + ((uint64_t)(c0 + c) * gy + y) * 8 * 64 + lane;
        const uint32_t *in2 = part + ((uint64_t)(c0 + c + (two ? 1u : 0u)) * gy + y) * 8 * 64 + lane;
        uint32_t x[8], z[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { x[k] = in[k * 64]; z[k] = in2[k * 64]; }
        add(x);
        if (two) add(z);
    }
    uint32_t cand = mark8[(uint64_t)y * npad + p] ? F[(uint64_t)p * W + w] : 0xFFFFFFFFu;
#pragma unroll
    for (int k = 31; k >= 0; k--) {
        if (k >= K) continue;
        cand = fs_pick(cand, acc[k]);
    }
    F[(uint64_t)p * W + w] = cand;
}

// Listed changes go to FS_SEG independent segments of the output buffer (cursor + base per segment):
// atomics on a single address run at ~90 per microsecond on this part, which would otherwise bound the pass.
constexpr uint32_t FS_SEG = 2048;

__device__ __forceinline__ void fs_emit(uint32_t d, uint32_t s, uint32_t sp, uint32_t n, uint32_t site0,
                                        unsigned long long *__restrict__ cursor, unsigned long long cap,
                                        uint64_t *__restrict__ out_key, uint8_t *__restrict__ out_val) {
    const unsigned long long at = atomicAdd(cursor, (unsigned long long)__builtin_popcount(d));
    uint32_t k = 0;
    while (d) {
        const uint32_t sh = __builtin_ctz(d);
        d &= d - 1;
        if (at + k < cap) {
            out_key[at + k] = ((uint64_t)(site0 + (sh >> 2)) << 32) | n;
            out_val[at + k] = (uint8_t)((((sp >> sh) & 0xFu) << 4) | ((s >> sh) & 0xFu));
        }
        k++;
    }
}

// backward pass for all nodes of one level (:114-141); states replace F in place (only internal
// nodes are read again, so leaves are not stored) and state changes are listed as they are found
// (:143-156) in segment (wave index mod FS_SEG).  cursor[seg * 8] ends as the number of changes of
// the segment; entries beyond `seg_cap` are dropped and the caller falls back to k_fs_emit.
__global__ __launch_bounds__(256) void k_fs_backward(uint32_t *__restrict__ F, const uint32_t *__restrict__ refw,
                                                     const uint32_t *__restrict__ parent, const uint32_t *__restrict__ n_children,
                                                     uint32_t lvl_begin, uint32_t lvl_end, uint32_t W, uint32_t gy, uint64_t npad, uint32_t site_base,
                                                     unsigned long long *__restrict__ cursor, unsigned long long seg_cap,
                                                     uint64_t *__restrict__ out_key, uint8_t *__restrict__ out_val,
                                                     const uint32_t *__restrict__ stored) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t seg = (blockIdx.x * 4 + wave) & (FS_SEG - 1);
    cursor += seg * 8;
    out_key += seg * seg_cap;
    out_val += seg * seg_cap;
    const uint32_t y = blockIdx.x % gy, n0 = lvl_begin + ((blockIdx.x / gy) * 4 + wave) * FS_NB;
    if (n0 >= lvl_end) return;
    const uint32_t w = y * 64 + lane;
    if (w >= W) return;
    const uint32_t have = fs_bits32(stored, (uint64_t)y * npad + n0);   // (a leaf without a cell in the tile has no row: it is {REF})
    const uint32_t r = refw[w];
    uint32_t f[FS_NB], sp[FS_NB];
    bool ok[FS_NB], keep_row[FS_NB];
#pragma unroll
    for (int u = 0; u < FS_NB; u++) {
        const uint32_t n = n0 + u;
        ok[u] = n < lvl_end;
        f[u] = 0; sp[u] = 0; keep_row[u] = false;
        if (ok[u]) {
            const uint32_t par = parent[n];
            keep_row[u] = n_children[n] != 0;
            sp[u] = par == 0xFFFFFFFFu ? r : F[(uint64_t)par * W + w];
            f[u] = *(((have >> u) & 1u) ? F + (uint64_t)n * W + w : refw + w);
        }
    }
#pragma unroll
    for (int u = 0; u < FS_NB; u++) {
        if (!ok[u]) continue;
        const uint32_t keep = nib_any(f[u] & sp[u]);
        const uint32_t s = (sp[u] & keep) | (nib_lowbit(f[u]) & ~keep);
        if (keep_row[u]) F[(uint64_t)(n0 + u) * W + w] = s;
        const uint32_t d = nib_any(s ^ sp[u]) & 0x11111111u;
        if (d) fs_emit(d, s, sp[u], n0 + u, site_base + w * 8, cursor, seg_cap, out_key, out_val);
    }
}

// fallback listing pass when the buffer given to k_fs_backward was too small.  Internal rows hold
// states, leaf rows still hold their sets, so a leaf's state is derived again.
__global__ __launch_bounds__(256) void k_fs_emit(const uint32_t *__restrict__ F, const uint32_t *__restrict__ refw,
                                                 const uint32_t *__restrict__ parent, const uint32_t *__restrict__ n_children,
                                                 uint32_t n_nodes, uint32_t W, uint32_t gy, uint64_t npad, uint32_t site_base,
                                                 unsigned long long *__restrict__ cursor, unsigned long long cap,
                                                 uint64_t *__restrict__ out_key, uint8_t *__restrict__ out_val,
                                                 const uint32_t *__restrict__ stored) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t y = blockIdx.x % gy, n = (blockIdx.x / gy) * 4 + wave;
    const uint32_t w = y * 64 + lane;
    if (n >= n_nodes || w >= W) return;
    const uint32_t par = parent[n];
    const uint32_t sp = par == 0xFFFFFFFFu ? refw[w] : F[(uint64_t)par * W + w];
    uint32_t s = fs_bit(stored, (uint64_t)y * npad + n) ? F[(uint64_t)n * W + w] : refw[w];
    if (!n_children[n]) {
        const uint32_t keep = nib_any(s & sp);
        s = (sp & keep) | (nib_lowbit(s) & ~keep);
    }
    const uint32_t d = nib_any(s ^ sp) & 0x11111111u;
    if (d) fs_emit(d, s, sp, n, site_base + w * 8, cursor, cap, out_key, out_val);
}

// gather the segments into one dense list
__global__ void k_fs_compact(const uint64_t *__restrict__ seg_key, const uint8_t *__restrict__ seg_val,
                             const unsigned long long *__restrict__ seg_begin, unsigned long long seg_cap,
                             uint64_t *__restrict__ out_key, uint8_t *__restrict__ out_val) {
    const uint32_t seg = blockIdx.x;
    const unsigned long long b = seg_begin[seg], e = seg_begin[seg + 1];
    for (unsigned long long i = threadIdx.x; i < e - b; i += blockDim.x) {
        out_key[b + i] = seg_key[seg * seg_cap + i];
        out_val[b + i] = seg_val[seg * seg_cap + i];
    }
}

// the sorted list as the four arrays the caller gets
__global__ void k_fs_split(const uint64_t *__restrict__ key, const uint8_t *__restrict__ val, uint64_t n, uint32_t *__restrict__ site,
                           uint32_t *__restrict__ node, uint8_t *__restrict__ par, uint8_t *__restrict__ nuc) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t k = key[i];
    const uint8_t v = val[i];
    site[i] = (uint32_t)(k >> 32); node[i] = (uint32_t)k; par[i] = v >> 4; nuc[i] = v & 0xF;
}

// Device buffers are kept from call to call (a pool per device, below): alloc() only ever grows them.
template <typename T>
struct Dev {
    T *p = nullptr;
    size_t cap = 0;
    ~Dev() { if (p) (void)hipFree(p); }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
    hipError_t alloc(size_t n) {
        if (n <= cap && p) return hipSuccess;
        if (p) { (void)hipFree(p); p = nullptr; cap = 0; }
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = std::max<size_t>(n, 1); else p = nullptr;
        return e;
    }
    hipError_t upload(const T *src, size_t n, hipStream_t st = nullptr) {
        hipError_t e = alloc(n);
        if (e != hipSuccess || n == 0) return e;
        return hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, st);
    }
};

// ---- topology on the device (round 5): the caller's breadth-first parent array is non-decreasing, so the children of a node are
// one run of it.  k_fs_heads checks the order and flags the run heads; a stream compaction lists them (heads[k] = first child of
// the k-th internal node); k_fs_topo turns the list into n_children and, by the node's rank among the internal nodes (index = level order), node / first child / child count.
// The host used to do this in three sequential passes over the 10M-entry array: 90 ms of a call whose kernels take 14.
__global__ void k_fs_heads(const uint32_t *__restrict__ parent, uint32_t n, uint8_t *__restrict__ flag, uint32_t *__restrict__ bad) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    if (j == 0) { flag[0] = 0; return; }
    const uint32_t p = parent[j], q = j > 1 ? parent[j - 1] : 0u;
    if (p >= j || (j > 1 && p < q)) atomicOr(bad, 1u);
    flag[j] = (j == 1 || p != q) ? 1 : 0;
}
__global__ void k_fs_topo(const uint32_t *__restrict__ parent, uint32_t n, const uint32_t *__restrict__ heads, const uint32_t *__restrict__ n_heads_p,
                          uint32_t *__restrict__ n_children, uint32_t *__restrict__ inodes,
                          uint32_t *__restrict__ ifirst, uint32_t *__restrict__ inch, uint32_t *__restrict__ n_wide_p, uint32_t *__restrict__ wide, uint32_t wide_cap) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x, nh = *n_heads_p;
    if (k >= nh) return;
    const uint32_t j = heads[k], e = k + 1 < nh ? heads[k + 1] : n, p = parent[j];
    n_children[p] = e - j; inodes[k] = p;
    ifirst[k] = j; inch[k] = e - j;   // (the same by the node's rank among the internal nodes: what the forward sweep indexes)
    if (e - j > FS_WIDE) {   // a polytomy: listed for k_fs_wide_* (any order; the host sorts the few there are)
        const uint32_t at = atomicAdd(n_wide_p, 1u);
        if (at < wide_cap) { wide[3 * (uint64_t)at] = p; wide[3 * (uint64_t)at + 1] = j; wide[3 * (uint64_t)at + 2] = e - j; }
    }
}
// ilvl_off[L] = internal nodes in front of level L = run heads among the nodes in front of level L + 1 (their children)
__global__ void k_fs_level_ranks(const uint32_t *__restrict__ heads, const uint32_t *__restrict__ n_heads_p, const uint32_t *__restrict__ lvl_off, uint32_t n_levels,
                                 uint32_t *__restrict__ ilvl_off) {
    const uint32_t L = blockIdx.x * blockDim.x + threadIdx.x;
    if (L > n_levels) return;
    const uint32_t nh = *n_heads_p;
    if (L == n_levels) { ilvl_off[L] = nh; return; }
    const uint32_t lim = lvl_off[L + 1];   // children of levels < L sit in front of lvl_off[L + 1]
    uint32_t lo = 0, hi = nh;              // first k with heads[k] >= lim
    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (heads[mid] < lim) lo = mid + 1; else hi = mid; }
    ilvl_off[L] = lo;
}

constexpr int FS_PIECES = 4;
struct FsPool {   // one per device, kept until the process ends: a call allocates nothing in the steady state
    std::mutex mu;
    // The caller's cells go up on a stream of their own, in pieces (see the per-pass preparation).  A process has four hardware queues
    // by default, and a stream that stays around shifts the queues of every stream made after it ("side stream"): this one
    // is kept from call to call (making and destroying it costs 0.35 ms of a 10 ms call) but dropped by ugp_fitch_release and whenever
    // a placement handle of the device is about to make its own streams (ugp::fitch_drop_streams, from ugp_capi.cpp).
    hipStream_t copy = nullptr;
    hipEvent_t up[FS_PIECES] = {};
    void drop_copy_stream() {
        if (!copy) return;
        (void)hipStreamSynchronize(copy);
        for (int k = 0; k < FS_PIECES; k++) if (up[k]) { (void)hipEventDestroy(up[k]); up[k] = nullptr; }
        (void)hipStreamDestroy(copy);
        copy = nullptr;
    }
    Dev<uint32_t> d_parent, d_nchild, d_inodes, d_heads, d_small, d_levels, d_F, d_refw, d_vnode, d_stored, d_osite, d_onode, d_ifirst, d_inch, d_desc, d_wide, d_wtab, d_part;
    Dev<uint8_t> d_flag, d_vnuc, d_oval, d_oval2, d_tmp, d_sel_tmp, d_opar, d_onuc, d_mark8;
    Dev<uint64_t> d_okey, d_okey2, d_voff;
    Dev<unsigned long long> d_cnt, d_segb;
};
FsPool *fs_pool(int device) {
    static std::mutex mu;
    static std::vector<FsPool *> pools;
    std::lock_guard<std::mutex> g(mu);
    if ((size_t)device >= pools.size()) pools.resize((size_t)device + 1, nullptr);
    if (!pools[device]) pools[device] = new FsPool();
    return pools[device];
}

}  // namespace

// the pool keeps up to 4 GiB of row storage and the sort buffers per device for the next call; a caller that is done
// building (the front end behind `-t`, before it places samples on the same device) hands them back.
namespace ugp {
void fitch_drop_streams(int device) {
    if (device < 0) return;
    FsPool *p = fs_pool(device);
    std::lock_guard<std::mutex> g(p->mu);
    if (p->copy && hipSetDevice(device) == hipSuccess) p->drop_copy_stream();
}
}  // namespace ugp

extern "C" void ugp_fitch_release(int device) {
    if (device < 0) return;
    FsPool *p = fs_pool(device);
    std::lock_guard<std::mutex> g(p->mu);
    if (hipSetDevice(device) != hipSuccess) return;
    (void)hipDeviceSynchronize();
    p->drop_copy_stream();
    for (Dev<uint32_t> *d : {&p->d_parent, &p->d_nchild, &p->d_inodes, &p->d_heads, &p->d_small, &p->d_levels, &p->d_F, &p->d_refw, &p->d_vnode,
                            &p->d_stored, &p->d_osite, &p->d_onode, &p->d_ifirst, &p->d_inch, &p->d_desc, &p->d_wide, &p->d_wtab, &p->d_part}) d->release();
    for (Dev<uint8_t> *d : {&p->d_flag, &p->d_vnuc, &p->d_oval, &p->d_oval2, &p->d_tmp, &p->d_sel_tmp, &p->d_opar, &p->d_onuc, &p->d_mark8}) d->release();
    for (Dev<uint64_t> *d : {&p->d_okey, &p->d_okey2, &p->d_voff}) d->release();
    for (Dev<unsigned long long> *d : {&p->d_cnt, &p->d_segb}) d->release();
}

struct ugp_fitch {
    std::vector<uint32_t> site, node;
    std::vector<uint8_t> par, nuc;
};

#define FS_TRY(expr)                                                                                         \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return ugp::set_error(UGP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

extern "C" int ugp_fitch_sankoff(int device, uint64_t n_nodes, const uint32_t *parent, const ugp_sites *sites, ugp_fitch **out) {
    if (!out) return ugp::set_error(UGP_ERR_INVALID, "null output handle");
    *out = nullptr;
    if (!parent || !sites || n_nodes == 0) return ugp::set_error(UGP_ERR_INVALID, "null or empty tree / site arrays");
    if (n_nodes >= 0xFFFFFFFFull) return ugp::set_error(UGP_ERR_UNSUPPORTED, "more than 2^32-2 nodes");
    const uint64_t S = sites->n_sites;
    if (S >= (1ull << 31)) return ugp::set_error(UGP_ERR_UNSUPPORTED, "more than 2^31 sites");
    if (S && (!sites->ref || !sites->var_off)) return ugp::set_error(UGP_ERR_INVALID, "null site arrays");
    const uint64_t n_var = S ? sites->var_off[S] : 0;
    if (n_var && (!sites->var_node || !sites->var_nuc)) return ugp::set_error(UGP_ERR_INVALID, "null variant arrays");
    // topology: breadth-first order means parent[] is non-decreasing and children are contiguous
    if (parent[0] != 0xFFFFFFFFu) return ugp::set_error(UGP_ERR_INVALID, "parent[0] must be the root (UINT32_MAX)");
    auto t_last = std::chrono::steady_clock::now();
    auto lap = [&]() {
        auto now = std::chrono::steady_clock::now();
        double ms = std::chrono::duration<double, std::milli>(now - t_last).count();
        t_last = now;
        return ms;
    };
    double t_topo = 0, t_alloc = 0, t_prep = 0, t_kern = 0, t_out = 0;
    const uint32_t N = (uint32_t)n_nodes;
    for (uint64_t s = 0; s < S; s++) {
        const uint8_t r = sites->ref[s];
        if (r != 1 && r != 2 && r != 4 && r != 8) return ugp::set_error(UGP_ERR_INVALID, "site reference allele is not one of A,C,G,T");
        if (sites->var_off[s + 1] < sites->var_off[s]) return ugp::set_error(UGP_ERR_INVALID, "var_off is not monotone");
    }
    ugp_fitch *res = new (std::nothrow) ugp_fitch();
    if (!res) return ugp::set_error(UGP_ERR_NOMEM, "out of host memory");
    struct Guard { ugp_fitch *r; ~Guard() { delete r; } } guard{res};

    FS_TRY(hipSetDevice(device));
    hipStream_t stream = nullptr;
    FsPool &P = *fs_pool(device);
    std::lock_guard<std::mutex> pool_lock(P.mu);   // (calls on one device take turns: they share the pooled buffers)
    auto &d_parent = P.d_parent; auto &d_nchild = P.d_nchild; auto &d_inodes = P.d_inodes;
    // ---- topology, on the device: run heads of parent[] -> first child / child count / internal nodes in level order.  The level
    // boundaries themselves are a handful of binary searches in the caller's array (a level is an index range; its children are
    // the nodes whose parent lies in it), meaningful once the device has confirmed the order.
    FS_TRY(d_parent.upload(parent, N, stream));
    FS_TRY(d_nchild.alloc(N)); FS_TRY(d_inodes.alloc((size_t)N + 8)); FS_TRY(P.d_heads.alloc(N)); FS_TRY(P.d_flag.alloc(N));
    FS_TRY(P.d_ifirst.alloc((size_t)N + 8)); FS_TRY(P.d_inch.alloc((size_t)N + 8));   // (+8: the forward sweep reads whole groups)
    FS_TRY(P.d_small.alloc(16));   // [0] order violated, [1] run heads, [2] cell flags, [3] polytomies
    const uint32_t wide_cap = N / (FS_WIDE + 1) + 1;   // (more nodes with more than FS_WIDE children each do not fit in N)
    FS_TRY(P.d_wide.alloc(3 * (size_t)wide_cap));
    FS_TRY(hipMemsetAsync(P.d_small.p, 0, 8 * sizeof(uint32_t), stream));
    FS_TRY(hipMemsetAsync(d_nchild.p, 0, (size_t)N * 4, stream));
    hipLaunchKernelGGL(k_fs_heads, dim3((N + 255) / 256), dim3(256), 0, stream, d_parent.p, N, P.d_flag.p, P.d_small.p);
    {
        size_t sel_bytes = 0;
        rocprim::counting_iterator<uint32_t> idx(0);
        FS_TRY(rocprim::select(nullptr, sel_bytes, idx, P.d_flag.p, P.d_heads.p, P.d_small.p + 1, (size_t)N, stream));
        FS_TRY(P.d_sel_tmp.alloc(sel_bytes));
        FS_TRY(rocprim::select(P.d_sel_tmp.p, sel_bytes, idx, P.d_flag.p, P.d_heads.p, P.d_small.p + 1, (size_t)N, stream));
    }
    hipLaunchKernelGGL(k_fs_topo, dim3((N + 255) / 256), dim3(256), 0, stream, d_parent.p, N, P.d_heads.p, P.d_small.p + 1, d_nchild.p, d_inodes.p,
                       P.d_ifirst.p, P.d_inch.p, P.d_small.p + 3, P.d_wide.p, wide_cap);
    std::vector<uint32_t> lvl_off{0, 1};   // nodes of level L are [lvl_off[L], lvl_off[L+1])
    while (lvl_off.back() < N) {
        // first j whose parent is not in front of the end of the last level: std::lower_bound over parent[1..N)
        const uint32_t lim = lvl_off.back();
        const uint32_t nxt = (uint32_t)(std::lower_bound(parent + 1, parent + N, lim) - parent);
        if (nxt <= lim) break;   // (no progress: the array is not a breadth-first expansion -- the device says so below)
        lvl_off.push_back(nxt);
    }
    if (lvl_off.back() != N) {
        uint32_t bad = 0;
        FS_TRY(hipMemcpyAsync(&bad, P.d_small.p, 4, hipMemcpyDeviceToHost, stream));
        FS_TRY(hipStreamSynchronize(stream));
        if (bad || N > 1) {
            return ugp::set_error(UGP_ERR_INVALID, "tree arrays are not in breadth-first order");
        }
    }
    const uint32_t n_levels = (uint32_t)lvl_off.size() - 1;
    std::vector<uint32_t> ilvl_off(n_levels + 1, 0);
    uint32_t n_wide = 0;
    {
        // (the level tables have a buffer of their own, sized by the tree -- a caterpillar has as many levels as nodes)
        FS_TRY(P.d_levels.alloc(3 * (size_t)(n_levels + 1)));   // (level begins, internal-node ranks, and -- below -- the forward sweep's groups)
        uint32_t *d_lvl = P.d_levels.p, *d_ilvl = P.d_levels.p + (n_levels + 1);
        FS_TRY(hipMemcpyAsync(d_lvl, lvl_off.data(), (n_levels + 1) * 4, hipMemcpyHostToDevice, stream));
        hipLaunchKernelGGL(k_fs_level_ranks, dim3((n_levels + 1 + 63) / 64), dim3(64), 0, stream, P.d_heads.p, P.d_small.p + 1, d_lvl, n_levels, d_ilvl);
        uint32_t bad = 0;
        FS_TRY(hipMemcpyAsync(ilvl_off.data(), d_ilvl, (n_levels + 1) * 4, hipMemcpyDeviceToHost, stream));
        FS_TRY(hipMemcpyAsync(&bad, P.d_small.p, 4, hipMemcpyDeviceToHost, stream));
        FS_TRY(hipMemcpyAsync(&n_wide, P.d_small.p + 3, 4, hipMemcpyDeviceToHost, stream));
        FS_TRY(hipStreamSynchronize(stream));
        if (bad) return ugp::set_error(UGP_ERR_INVALID, "tree arrays are not in breadth-first order");
    }
    // Polytomies (more than FS_WIDE children): sorted by node = by level, their children cut into chunks of FS_CHUNK rows.  One table
    // on the device: [chunk first rows | chunk row counts | nodes | child counts | first chunk (within the level)].
    std::vector<uint32_t> wl_off(n_levels + 1, 0), cl_off(n_levels + 1, 0);   // polytomies / chunks in front of level L
    uint32_t n_chunks = 0, max_level_chunks = 0;
    const uint32_t *d_chunk_first = nullptr, *d_chunk_rows = nullptr, *d_wide_node = nullptr, *d_wide_nc = nullptr, *d_wide_chunk0 = nullptr;
    if (n_wide) {
        if (n_wide > wide_cap) return ugp::set_error(UGP_ERR_INVALID, "tree arrays are not in breadth-first order");
        std::vector<uint32_t> raw(3 * (size_t)n_wide);
        FS_TRY(hipMemcpyAsync(raw.data(), P.d_wide.p, raw.size() * 4, hipMemcpyDeviceToHost, stream));
        FS_TRY(hipStreamSynchronize(stream));
        std::vector<uint32_t> ord(n_wide);
        for (uint32_t i = 0; i < n_wide; i++) ord[i] = i;
        std::sort(ord.begin(), ord.end(), [&](uint32_t a, uint32_t b) { return raw[3 * (size_t)a] < raw[3 * (size_t)b]; });
        for (uint32_t i = 0; i < n_wide; i++) n_chunks += (raw[3 * (size_t)i + 2] + FS_CHUNK - 1) / FS_CHUNK;
        std::vector<uint32_t> tab(2 * (size_t)n_chunks + 3 * (size_t)n_wide);
        uint32_t *c_first = tab.data(), *c_rows = c_first + n_chunks, *w_node = c_rows + n_chunks, *w_nc = w_node + n_wide, *w_c0 = w_nc + n_wide;
        uint32_t ci = 0, L = 0;
        for (uint32_t i = 0; i < n_wide; i++) {
            const uint32_t nd = raw[3 * (size_t)ord[i]], first = raw[3 * (size_t)ord[i] + 1], nc = raw[3 * (size_t)ord[i] + 2];
            while (nd >= lvl_off[L + 1]) { L++; wl_off[L] = i; cl_off[L] = ci; }
            w_node[i] = nd; w_nc[i] = nc; w_c0[i] = ci - cl_off[L];
            for (uint32_t at = 0; at < nc; at += FS_CHUNK) { c_first[ci] = first + at; c_rows[ci] = std::min(FS_CHUNK, nc - at); ci++; }
        }
        while (L < n_levels) { L++; wl_off[L] = n_wide; cl_off[L] = ci; }
        for (uint32_t l = 0; l < n_levels; l++) max_level_chunks = std::max(max_level_chunks, cl_off[l + 1] - cl_off[l]);
        FS_TRY(P.d_wtab.alloc(tab.size()));
        FS_TRY(hipMemcpyAsync(P.d_wtab.p, tab.data(), tab.size() * 4, hipMemcpyHostToDevice, stream));
        FS_TRY(hipStreamSynchronize(stream));   // (`tab` goes out of scope)
        d_chunk_first = P.d_wtab.p; d_chunk_rows = d_chunk_first + n_chunks; d_wide_node = d_chunk_rows + n_chunks;
        d_wide_nc = d_wide_node + n_wide; d_wide_chunk0 = d_wide_nc + n_wide;
    }
    if (S == 0) { guard.r = nullptr; *out = res; return UGP_OK; }
    // the forward sweep's waves take FS_FN internal nodes of a level each: goff[L] = the first such group of level L
    std::vector<uint32_t> goff(n_levels + 1, 0);
    for (uint32_t L = 0; L < n_levels; L++) goff[L + 1] = goff[L] + (ilvl_off[L + 1] - ilvl_off[L] + FS_FN - 1) / FS_FN;
    const uint32_t n_groups = goff[n_levels];
    const uint32_t *d_ilvl = P.d_levels.p + (n_levels + 1), *d_goff = P.d_levels.p + 2 * (size_t)(n_levels + 1);
    FS_TRY(hipMemcpyAsync(P.d_levels.p + 2 * (size_t)(n_levels + 1), goff.data(), (n_levels + 1) * 4, hipMemcpyHostToDevice, stream));
    t_topo = lap();

    // sites per pass: F takes N * W * 4 bytes.  Passes of up to 16 GiB (or half of the free HBM if that is less; UGP_FITCH_BYTES
    // overrides): 10 M nodes x 2 048 sites go in one -- one upload, one sweep of 2 x 46 launches, one sort, one download instead of
    // four of each (18.9 -> 15.0 ms per call, round 6) -- and the buffer is pooled, and mostly untouched: rows that are never
    // stored (below) are never written either.
    size_t free_b = 0, total_b = 0;
    FS_TRY(hipMemGetInfo(&free_b, &total_b));
    uint64_t budget = std::min<uint64_t>((free_b + P.d_F.cap * 4) / 2, 16ull << 30);
    if (const char *e = getenv("UGP_FITCH_BYTES")) budget = strtoull(e, nullptr, 10);
    budget = std::min<uint64_t>(budget, 16ull << 30);   // (k_fs_forward addresses F in 32-bit words)
    uint64_t W_max = std::max<uint64_t>(budget / (((uint64_t)N + 1) * 4), 1);   // (+1: the reference word's own row, behind the nodes')
    if (W_max >= 64) W_max &= ~63ull;   // whole 512-site wave rows
    const uint64_t W_all = (S + 7) / 8;
    const uint32_t W_pass = (uint32_t)std::min<uint64_t>(W_max, W_all);
    auto &d_F = P.d_F; auto &d_refw = P.d_refw; auto &d_vnode = P.d_vnode;
    auto &d_vnuc = P.d_vnuc; auto &d_oval = P.d_oval; auto &d_oval2 = P.d_oval2; auto &d_tmp = P.d_tmp;
    auto &d_okey = P.d_okey; auto &d_okey2 = P.d_okey2; auto &d_voff = P.d_voff;
    auto &d_cnt = P.d_cnt; auto &d_segb = P.d_segb;
    FS_TRY(d_F.alloc(((size_t)N + 1) * W_pass));
    FS_TRY(d_refw.alloc(W_pass));
    FS_TRY(d_cnt.alloc((size_t)FS_SEG * 8 + 8));
    FS_TRY(d_segb.alloc(FS_SEG + 1));
    std::vector<uint32_t> refw(W_pass), vnode;
    std::vector<uint8_t> vnuc;
    std::vector<uint64_t> voff;
    std::vector<unsigned long long> h_cnt((size_t)FS_SEG * 8), h_segb(FS_SEG + 1);
    std::vector<uint32_t> seen_site;   // (slow path) last site, 1-based, a node had a cell at -- and where
    std::vector<uint64_t> seen_at;
    t_alloc = lap();

    for (uint64_t w0 = 0; w0 < W_all; w0 += W_pass) {
        const uint32_t W = (uint32_t)std::min<uint64_t>(W_pass, W_all - w0);
        const uint64_t s0 = w0 * 8, s1 = std::min<uint64_t>(S, s0 + (uint64_t)W * 8);
        const uint32_t n_s = (uint32_t)(s1 - s0);
        std::fill(refw.begin(), refw.end(), 0x11111111u);
        for (uint64_t s = s0; s < s1; s++) {
            const uint32_t sh = (uint32_t)((s - s0) & 7) * 4;
            uint32_t &x = refw[(s - s0) >> 3];
            x = (x & ~(0xFu << sh)) | ((uint32_t)sites->ref[s] << sh);
        }
        // Genotype cells of this pass.  A node named twice at one site keeps the last cell (:47-62 runs in
        // order).  Cells whose node indices ascend within every site cannot repeat a node and are uploaded as
        // they are -- the scatter kernel checks that (and the cells' ranges) as it goes; when it finds them out of
        // order the pass starts over with the cells filtered through a last-seen table on the host.
        const uint64_t v0 = sites->var_off[s0], v1 = sites->var_off[s1];
        const uint32_t *cell_node = sites->var_node + v0;
        const uint8_t *cell_nuc = sites->var_nuc + v0;
        uint64_t n_cells = v1 - v0;
        voff.resize(n_s + 1);
        for (uint32_t k = 0; k <= n_s; k++) voff[k] = sites->var_off[s0 + k] - v0;
        const uint32_t gy = (W + 63) / 64;
        unsigned long long seg_cap = 0;
        const uint64_t npad = ((uint64_t)N + 63) & ~63ull;
        const size_t bm_words = (size_t)(npad / 32 * gy) + 4;   // (tile-major, whole 64-bit words per tile: k_fs_init takes one per wave)
        FS_TRY(P.d_mark8.alloc((size_t)npad * gy));
        FS_TRY(P.d_stored.alloc(bm_words));
        FS_TRY(P.d_desc.alloc((size_t)n_groups * gy));
        if (max_level_chunks) FS_TRY(P.d_part.alloc((size_t)max_level_chunks * gy * 8 * 64));
        auto grid = [&](uint64_t nodes, int per_wave) { return dim3((unsigned)((nodes + 4ull * per_wave - 1) / (4ull * per_wave) * gy)); };
        for (int attempt = 0; attempt < 2; attempt++) {
            FS_TRY(hipMemcpyAsync(d_refw.p, refw.data(), (size_t)W * 4, hipMemcpyHostToDevice, stream));
            FS_TRY(hipMemcpyAsync(d_F.p + (size_t)N * W, refw.data(), (size_t)W * 4, hipMemcpyHostToDevice, stream));   // (row N: what a missing row reads as)
            FS_TRY(d_voff.upload(voff.data(), voff.size(), stream));
            FS_TRY(d_vnode.alloc(n_cells)); FS_TRY(d_vnuc.alloc(n_cells));
            FS_TRY(hipMemsetAsync(P.d_mark8.p, 0, (size_t)npad * gy, stream));
            FS_TRY(hipMemsetAsync(P.d_small.p + 2, 0, 4, stream));
            // (round 6) Only the (node, tile) items that hold a cell get a row: mark them, initialise those, then drop the cells in.  The
            // cells are grouped by site, i.e. by tile: they go up in pieces of whole tiles on a stream of their own, and the kernels of a
            // piece run while the next piece crosses PCIe (68 MB from pageable memory are 1.3 ms, the kernels 1.7).
            uint32_t n_piece = (n_cells >= (1u << 20) && gy >= 2) ? std::min<uint32_t>(gy, FS_PIECES) : 1u;
            if (const char *e = getenv("UGP_FITCH_PIECES")) n_piece = (uint32_t)std::max(1, std::min<int>(atoi(e), (int)std::min<uint32_t>(gy, FS_PIECES)));   // (tests: small inputs in pieces)
            if (n_piece > 1 && !P.copy) {
                FS_TRY(hipStreamCreateWithFlags(&P.copy, hipStreamNonBlocking));
                for (int k = 0; k < FS_PIECES; k++) FS_TRY(hipEventCreateWithFlags(&P.up[k], hipEventDisableTiming));
            }
            for (uint32_t k = 0; k < n_piece; k++) {
                const uint32_t y0 = (uint32_t)((uint64_t)gy * k / n_piece), y1 = (uint32_t)((uint64_t)gy * (k + 1) / n_piece), ny = y1 - y0;
                const uint64_t c0 = voff[std::min<uint64_t>((uint64_t)y0 * 512, n_s)], c1 = voff[std::min<uint64_t>((uint64_t)y1 * 512, n_s)];
                if (c1 > c0) {
                    hipStream_t up_on = n_piece > 1 ? P.copy : stream;
                    FS_TRY(hipMemcpyAsync(d_vnode.p + c0, cell_node + c0, (c1 - c0) * 4, hipMemcpyHostToDevice, up_on));
                    FS_TRY(hipMemcpyAsync(d_vnuc.p + c0, cell_nuc + c0, c1 - c0, hipMemcpyHostToDevice, up_on));
                    if (n_piece > 1) {
                        FS_TRY(hipEventRecord(P.up[k], P.copy));
                        FS_TRY(hipStreamWaitEvent(stream, P.up[k], 0));
                    }
                    hipLaunchKernelGGL(k_fs_mark, dim3((unsigned)((c1 - c0 + 255) / 256)), dim3(256), 0, stream, d_voff.p, n_s, d_vnode.p, c0, c1, npad, N,
                                       P.d_mark8.p);
                }
                hipLaunchKernelGGL(k_fs_init, dim3((unsigned)(((uint64_t)N + 255) / 256 * ny)), dim3(256), 0, stream, d_F.p, d_refw.p, d_nchild.p, N, W, y0, ny,
                                   npad, P.d_mark8.p, P.d_stored.p);
                if (n_groups)
                    hipLaunchKernelGGL(k_fs_desc, dim3((unsigned)(((uint64_t)n_groups * ny * FS_FN + 255) / 256)), dim3(256), 0, stream, d_inodes.p, d_ilvl,
                                       d_goff, n_levels, n_groups, y0, ny, npad, P.d_mark8.p, P.d_desc.p);
                if (c1 > c0)
                    hipLaunchKernelGGL(k_fs_scatter, dim3((unsigned)((c1 - c0 + 255) / 256)), dim3(256), 0, stream, d_F.p, d_refw.p, d_nchild.p, d_voff.p, n_s,
                                       d_vnode.p, d_vnuc.p, c0, c1, W, N, P.d_small.p + 2);
            }
            FS_TRY(hipMemsetAsync(d_cnt.p, 0, ((size_t)FS_SEG * 8 + 8) * sizeof(unsigned long long), stream));
            // room for the listed changes: without cells on internal nodes a site has at most (cells + 1) of
            // them, and usually far fewer; anything beyond the guess is handled by the exact pass below
            seg_cap = (n_cells + n_s) / FS_SEG * 3 / 2 + 256;
            if (const char *e = getenv("UGP_FITCH_EMIT_CAP")) seg_cap = strtoull(e, nullptr, 10);
            FS_TRY(d_okey.alloc(seg_cap * FS_SEG));
            FS_TRY(d_oval.alloc(seg_cap * FS_SEG));
            uint32_t cell_flags = 0;
            if (n_cells) {
                FS_TRY(hipMemcpyAsync(&cell_flags, P.d_small.p + 2, 4, hipMemcpyDeviceToHost, stream));
                FS_TRY(hipStreamSynchronize(stream));   // (the upload buffers of the caller may be reused / the verdict is needed)
            }
            if (cell_flags & 1u) return ugp::set_error(UGP_ERR_INVALID, "variant node index out of range");
            if (cell_flags & 2u) return ugp::set_error(UGP_ERR_INVALID, "variant allele mask must be 1..15");
            if (!(cell_flags & 4u) || attempt == 1) break;
            // (rare) some site names a node twice or out of order: keep the last cell of every (site, node), in order of first appearance
            if (seen_site.empty()) { seen_site.assign(N, 0); seen_at.assign(N, 0); }
            vnode.clear(); vnuc.clear();
            for (uint64_t s = s0; s < s1; s++) {
                voff[s - s0] = vnode.size();
                for (uint64_t v = sites->var_off[s]; v < sites->var_off[s + 1]; v++) {
                    const uint32_t nd = sites->var_node[v];
                    if (seen_site[nd] == (uint32_t)(s + 1)) { vnuc[seen_at[nd]] = sites->var_nuc[v]; continue; }
                    seen_site[nd] = (uint32_t)(s + 1);
                    seen_at[nd] = vnode.size();
                    vnode.push_back(nd); vnuc.push_back(sites->var_nuc[v]);
                }
            }
            voff[n_s] = vnode.size();
            // (the filtered list may still not ascend: that is fine -- no node is named twice any more; the flag is ignored on the second go)
            cell_node = vnode.data(); cell_nuc = vnuc.data(); n_cells = vnode.size();
        }
        t_prep += lap();

        for (uint32_t L = n_levels; L-- > 0;) {
            const uint32_t cnt = ilvl_off[L + 1] - ilvl_off[L];
            if (cnt)
                hipLaunchKernelGGL(k_fs_forward, grid(cnt, FS_FN), dim3(256), 0, stream, d_F.p, d_inodes.p + ilvl_off[L], P.d_ifirst.p + ilvl_off[L],
                                   P.d_inch.p + ilvl_off[L], P.d_desc.p + goff[L], n_groups, cnt, W, gy, npad, P.d_stored.p, N);
            if (const uint32_t nw = wl_off[L + 1] - wl_off[L]) {   // the level's polytomies: chunks counted in parallel, then summed
                const uint32_t nc = cl_off[L + 1] - cl_off[L];
                hipLaunchKernelGGL(k_fs_wide_count, grid(nc, 1), dim3(256), 0, stream, d_F.p, d_chunk_first + cl_off[L], d_chunk_rows + cl_off[L], nc, W, gy,
                                   npad, P.d_stored.p, N, P.d_part.p);
                hipLaunchKernelGGL(k_fs_wide_final, grid(nw, 1), dim3(256), 0, stream, d_F.p, d_wide_node + wl_off[L], d_wide_nc + wl_off[L],
                                   d_wide_chunk0 + wl_off[L], nw, W, gy, npad, P.d_mark8.p, P.d_part.p);
            }
        }
        for (uint32_t L = 0; L < n_levels; L++) {
            const uint32_t cnt = lvl_off[L + 1] - lvl_off[L];
            hipLaunchKernelGGL(k_fs_backward, grid(cnt, FS_NB), dim3(256), 0, stream, d_F.p, d_refw.p, d_parent.p, d_nchild.p, lvl_off[L], lvl_off[L + 1],
                               W, gy, npad, (uint32_t)s0, d_cnt.p, seg_cap, d_okey.p, d_oval.p, P.d_stored.p);
        }
        FS_TRY(hipGetLastError());
        FS_TRY(hipMemcpyAsync(h_cnt.data(), d_cnt.p, h_cnt.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost, stream));
        FS_TRY(hipStreamSynchronize(stream));
        unsigned long long n_mut = 0;
        bool overflow = false;
        for (uint32_t g = 0; g < FS_SEG; g++) {
            h_segb[g] = n_mut;
            n_mut += h_cnt[(size_t)g * 8];
            overflow = overflow || h_cnt[(size_t)g * 8] > seg_cap;
        }
        h_segb[FS_SEG] = n_mut;
        t_kern += lap();
        if (n_mut == 0) continue;
        FS_TRY(d_okey2.alloc(n_mut));
        FS_TRY(d_oval2.alloc(n_mut));
        if (overflow) {   // the guess was too small for some segment: list again, exactly
            hipLaunchKernelGGL(k_fs_emit, grid(N, 1), dim3(256), 0, stream, d_F.p, d_refw.p, d_parent.p, d_nchild.p, N, W, gy, npad, (uint32_t)s0,
                               d_cnt.p + (size_t)FS_SEG * 8, n_mut, d_okey2.p, d_oval2.p, P.d_stored.p);
        } else {
            FS_TRY(hipMemcpyAsync(d_segb.p, h_segb.data(), h_segb.size() * sizeof(unsigned long long), hipMemcpyHostToDevice, stream));
            hipLaunchKernelGGL(k_fs_compact, dim3(FS_SEG), dim3(256), 0, stream, d_okey.p, d_oval.p, d_segb.p, seg_cap, d_okey2.p, d_oval2.p);
        }
        FS_TRY(hipGetLastError());
        FS_TRY(hipStreamSynchronize(stream));   // (d_okey / d_oval are about to become the sort's output)
        FS_TRY(d_okey.alloc(n_mut));
        FS_TRY(d_oval.alloc(n_mut));
        // deterministic order: by site, then breadth-first node index
        // (key = site << 32 | node: a stable sort by the node's bits, then by the site's -- the radix passes over the empty bits between
        // the two fields and above the site are not run: 35 of 64 bits at 10 M nodes x 2 048 sites)
        auto bits_of = [](uint64_t n) { unsigned b = 1; while (b < 32 && (n >> b)) b++; return b; };
        const unsigned nb_bits = bits_of(N), sb_bits = bits_of(s1 > 0 ? s1 - 1 : 0);
        size_t tmp_bytes = 0, tmp2 = 0;
        FS_TRY(rocprim::radix_sort_pairs(nullptr, tmp_bytes, d_okey2.p, d_okey.p, d_oval2.p, d_oval.p, (size_t)n_mut, 0u, nb_bits, stream));
        FS_TRY(rocprim::radix_sort_pairs(nullptr, tmp2, d_okey.p, d_okey2.p, d_oval.p, d_oval2.p, (size_t)n_mut, 32u, 32u + sb_bits, stream));
        tmp_bytes = std::max(tmp_bytes, tmp2);
        FS_TRY(d_tmp.alloc(tmp_bytes));
        FS_TRY(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_okey2.p, d_okey.p, d_oval2.p, d_oval.p, (size_t)n_mut, 0u, nb_bits, stream));
        FS_TRY(rocprim::radix_sort_pairs(d_tmp.p, tmp_bytes, d_okey.p, d_okey2.p, d_oval.p, d_oval2.p, (size_t)n_mut, 32u, 32u + sb_bits, stream));
        // the four arrays of the result are made on the device and land in the handle's vectors directly
        FS_TRY(P.d_osite.alloc(n_mut)); FS_TRY(P.d_onode.alloc(n_mut)); FS_TRY(P.d_opar.alloc(n_mut)); FS_TRY(P.d_onuc.alloc(n_mut));
        hipLaunchKernelGGL(k_fs_split, dim3((unsigned)((n_mut + 255) / 256)), dim3(256), 0, stream, d_okey2.p, d_oval2.p, (uint64_t)n_mut, P.d_osite.p,
                           P.d_onode.p, P.d_opar.p, P.d_onuc.p);
        const size_t base = res->site.size();
        res->site.resize(base + n_mut); res->node.resize(base + n_mut); res->par.resize(base + n_mut); res->nuc.resize(base + n_mut);
        FS_TRY(hipMemcpyAsync(res->site.data() + base, P.d_osite.p, n_mut * 4, hipMemcpyDeviceToHost, stream));
        FS_TRY(hipMemcpyAsync(res->node.data() + base, P.d_onode.p, n_mut * 4, hipMemcpyDeviceToHost, stream));
        FS_TRY(hipMemcpyAsync(res->par.data() + base, P.d_opar.p, n_mut, hipMemcpyDeviceToHost, stream));
        FS_TRY(hipMemcpyAsync(res->nuc.data() + base, P.d_onuc.p, n_mut, hipMemcpyDeviceToHost, stream));
        FS_TRY(hipStreamSynchronize(stream));
        t_out += lap();
    }
    if (getenv("UGP_FITCH_VERBOSE"))
        fprintf(stderr, "[ugp_fitch] topology %.1f ms, alloc %.1f ms, per-pass prep %.1f ms, kernels %.1f ms, sort+download %.1f ms\n",
                t_topo, t_alloc, t_prep, t_kern, t_out);
    guard.r = nullptr;
    *out = res;
    return UGP_OK;
}

extern "C" uint64_t ugp_fitch_count(const ugp_fitch *f) { return f ? f->site.size() : 0; }

extern "C" int ugp_fitch_get(const ugp_fitch *f, uint32_t *site, uint32_t *node, uint8_t *par_nuc, uint8_t *mut_nuc) {
    if (!f) return ugp::set_error(UGP_ERR_INVALID, "null handle");
    const size_t n = f->site.size();
    if (n && (!site || !node || !par_nuc || !mut_nuc)) return ugp::set_error(UGP_ERR_INVALID, "null output arrays");
    if (n) {
        memcpy(site, f->site.data(), n * 4);
        memcpy(node, f->node.data(), n * 4);
        memcpy(par_nuc, f->par.data(), n);
        memcpy(mut_nuc, f->nuc.data(), n);
    }
    return UGP_OK;
}

extern "C" void ugp_fitch_destroy(ugp_fitch *f) { delete f; }
