// ugp_kernels.hip -- gfx950 kernels of the placement hot path.
//
// What is computed (DESIGN.md "What the kernel computes"; reference:
// usher_mapper.cpp:167-504 as driven by usher_common.cpp:389-449): for every
// node n of the tree and every query sample s,
//     D(n,s)    = D(parent,s) + sum_m ([prev(m) in S_s(pos m)] - [mut(m) in S_s(pos m)])
//     cost(n,s) = D(parent,s) + sum_m min(delta, 0)         (cost(root) = D(root))
// plus the eligibility predicate, and per sample the reduction
//     (min cost, #ties, argmax (n_leaves, bfs_j) among ties).
//
// Mapping to the hardware.  Lanes = samples: a 64-lane wavefront owns a tile of
// query samples and walks a contiguous range of the tree's DFS record stream.
// The stream is wave-uniform: 64 dwords at a time are loaded coalesced into one
// VGPR and handed to the scalar unit with v_readlane, so record decode, slot
// numbers and loop control run on SGPRs.  Per-lane memory traffic is one
// coalesced row of the tile's 4-bit allele table per tree mutation and the D
// stack in LDS (depth <= log2 N by construction, see ugp_flatten.cpp).
// Integer work only; no MFMA.
//
//   k_best8   phase 1, the dominant kernel: 8 samples per lane (512 per wave),
//             nibble-parallel membership tests on the lane's table dword, 4-bit
//             SWAR accumulators per node, packed 16-bit D / cost / running
//             minimum, software-pipelined row prefetch.  Produces the minimum
//             cost per (chunk, sample).
//   k_gbest / k_select / k_ties / k_final   phase 2: global minimum per sample,
//             then only the (chunk, 64-sample tile) pairs that attain it are
//             re-walked (one sample per lane, 32-bit) to count ties and pick the
//             reference's winner.
//   k_place   one sample per lane, 32-bit: per-node scores (-p), tie lists, and
//             the general fallback when 16-bit counters could overflow.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <stdint.h>

#include <algorithm>
#include <type_traits>

#include "ugp_bound3.hpp"
#include "ugp_kernels.hpp"
#include "ugp_update.hpp"

namespace ugp {

typedef unsigned short us2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t u32xhot __attribute__((ext_vector_type(UGP_HOT_SLOTS)));   // one element per hot slot
constexpr uint32_t HOT_MASK = UGP_HOT_SLOTS - 1u;
static_assert(UGP_HOT_SLOTS == 8 || UGP_HOT_SLOTS == 16, "hot slots: 8 or 16");

__device__ __forceinline__ uint32_t pk_add(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) + __builtin_bit_cast(us2, b));
}
__device__ __forceinline__ uint32_t pk_sub(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_bit_cast(us2, a) - __builtin_bit_cast(us2, b));
}
__device__ __forceinline__ uint32_t pk_min(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t pk_max(uint32_t a, uint32_t b) {
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
}
__device__ __forceinline__ uint32_t rdlane(uint32_t v, uint32_t l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l);
}

// ----------------------------------------------------------- allele tiles

// table[tile512][4 + site][64 dwords]: one 256-byte row per (512-sample tile, site), behind 4 constant rows;
// dword l, nibble j = allele set of sample tile*512 + 8*l + j at that site.
// Initialised to the reference base everywhere (a sample without a VCF row at a
// position carries the reference allele, usher_mapper.cpp:244, 301, 425).
// Rows 0..3 of every tile are constants ("every sample carries A / C / G / T"): k_best8 fetches the constant
// row of a site's reference base instead of the site's own row wherever no sample of the tile differs from it.
__global__ void k_fill_table(uint32_t *__restrict__ table, const uint8_t *__restrict__ site_ref,
                             uint32_t n_rows, uint64_t total_dwords) {
    // (16 bytes per thread and store: a row is 64 dwords = 16 such pieces, so a wave writes four rows per instruction)
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x, total4 = total_dwords >> 2;
    uint4 *t4 = (uint4 *)table;
    for (; i < total4; i += stride) {
        const uint32_t row = (uint32_t)((i >> 4) % n_rows);
        const uint32_t v = row < TABLE_CONST_ROWS ? (0x11111111u << row) : (uint32_t)site_ref[row - TABLE_CONST_ROWS] * 0x11111111u;
        t4[i] = make_uint4(v, v, v, v);
    }
}

// One thread per VCF row: overwrite the sample's nibble at tree sites and count
// D_bottom = #{non-missing rows whose allele set excludes the reference base}
// (usher_mapper.cpp:292-388 with an empty ancestral list).
__global__ void k_scatter_entries(uint32_t *__restrict__ table, uint32_t *__restrict__ dbottom,
                                  const int32_t *__restrict__ pos, const uint8_t *__restrict__ ref,
                                  const uint8_t *__restrict__ nuc, const uint8_t *__restrict__ is_missing,
                                  const uint32_t *__restrict__ ent_q, const int32_t *__restrict__ pos2site,
                                  uint32_t max_pos, uint32_t n_sites, uint64_t n_ent, uint32_t q_base,
                                  uint32_t *__restrict__ active, uint32_t active_words,
                                  const uint32_t *__restrict__ slot_of,
                                  const uint32_t *__restrict__ row_list, const uint32_t *__restrict__ n_listed, uint32_t n_q,
                                  const unsigned long long *__restrict__ err, uint32_t *__restrict__ useful, uint32_t useful_words) {
    // A batch whose rows failed k_rows_prepare's checks (its verdict is read by the host only after the whole pipeline has been
    // queued: ugp_place_batch_async) is placed as if it had no rows at all: the table stays "reference everywhere", D(bottom) 0,
    // so that nothing downstream ever sees duplicate or unsorted rows.  The caller gets the error, never these results.
    if (err && *err != ~0ull) return;
    // (with a row list -- the rows that are not missing, k_nmask_build -- the kernel strides over it: its length is only known
    // on the device; rows of samples outside [q_base, q_base + n_q) belong to another sub-batch)
    const uint64_t n_items = row_list ? (uint64_t)*n_listed : n_ent;
    for (uint64_t it = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; __builtin_amdgcn_ballot_w64(it < n_items) != 0; it += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t e = (it < n_items) ? (row_list ? (uint64_t)row_list[it] : it) : 0;
    bool valid = it < n_items;
    uint32_t q = 0, r = 0, a = 0, miss = 1;
    int32_t p = -1;
    if (valid && row_list) { const uint32_t qa = ent_q[e]; valid = qa >= q_base && qa - q_base < n_q; }
    if (valid) {
        q = ent_q[e] - q_base;                  // sample index within this launch
        if (slot_of) q = slot_of[q];            // ... and its place in the locality-sorted tiles
        p = pos[e];
        r = ref[e];
        miss = is_missing[e];
        a = miss ? 15u : (uint32_t)nuc[e];
    }
    // D_bottom.  The rows of a sample are consecutive, so a wave usually serves one sample: one atomic per wave
    // instead of one per row (same-address atomics of a wave are executed one after another).
    const bool mism = valid && !miss && (a & r) == 0;
    const unsigned long long mm = __builtin_amdgcn_ballot_w64(mism);
    const uint32_t q0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q);
    if (__builtin_amdgcn_ballot_w64(valid && q != q0) == 0) {
        if (mm && (threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(mm)) atomicAdd(&dbottom[q0], (uint32_t)__builtin_popcountll(mm));
    } else if (mism) atomicAdd(&dbottom[q], 1u);
    if (!valid || p < 0 || (uint32_t)p > max_pos) continue;
    const int32_t site = pos2site[p];
    if (site < 0) continue;
    const uint32_t tile = q >> 9, within = q & 511;
    uint32_t *w = table + ((uint64_t)tile * (n_sites + TABLE_CONST_ROWS) + TABLE_CONST_ROWS + (uint32_t)site) * 64 + (within >> 3);
    const uint32_t sh = (within & 7) * 4;
    atomicXor(w, ((r ^ a) & 15u) << sh);   // nibble was r (k_fill_table); rows are unique per (sample, position)
    // the row of (tile, site) is no longer "reference everywhere" (look before setting: after the first few samples of a
    // tile most bits are set, and a run of N cells hits the same word with every lane)
    if (r != a) {
        uint32_t *aw = &active[(uint64_t)tile * active_words + ((uint32_t)site >> 5)];
        const uint32_t bit = 1u << ((uint32_t)site & 31u);
        if (!(__hip_atomic_load(aw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bit)) atomicOr(aw, bit);
    }
    // third pruning bound: the alleles of a set that excludes the reference base are "useful" for this tile (a mutation to one of them
    // can match this sample's variant)
    if (useful && (a & r) == 0) {
        uint32_t *uw = &useful[(uint64_t)tile * useful_words + ((uint32_t)site >> 3)];
        const uint32_t bits = (a & 15u) << (((uint32_t)site & 7u) * 4u);
        if ((__hip_atomic_load(uw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(uw, bits);
    }
    if (!row_list) break;   // (one row per thread)
    }
}

// The same for the plain case (every row of the launch, no row list), K rows per thread: a row is a chain of five dependent
// accesses (sample -> slot, position -> site, the table word, the active-row word), and next to the persistent walks of the
// batches in front a CU has room for about one more wave -- with one row per thread the launch was 5,600 waves taking turns
// (64 us in the pipelined trace, twice per batch); four independent chains per thread need a quarter of the waves for the same
// latency each.
template <int K>
__global__ void __launch_bounds__(64) k_scatter_rows(uint32_t *__restrict__ table, uint32_t *__restrict__ dbottom,
                                                     const int32_t *__restrict__ pos, const uint8_t *__restrict__ ref,
                                                     const uint8_t *__restrict__ nuc, const uint8_t *__restrict__ is_missing,
                                                     const uint32_t *__restrict__ ent_q, const int32_t *__restrict__ pos2site,
                                                     uint32_t max_pos, uint32_t n_sites, uint64_t n_ent, uint32_t q_base,
                                                     uint32_t *__restrict__ active, uint32_t active_words, const uint32_t *__restrict__ slot_of,
                                                     const unsigned long long *__restrict__ err, uint32_t *__restrict__ useful, uint32_t useful_words) {
    if (err && *err != ~0ull) return;   // (as k_scatter_entries: a batch with bad rows is built as if it had none)
    const uint64_t base = (uint64_t)blockIdx.x * 64u * K + threadIdx.x;
    bool valid[K];
    uint32_t q[K], r[K], a[K], miss[K];
    int32_t p[K], site[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        const uint64_t e = base + (uint64_t)j * 64u;
        valid[j] = e < n_ent;
        q[j] = valid[j] ? ent_q[e] - q_base : 0u;
        p[j] = valid[j] ? pos[e] : -1;
        r[j] = valid[j] ? ref[e] : 0u;
        miss[j] = valid[j] ? is_missing[e] : 1u;
        a[j] = valid[j] ? (uint32_t)nuc[e] : 0u;
    }
#pragma unroll
    for (int j = 0; j < K; j++) {
        if (valid[j] && slot_of) q[j] = slot_of[q[j]];
        if (miss[j]) a[j] = 15u;
        site[j] = (valid[j] && p[j] >= 0 && (uint32_t)p[j] <= max_pos) ? pos2site[p[j]] : -1;
    }
#pragma unroll
    for (int j = 0; j < K; j++) {
        // D_bottom: the rows of a sample are consecutive, so a wave usually serves one sample -- one atomic per wave
        const bool mism = valid[j] && !miss[j] && (a[j] & r[j]) == 0;
        const unsigned long long mm = __builtin_amdgcn_ballot_w64(mism);
        const uint32_t q0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)q[j]);
        if (__builtin_amdgcn_ballot_w64(valid[j] && q[j] != q0) == 0) {
            if (mm && threadIdx.x == (uint32_t)__builtin_ctzll(mm)) atomicAdd(&dbottom[q0], (uint32_t)__builtin_popcountll(mm));
        } else if (mism) atomicAdd(&dbottom[q[j]], 1u);
    }
    uint32_t *aw[K];
    uint32_t have[K];
#pragma unroll
    for (int j = 0; j < K; j++) {
        aw[j] = nullptr; have[j] = 0;
        if (site[j] < 0) continue;
        const uint32_t tile = q[j] >> 9, within = q[j] & 511u;
        uint32_t *w = table + ((uint64_t)tile * (n_sites + TABLE_CONST_ROWS) + TABLE_CONST_ROWS + (uint32_t)site[j]) * 64 + (within >> 3);
        atomicXor(w, ((r[j] ^ a[j]) & 15u) << ((within & 7u) * 4u));   // nibble was r (k_fill_table); rows are unique per (sample, position)
        if (r[j] != a[j]) {
            aw[j] = &active[(uint64_t)tile * active_words + ((uint32_t)site[j] >> 5)];
            have[j] = __hip_atomic_load(aw[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (look before setting)
        }
    }
#pragma unroll
    for (int j = 0; j < K; j++) {
        if (!aw[j]) continue;
        const uint32_t bit = 1u << ((uint32_t)site[j] & 31u);
        if (!(have[j] & bit)) atomicOr(aw[j], bit);
    }
    if (useful) {   // third pruning bound: see k_scatter_entries
#pragma unroll
        for (int j = 0; j < K; j++) {
            if (site[j] < 0 || (a[j] & r[j]) != 0) continue;
            uint32_t *uw = &useful[(uint64_t)(q[j] >> 9) * useful_words + ((uint32_t)site[j] >> 3)];
            const uint32_t bits = (a[j] & 15u) << (((uint32_t)site[j] & 7u) * 4u);
            if ((__hip_atomic_load(uw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) atomicOr(uw, bits);
        }
    }
}

// ---- tiles of batches with many MISSING rows (N cells come in runs of hundreds to thousands per sample) ---------------------
// One atomic per row into a 200 MB table (k_scatter_entries) or a binary search and a serial row loop per (sample, site block)
// (k_build_tiles) cost 1.4-1.7 ms per build for 41 M rows, twice per call.  The N cells are turned into one bit per
// (sample, site) when the query set arrives -- nmask[sample][word], k_nmask_build, one workgroup per sample, LDS atomics, rows
// read once and coalesced -- and every tile build is then a bit-matrix transpose of those masks into nibbles (k_ntiles: reads
// 3 KB per sample, writes the table once, coalesced, no atomics); the few rows that are not missing go through the scatter
// kernel by way of a list.
__global__ void __launch_bounds__(256) k_nmask_build(const uint64_t *__restrict__ ent_off, const int32_t *__restrict__ pos, const uint8_t *__restrict__ is_missing,
                                                     const int32_t *__restrict__ pos2site, uint32_t max_pos, uint32_t words, uint32_t *__restrict__ nmask,
                                                     uint32_t *__restrict__ plain_rows, uint32_t *__restrict__ n_plain) {
    extern __shared__ uint32_t nm_lds[];   // [words]
    const uint32_t q = blockIdx.x, tid = threadIdx.x;
    for (uint32_t i = tid; i < words; i += 256) nm_lds[i] = 0;
    __syncthreads();
    const uint64_t rb = ent_off[q], re = ent_off[q + 1];
    for (uint64_t r0 = rb; r0 < re; r0 += 256) {
        const uint64_t r = r0 + tid;
        const bool in = r < re;
        const bool miss = in && is_missing[r] != 0;
        if (miss) {
            const int32_t p = pos[r];
            if (p >= 0 && (uint32_t)p <= max_pos) {
                const int32_t site = pos2site[p];
                if (site >= 0) atomicOr(&nm_lds[(uint32_t)site >> 5], 1u << ((uint32_t)site & 31u));
            }
        }
        if (plain_rows) {   // the other rows, in row order within a wave
            const bool plain = in && !miss;
            const unsigned long long pm = __builtin_amdgcn_ballot_w64(plain);
            if (pm) {
                uint32_t base = 0;
                if ((tid & 63u) == (uint32_t)__builtin_ctzll(pm)) base = atomicAdd(n_plain, (uint32_t)__builtin_popcountll(pm));
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, __builtin_ctzll(pm));
                if (plain) plain_rows[base + (uint32_t)__builtin_popcountll(pm & ((1ull << (tid & 63u)) - 1ull))] = (uint32_t)r;
            }
        }
    }
    __syncthreads();
    for (uint32_t i = tid; i < words; i += 256) nmask[(uint64_t)q * words + i] = nm_lds[i];
}

// Workgroup = (16 mask words = 512 sites, tile): the words of the tile's 512 samples staged in LDS, then thread (column l,
// word group g) builds dword l of the rows of its sites -- nibble j = 0xF where sample 8l + j is N, else the reference base --
// and a wave stores one 256-byte row per instruction.  Also writes the tile's constant rows and the active-row bits of its
// sites (plain stores: the scatter of the remaining rows ORs its own in afterwards).
__global__ void __launch_bounds__(256) k_ntiles(uint32_t *__restrict__ table, uint32_t *__restrict__ active, uint32_t active_words,
                                                const uint32_t *__restrict__ nmask, uint32_t words, const uint32_t *__restrict__ order, uint32_t q0,
                                                uint32_t nq, const uint8_t *__restrict__ site_ref, uint32_t n_sites) {
    __shared__ uint32_t m[16][512 + 8];
    const uint32_t tile = blockIdx.y, wb = blockIdx.x * 16, tid = threadIdx.x;
    for (uint32_t i = tid; i < 512 * 16; i += 256) {
        const uint32_t slot = tile * 512 + (i >> 4), w = i & 15u;
        uint32_t v = 0;
        if (slot < nq && wb + w < words) v = nmask[(uint64_t)(q0 + (order ? order[slot] : slot)) * words + wb + w];
        m[w][i >> 4] = v;
    }
    __syncthreads();
    const uint64_t n_rows = (uint64_t)n_sites + TABLE_CONST_ROWS;
    const uint32_t l = tid & 63u, g = tid >> 6;
    for (uint32_t w = g * 4; w < g * 4 + 4; w++) {
        uint32_t mw[8];
#pragma unroll
        for (int j = 0; j < 8; j++) mw[j] = m[w][8 * l + j];
        uint32_t act = 0;
        const uint32_t s0 = (wb + w) * 32;
        for (uint32_t k = 0; k < 32 && s0 + k < n_sites; k++) {
            uint32_t b = 0;
#pragma unroll
            for (int j = 0; j < 8; j++) b |= ((mw[j] >> k) & 1u) << j;
            uint32_t t = (b | (b << 12)) & 0x000F000Fu;   // bit j of b -> bit 4j
            t = (t | (t << 6)) & 0x03030303u;
            t = (t | (t << 3)) & 0x11111111u;
            table[((uint64_t)tile * n_rows + TABLE_CONST_ROWS + s0 + k) * 64 + l] = (uint32_t)site_ref[s0 + k] * 0x11111111u | (t * 15u);
            if (__builtin_amdgcn_ballot_w64(b != 0) != 0) act |= 1u << k;
        }
        if (l == 0 && wb + w < active_words) active[(uint64_t)tile * active_words + wb + w] = act;
    }
    if (blockIdx.x == 0) table[(uint64_t)tile * n_rows * 64 + tid] = 0x11111111u << (tid >> 6);   // the four constant rows
}

// ---- tile build for batches with many rows per sample (high-ambiguity queries: thousands of N cells each) ------------------
// k_scatter_entries does one global atomic per row, each on a different 64-byte line of a 200 MB table: 41 M rows took
// 1.8 ms, twice per step.  Here a workgroup owns (tile, block of TB_SITES sites): it keeps those rows of the table in LDS,
// every thread finds its samples' rows that fall into the block's position range (the rows of a sample are sorted by
// position: one binary search), applies them with LDS atomics, and the block is written out coalesced -- which also
// replaces k_fill_table.  D(bottom) comes from k_row_counts, one wave per sample.
constexpr uint32_t TB_SITES = 128;
__global__ void __launch_bounds__(256) k_build_tiles(uint32_t *__restrict__ table, uint32_t *__restrict__ active, uint32_t active_words,
                                                     const uint64_t *__restrict__ ent_off, uint32_t q0, const uint32_t *__restrict__ order, uint32_t nq,
                                                     const int32_t *__restrict__ pos, const uint8_t *__restrict__ ref, const uint8_t *__restrict__ nuc,
                                                     const uint8_t *__restrict__ is_missing, const int32_t *__restrict__ pos2site,
                                                     const int32_t *__restrict__ site_pos, const uint8_t *__restrict__ site_ref, uint32_t n_sites,
                                                     uint32_t max_pos, const unsigned long long *__restrict__ err) {
    __shared__ uint32_t rows[TB_SITES * 64];
    const bool bad_rows = err && *err != ~0ull;   // (see k_scatter_entries: such a batch is built as if it had no rows)
    __shared__ uint32_t act[TB_SITES];
    const uint32_t tile = blockIdx.y, s0 = blockIdx.x * TB_SITES, s1 = min(s0 + TB_SITES, n_sites), tid = threadIdx.x;
    for (uint32_t i = tid; i < TB_SITES * 64; i += 256) {
        const uint32_t s = s0 + (i >> 6);
        rows[i] = s < n_sites ? (uint32_t)site_ref[s] * 0x11111111u : 0u;
    }
    if (tid < TB_SITES) act[tid] = 0;
    __syncthreads();
    const int32_t P0 = site_pos[s0];                                     // positions of the block's sites: [P0, P1)
    const int64_t P1 = s1 < n_sites ? (int64_t)site_pos[s1] : (int64_t)max_pos + 1;
    for (uint32_t i = tid; i < 512; i += 256) {
        const uint32_t slot = tile * 512 + i;
        if (slot >= nq || bad_rows) continue;
        const uint32_t q = q0 + (order ? order[slot] : slot);
        uint64_t lo = ent_off[q], hi = ent_off[q + 1];
        const uint64_t re = hi;
        while (lo < hi) { const uint64_t mid = (lo + hi) >> 1; if (pos[mid] < P0) lo = mid + 1; else hi = mid; }   // first row at or behind P0
        const uint32_t sh = (i & 7u) * 4u, col = i >> 3;
        for (uint64_t r = lo; r < re; r++) {
            const int32_t p = pos[r];
            if ((int64_t)p >= P1) break;
            const int32_t site = pos2site[p];
            if (site < 0) continue;
            const uint32_t rr = ref[r], a = is_missing[r] ? 15u : (uint32_t)nuc[r];
            atomicXor(&rows[((uint32_t)site - s0) * 64 + col], ((rr ^ a) & 15u) << sh);   // nibble was the reference base; rows are unique per (sample, position)
            if (rr != a) act[(uint32_t)site - s0] = 1;
        }
    }
    __syncthreads();
    const uint64_t n_rows = (uint64_t)n_sites + TABLE_CONST_ROWS;
    uint32_t *out = table + ((uint64_t)tile * n_rows + TABLE_CONST_ROWS + s0) * 64;
    for (uint32_t i = tid; i < (s1 - s0) * 64; i += 256) out[i] = rows[i];
    if (blockIdx.x == 0)   // the four constant rows of the tile
        table[(uint64_t)tile * n_rows * 64 + tid] = 0x11111111u << (tid >> 6);
    if (tid < TB_SITES / 32) {   // "some sample of the tile is not reference here" bits of 32 sites
        uint32_t bits = 0;
        for (uint32_t k = 0; k < 32; k++) bits |= (act[tid * 32 + k] ? 1u : 0u) << k;
        if ((s0 >> 5) + tid < active_words) active[(uint64_t)tile * active_words + (s0 >> 5) + tid] = bits;
    }
}

// D(bottom) = #{non-missing rows whose allele set excludes the reference base}, one wave per sample slot
// (usher_mapper.cpp:292-388 with an empty ancestral list).
__global__ void __launch_bounds__(256) k_row_counts(const uint64_t *__restrict__ ent_off, uint32_t q0, const uint32_t *__restrict__ order, uint32_t nq,
                                                    const uint8_t *__restrict__ ref, const uint8_t *__restrict__ nuc, const uint8_t *__restrict__ is_missing,
                                                    uint32_t *__restrict__ dbottom, const unsigned long long *__restrict__ err) {
    const uint32_t slot = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, lane = threadIdx.x & 63u;
    if (slot >= nq) return;
    const uint32_t q = q0 + (order ? order[slot] : slot);
    const uint64_t rb = ent_off[q], re = (err && *err != ~0ull) ? rb : ent_off[q + 1];
    uint32_t d = 0;
    for (uint64_t r = rb + lane; r < re; r += 64) {
        const uint32_t rr = ref[r], a = is_missing[r] ? 15u : (uint32_t)nuc[r];
        d += (!is_missing[r] && (a & rr) == 0) ? 1u : 0u;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) d += __shfl_xor(d, o);
    if (lane == 0) dbottom[slot] = d;
}

// Query rows on arrival (one thread per VCF row): the sample each row belongs to (binary search in the CSR
// offsets) and the checks that validate_queries() used to run on the host -- rows of a sample sorted by
// position without duplicates, REF one of A,C,G,T, allele mask in range, REF equal to the tree's reference
// base at tree sites.  The first offending row (smallest row index, then smallest kind) is left in *err as
// (row << 3) | kind; ~0 = clean.
__global__ void k_rows_prepare(const uint64_t *__restrict__ ent_off, uint32_t n_queries, uint64_t n_ent,
                               const int32_t *__restrict__ pos, const uint8_t *__restrict__ ref, const uint8_t *__restrict__ nuc,
                               const uint8_t *__restrict__ is_missing, const int32_t *__restrict__ pos2site,
                               const uint8_t *__restrict__ site_ref, uint32_t max_pos, uint32_t n_sites,
                               uint32_t *__restrict__ ent_q, unsigned long long *__restrict__ err) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_ent) return;
    uint32_t lo = 0, hi = n_queries;   // last q with ent_off[q] <= e
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (ent_off[mid] <= e) lo = mid; else hi = mid; }
    ent_q[e] = lo;
    uint32_t kind = 0;
    const int32_t p = pos[e];
    const uint32_t r = ref[e];
    if (e > ent_off[lo] && p <= pos[e - 1]) kind = ROWS_UNSORTED;
    else if (r != 1 && r != 2 && r != 4 && r != 8) kind = ROWS_BAD_REF;
    else if (!is_missing[e] && (nuc[e] == 0 || nuc[e] > 15)) kind = ROWS_BAD_MASK;
    else if (p >= 0 && (uint32_t)p <= max_pos && n_sites) {
        const int32_t site = pos2site[p];
        if (site >= 0 && site_ref[site] != r) kind = ROWS_REF_MISMATCH;
    }
    if (kind) atomicMin(err, (unsigned long long)((e << 3) | kind));
}

hipError_t launch_rows_prepare(const uint64_t *ent_off, uint32_t n_queries, uint64_t n_ent, const int32_t *pos, const uint8_t *ref,
                               const uint8_t *nuc, const uint8_t *is_missing, const int32_t *pos2site, const uint8_t *site_ref,
                               uint32_t max_pos, uint32_t n_sites, uint32_t *ent_q, unsigned long long *err, hipStream_t s) {
    if (n_ent == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rows_prepare, dim3((uint32_t)((n_ent + 255) / 256)), dim3(256), 0, s, ent_off, n_queries, n_ent, pos, ref, nuc,
                       is_missing, pos2site, site_ref, max_pos, n_sites, ent_q, err);
    return hipGetLastError();
}

// ====================================================================
// One sample per lane, 32-bit walk (scores, tie lists, phase 2, fallback)
// ====================================================================

// Stream reader of the one-sample-per-lane walks: 64 words at a time, one per lane -- and, with each such window, the tile's
// 32-byte piece of the table row every word names (a header or key word names none, or a harmless one) staged in LDS, so
// that the mutation words of the window are served from there: one exposed memory latency per 64 words instead of one per
// mutation (these walks are chains of dependent loads).  One wave per block.
struct Reader {
    const uint32_t *p;
    uint32_t base, end;   // dword offsets; uniform
    uint32_t buf;         // lane l holds p[base + l]
    uint32_t cur;         // uniform
    const uint32_t *tab8; // the 64-sample tile's 8 dwords of row 0 (+ 64 * site)
    uint32_t *rowbuf;     // LDS [64][8]
    uint32_t n_sites;
    __device__ __forceinline__ void window(uint32_t lane) {
        buf = (base + lane < end) ? p[base + lane] : 0u;
        const uint32_t site = buf & 0x3FFFFFu;
        __syncthreads();   // (orders the previous window's LDS reads before these writes)
        if (base + lane < end && site < n_sites) {
            const uint4 *src = (const uint4 *)(tab8 + (uint64_t)site * 64);
            const uint4 lo = src[0], hi = src[1];
            uint4 *dst = (uint4 *)(rowbuf + lane * 8u);
            dst[0] = lo; dst[1] = hi;
        }
        __syncthreads();
    }
    __device__ __forceinline__ void init(const uint32_t *ptr, uint32_t begin, uint32_t end_, uint32_t lane, const uint32_t *tab8_, uint32_t *rowbuf_,
                                         uint32_t n_sites_) {
        p = ptr; base = begin; end = end_; cur = 0; tab8 = tab8_; rowbuf = rowbuf_; n_sites = n_sites_;
        window(lane);
    }
    __device__ __forceinline__ bool done() const { return base + cur >= end; }
    __device__ __forceinline__ uint32_t next(uint32_t lane) {
        if (cur == 64) {
            base += 64; cur = 0;
            window(lane);
        }
        uint32_t w = rdlane(buf, cur);
        cur++;
        return w;
    }
    // the sample column `col` (= lane >> 3) of the row named by the word that next() returned last
    __device__ __forceinline__ uint32_t row(uint32_t col) const { return rowbuf[(cur - 1u) * 8u + col]; }
};

struct WalkOut { uint32_t best, cnt, key; };

// Phase-2 walk of one chunk of the tie stream: count / key of the eligible nodes whose cost equals the
// lane's wanted score.  Only lanes with `relevant` set look for ties in this chunk (the chunk minimum of
// phase 1 equals their global minimum); a subtree is jumped over when D(node) - hsub > want for all of
// them, since cost(d) >= D(node) - hsub for every descendant d.
// Rows through LDS: the stream is read 64 words at a time (one per lane), and together with such a window every lane
// fetches the tile's 32-byte piece of the table row its word names (a header or key word names none, or a harmless one)
// into `rowbuf`; the mutation words of the window are then served from LDS.  One exposed memory latency per 64 words
// instead of one per mutation: this walk is a chain of dependent loads (k_ties 0.43 -> 0.28 ms at 16,384 samples).
// (Fetching the next window ahead into registers was tried: slower -- the walk jumps, and the registers cost occupancy.)
// LIST: the tied nodes are also appended to the samples' lists (a.tie_count / tie_j / tie_hu, indexed by the sample's position in
// the caller's batch: `list_q`) -- ugp_tied_nodes from the chunks that attain the minimum instead of a second walk of the tree.
template <bool LIST, bool MIN = false>
__device__ __forceinline__ WalkOut walk_ties(const PlaceArgs &a, uint32_t *slots, uint32_t *rowbuf, uint32_t tile, uint32_t c, uint32_t lane,
                                             uint32_t want, bool relevant, const uint32_t *rank2bfs, uint32_t list_q, uint32_t skip_bfs = 0xFFFFFFFFu) {
    const uint32_t *tab8 = a.table + ((uint64_t)(tile >> 3) * (a.n_sites + TABLE_CONST_ROWS) + TABLE_CONST_ROWS) * 64 + (tile & 7u) * 8;   // + 64 * site: 8 dwords
    const uint32_t col = lane >> 3;
    const uint32_t sh = (lane & 7u) * 4u;
    const uint32_t dbot = a.dbottom[tile * 64 + lane];
    WalkOut o; o.best = MIN ? 0xFFFFu : 0u; o.cnt = 0; o.key = 0;
    // D in the low half, B in the high half of one word (both below 2^15 on the packed path): B = the part of D at sites where
    // the sample's set holds the reference base -- the second pruning bound of k_best8 (ugp_flatten.hpp), here as well
    uint32_t dcur = 0;
    for (int phase = 0; phase < 2; phase++) {
        const uint32_t *p = phase == 0 ? a.pre_stream : a.stream_t;
        uint32_t pos = phase == 0 ? a.chunk_pre_off[c] : a.chunk_t_off[c];          // uniform
        const uint32_t end = phase == 0 ? a.chunk_pre_off[c + 1] : a.chunk_t_off[c + 1];
        uint32_t base = pos;
        uint32_t buf = 0;
        auto window = [&]() {   // words base .. base + 63 and the rows they name
            buf = (base + lane < end) ? p[base + lane] : 0u;
            const uint32_t site = buf & 0x3FFFFFu;
            __syncthreads();   // (one wave per block: orders the previous window's LDS reads before these writes)
            if (base + lane < end && site < a.n_sites) {
                const uint4 *src = (const uint4 *)(tab8 + (uint64_t)site * 64);
                const uint4 lo = src[0], hi = src[1];
                uint4 *dst = (uint4 *)(rowbuf + lane * 8u);
                dst[0] = lo; dst[1] = hi;
            }
            __syncthreads();
        };
        window();
        auto next = [&]() -> uint32_t {
            if (pos - base >= 64u) { base = pos; window(); }
            const uint32_t w = rdlane(buf, pos - base);
            pos++;
            return w;
        };
        bool have_info = false;
        uint32_t info = 0, info_hr = 255;
        while (pos < end) {
            const uint32_t w0 = next();
            const uint32_t key_w = next();
            const uint32_t nmut = w0 & 0xFFFFu;
            if (nmut == T_INFO_MARK) { have_info = true; info = key_w; info_hr = (w0 >> 16) & 0xFFu; continue; }
            // (bit 31 of the key word: the node was rewritten since the tree was flattened, ugp_mat_update -- no candidate any more)
            const uint32_t key = key_w & ~KEY_EXCLUDED;
            const bool excluded = (key_w & KEY_EXCLUDED) != 0;
            const uint32_t rslot = (w0 >> 16) & 63u, wslot = (w0 >> 22) & 63u;
            uint32_t xpar;   // (D | B << 16) of the parent
            if (rslot == RS_REG) xpar = dcur;
            else if (rslot == RS_BOTTOM) xpar = dbot;   // (B = 0 below the root)
            else xpar = slots[rslot * 64 + lane];
            const uint32_t dpar = xpar & 0xFFFFu;
            int tsum = 0, neg = 0;
            uint32_t common = 0, n_before = 0;
            for (uint32_t m = 0; m < nmut; m++) {
                const uint32_t w = next();
                const uint32_t mi = (w >> 22) & 3u, pi = (w >> 24) & 3u, ri = (w >> 26) & 3u;
                const uint32_t x = rowbuf[(pos - 1u - base) * 8u + col];
                const uint32_t nib = (x >> sh) & 15u;
                const int cc = (int)((nib >> mi) & 1u), pp = (int)((nib >> pi) & 1u);
                const int d = pp - cc;
                tsum += d + (((nib >> ri) & 1u) ? d * 65536 : 0);
                if (!(w & M_AFTER_MASK)) {
                    n_before++;
                    common += (uint32_t)cc;
                    neg += min(d, 0);
                }
            }
            const uint32_t xn = xpar + (uint32_t)tsum;   // (both halves are counts of a real state: no borrow survives the node)
            if (wslot != WS_NONE) slots[wslot * 64 + lane] = xn;
            dcur = xn;
            const uint32_t dn = xn & 0xFFFFu;
            if (!(w0 & F_NOSCORE)) {
                uint32_t cost, hu;
                bool elig;
                if (w0 & F_ROOT) {
                    cost = dn; elig = true; hu = 0;
                } else {
                    cost = dpar + (uint32_t)neg;
                    const bool masked = (w0 & F_MASKED) != 0;
                    const bool free_internal = !(w0 & F_LEAF) && !masked && nmut == 0;
                    elig = (common > 0) || free_internal;
                    hu = (masked || common != n_before) ? 1u : 0u;
                }
                // MIN (k_fix_skip): the smallest cost of a candidate of this chunk other than the lane's excluded node, instead of the ties
                if (MIN) {
                    if (elig && !excluded && cost < o.best && rank2bfs[key >> 1] != skip_bfs) o.best = cost;
                } else
                if (relevant && elig && !excluded && cost == want && (skip_bfs == 0xFFFFFFFFu || rank2bfs[key >> 1] != skip_bfs)) {
                    // (the extended searches rank ties by the caller's rule -- its node order, its distances -- and name nodes by their
                    // position in the caller's order: a.alt_rank / a.out_index by BFS index; a tie is a rare event)
                    const uint32_t rk = a.alt_rank ? (a.alt_rank[rank2bfs[key >> 1]] << 1) : key;
                    o.cnt++; o.key = max(o.key, rk | hu);
                    if (LIST) {
                        const uint32_t i = atomicAdd(&a.tie_count[list_q], 1u);
                        if (i < a.tie_cap) {
                            const uint32_t bfs = rank2bfs[key >> 1];
                            a.tie_j[(uint64_t)list_q * a.tie_cap + i] = a.out_index ? a.out_index[bfs] : bfs;
                            a.tie_hu[(uint64_t)list_q * a.tie_cap + i] = (uint8_t)hu;
                        }
                    }
                }
            }
            if (have_info) {
                have_info = false;
                const uint32_t hs = info >> 24;
                // D - hsub <= want and B - (second hits below) <= want: a descendant may still tie
                const bool near = MIN || (relevant && dn <= want + hs && (info_hr == 255u || (xn >> 16) <= want + info_hr));
                if (__builtin_amdgcn_ballot_w64(near) == 0) {
                    pos += info & 0xFFFFFFu;
                    if (pos - base >= 64u && pos < end) { base = pos; window(); }
                }
            }
        }
    }
    return o;
}

// MODE 0: full reduction (min, count, key)   MODE 1: per-node scores
// MODE 2: append tied nodes to lists         MODE 3: count / key of nodes with cost == want
// EX: the extended search of the other mapper2_body callers -- a node counts only if the shared mask admits it and it
// is not the sample's excluded node; ties are ranked by a.alt_rank; indices are reported through a.out_index.
template <int MODE, bool EX>
__device__ __forceinline__ WalkOut walk(const PlaceArgs &a, uint32_t *slots, uint32_t *rowbuf, uint32_t tile, uint32_t c0, uint32_t c1,
                                        uint32_t lane, uint32_t want_best) {
    const uint32_t *tab8 = a.table + ((uint64_t)(tile >> 3) * (a.n_sites + TABLE_CONST_ROWS) + TABLE_CONST_ROWS) * 64 + (tile & 7u) * 8;
    const uint32_t col = lane >> 3;
    const uint32_t sh = (lane & 7u) * 4u;
    const uint32_t q = tile * 64 + lane;
    const uint32_t dbot = a.dbottom[q];
    WalkOut o; o.best = 0x7fffffffu; o.cnt = 0; o.key = 0;
    uint32_t dcur = 0;
    uint32_t node_idx = a.chunk_node_off[c0];   // DFS index of the next body record
    const uint32_t skip_bfs = (EX && a.skip && q < a.n_queries) ? a.skip[q] : 0xFFFFFFFFu;
    for (int phase = 0; phase < 2; phase++) {
        Reader rd;
        if (phase == 0) rd.init(a.pre_stream, a.chunk_pre_off[c0], a.chunk_pre_off[c0 + 1], lane, tab8, rowbuf, a.n_sites);
        else rd.init(a.stream, a.chunk_body_off[c0], a.chunk_body_off[c1], lane, tab8, rowbuf, a.n_sites);
        while (!rd.done()) {
            const uint32_t w0 = rd.next(lane);
            const uint32_t key_w = rd.next(lane);
            const uint32_t key = key_w & ~KEY_EXCLUDED;
            const bool excluded = (key_w & KEY_EXCLUDED) != 0;   // rewritten since the tree was flattened (ugp_mat_update): scored, but no candidate
            const uint32_t nmut = w0 & 0xFFFFu;
            const uint32_t rslot = (w0 >> 16) & 63u, wslot = (w0 >> 22) & 63u;
            uint32_t dpar;
            if (rslot == RS_REG) dpar = dcur;
            else if (rslot == RS_BOTTOM) dpar = dbot;
            else dpar = slots[rslot * 64 + lane];
            int tsum = 0, neg = 0;
            uint32_t common = 0, n_before = 0;
            for (uint32_t m = 0; m < nmut; m++) {
                const uint32_t w = rd.next(lane);
                const uint32_t mi = (w >> 22) & 3u, pi = (w >> 24) & 3u;
                const uint32_t nib = (rd.row(col) >> sh) & 15u;
                const int c = (int)((nib >> mi) & 1u), p = (int)((nib >> pi) & 1u);
                const int d = p - c;
                tsum += d;
                if (!(w & M_AFTER_MASK)) {
                    n_before++;
                    common += (uint32_t)c;
                    neg += min(d, 0);
                }
            }
            const uint32_t dn = dpar + (uint32_t)tsum;
            if (wslot != WS_NONE) slots[wslot * 64 + lane] = dn;
            dcur = dn;
            if (!(w0 & F_NOSCORE)) {
                uint32_t cost, hu;
                bool elig;
                if (w0 & F_ROOT) {
                    cost = dn; elig = true; hu = 0;
                } else {
                    cost = dpar + (uint32_t)neg;
                    const bool masked = (w0 & F_MASKED) != 0;
                    const bool free_internal = !(w0 & F_LEAF) && !masked && nmut == 0;
                    elig = (common > 0) || free_internal;
                    hu = (masked || common != n_before) ? 1u : 0u;
                }
                uint32_t bfs = 0, rkey = key;
                bool cand = !excluded;
                if (EX || MODE == 1 || MODE == 2) bfs = a.dfs2bfs[node_idx];
                if (EX) {
                    cand = cand && (!a.node_mask || a.node_mask[bfs]) && bfs != skip_bfs;
                    if (a.alt_rank) rkey = a.alt_rank[bfs] << 1;
                }
                const uint32_t oidx = (EX && a.out_index) ? a.out_index[bfs] : bfs;
                if (MODE == 0) {
                    const uint32_t k = rkey | hu;
                    if (elig && cand) {
                        if (cost < o.best) { o.best = cost; o.cnt = 1; o.key = k; }
                        else if (cost == o.best) { o.cnt++; o.key = max(o.key, k); }
                    }
                    if (EX && a.scores && cand && q < a.n_queries) a.scores[(uint64_t)q * a.n_nodes + oidx] = (int32_t)(cost + (elig ? 0u : 1u));
                } else if (MODE == 1) {
                    if (q < a.n_queries) a.scores[(uint64_t)q * a.n_nodes + bfs] = (int32_t)(cost + (elig ? 0u : 1u));
                } else if (MODE == 2) {
                    if (elig && cand && cost == want_best) {
                        const uint32_t i = atomicAdd(&a.tie_count[q], 1u);
                        if (i < a.tie_cap) {
                            a.tie_j[(uint64_t)q * a.tie_cap + i] = oidx;
                            a.tie_hu[(uint64_t)q * a.tie_cap + i] = (uint8_t)hu;
                        }
                    }
                } else {
                    if (elig && cand && cost == want_best) { o.cnt++; o.key = max(o.key, key | hu); }
                }
                node_idx++;
            }
        }
    }
    return o;
}

template <int MODE, bool EX>   // MODE 0, 1, 2 (see walk)
__global__ void __launch_bounds__(64) k_place(PlaceArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t slots[];   // [max_slots][64]
    __shared__ __attribute__((aligned(16))) uint32_t rowbuf[64 * 8];
    const uint32_t lane = threadIdx.x;
    uint32_t tile, g;
    {
        const uint32_t b = blockIdx.x, G = a.n_groups;
        if ((G & 7u) == 0) {
            const uint32_t xcd = b & 7u, r = b >> 3, Gx = G >> 3;
            tile = r / Gx;
            g = (r % Gx) * 8 + xcd;
        } else {
            tile = b / G;
            g = b % G;
        }
    }
    const uint32_t c0 = (uint32_t)(((uint64_t)g * a.n_chunks) / a.n_groups);
    const uint32_t c1 = (uint32_t)(((uint64_t)(g + 1) * a.n_chunks) / a.n_groups);
    const uint64_t o = ((uint64_t)tile * a.n_groups + g) * 64 + lane;
    if (c0 >= c1) {
        if (MODE == 0) { a.part_best[o] = 0x7fffffffu; a.part_cnt[o] = 0; a.part_key[o] = 0; }
        return;
    }
    const uint32_t q = tile * 64 + lane;
    uint32_t want_best = 0;
    if (MODE == 2) want_best = (q < a.n_queries) ? (uint32_t)a.best_in[q] : 0xffffffffu;
    WalkOut r = walk<MODE, EX>(a, slots, rowbuf, tile, c0, c1, lane, want_best);
    if (MODE == 0) { a.part_best[o] = r.best; a.part_cnt[o] = r.cnt; a.part_key[o] = r.key; }
}

// Merge the per-group partial reductions of each sample (usher_mapper.cpp:
// 465-497 applied across groups) and translate the tie rank back to the BFS
// index the reference reports (*input.best_j).
__global__ void k_merge(const uint32_t *__restrict__ part_best, const uint32_t *__restrict__ part_cnt,
                        const uint32_t *__restrict__ part_key, const uint32_t *__restrict__ rank2bfs,
                        uint32_t n_groups, uint32_t n_queries, ugp_result *__restrict__ out) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    const uint32_t tile = q >> 6, lane = q & 63;
    uint32_t best = 0x7fffffffu, cnt = 0, key = 0;
    for (uint32_t g = 0; g < n_groups; g++) {
        const uint64_t o = ((uint64_t)tile * n_groups + g) * 64 + lane;
        const uint32_t b = part_best[o], c = part_cnt[o], k = part_key[o];
        if (c == 0) continue;
        if (b < best) { best = b; cnt = c; key = k; }
        else if (b == best) { cnt += c; key = max(key, k); }
    }
    ugp_result r;
    r.best_set_difference = (int32_t)best;
    r.num_best = cnt;
    r.best_j = rank2bfs[key >> 1];
    r.best_has_unique = key & 1u;
    out[q] = r;
}

// ====================================================================
// Per-node scores (-p), level by level
// ====================================================================
//
// out[q * n_nodes + j] for every sample q and every node j in the reference's breadth-first index
// (usher_common.cpp:406-412, 557-578): the product is its own output, 4 bytes per (node, sample), laid out with the
// NODE index running fastest.  A depth-first walk (k_place<1>) writes one sample's scores to addresses all over its
// 40 MB row -- one 4-byte store per 32-byte sector, 3.6 % of the HBM write rate.  Here the tree is walked in the
// output's own order: one launch per level of the breadth-first expansion (a level is an index range), one thread
// per node, so the 64 lanes of a wave store 64 consecutive scores of one sample: 256-byte stores.  What a depth-
// first walk gets for free, D(parent), comes from the previous level's D array (children of a node are neighbours:
// their lanes read the same 16 bytes), and each level leaves its own D behind -- only for nodes that have children.
// Per (node, 8 samples): one 16- or 32-byte D read, one table dword per mutation (a 32-byte sector out of L2), 8
// coalesced stores, one D write.  Algorithmic bytes per (node, sample): 4 out + 2 x sizeof(D) = 8 (12 with 32-bit D).
// The node's record is the 32-bit stream's (w0: mutation count, leaf / root / masked flags; one word per mutation).
// Needs a tree numbered as a breadth-first expansion (node_pair: {first child - 1, record offset} per node).
template <typename DT>
__global__ void __launch_bounds__(1024) k_scores_level(const uint2 *__restrict__ node_pair, const uint32_t *__restrict__ parent,
                                                      const uint32_t *__restrict__ stream, const uint32_t *__restrict__ table, uint32_t n_sites,
                                                      const uint32_t *__restrict__ dbottom, uint32_t lv_begin, uint32_t lv_end, uint32_t prev_begin,
                                                      const DT *__restrict__ d_prev, DT *__restrict__ d_cur, uint32_t d_stride /* nodes per block of SB samples */,
                                                      uint32_t qpad, uint32_t n_queries, uint64_t n_nodes, int32_t *__restrict__ scores) {
    const uint32_t n = lv_begin + blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= lv_end) return;
    const uint32_t *rec = stream + node_pair[n].y;
    const uint32_t w0 = rec[0];
    const uint32_t nmut = w0 & 0xFFFFu;
    const bool root = (w0 & F_ROOT) != 0, leaf = (w0 & F_LEAF) != 0, masked = (w0 & F_MASKED) != 0;
    const uint32_t m0 = nmut > 0 ? rec[2] : 0u, m1 = nmut > 1 ? rec[3] : 0u;   // (most nodes carry at most two mutations: kept in registers)
    // D arrays: [block of SB samples][node of the level][SB] -- the lanes of a wave (consecutive nodes) read and write
    // consecutive 128- or 256-byte pieces; the children of a node read the same piece
    constexpr uint32_t SB = SCORES_SB, NW = SB / 8u;   // samples per step, table dwords per step
    const uint64_t prow = root ? 0 : (uint64_t)(parent[n] - prev_begin) * SB;
    const uint64_t crow = (uint64_t)(n - lv_begin) * SB;
    const uint32_t n_rows = n_sites + TABLE_CONST_ROWS;
    const bool free_internal = !leaf && !masked && nmut == 0;
    // SB samples = NW dwords of a table row per step.  (SB = 64 reads a row's 32-byte sector once instead of twice -- a third fewer
    // bytes fetched, the same time: the kernel is bound by its writes -- and costs registers: 32.)
    for (uint32_t qb = blockIdx.y; qb * SB < qpad; qb += gridDim.y) {
        const uint32_t q0 = qb * SB;
        const uint32_t *trow = table + ((uint64_t)(q0 >> 9) * n_rows + TABLE_CONST_ROWS) * 64 + ((q0 & 511u) >> 3);   // + 64 * site: 16-byte aligned
        // per dword (8 samples) 4-bit counters, as in k_best8: P = prev in S, C = mut in S over all words (D), Cb / N = C and C & ~P over
        // the words in front of the node's first masked mutation (eligibility, cost).  A node with more than 15 words (rare) takes
        // the plain sums instead.
        uint32_t aP[NW], aC[NW], aN[NW], aCb[NW];
#pragma unroll
        for (uint32_t t = 0; t < NW; t++) { aP[t] = 0; aC[t] = 0; aN[t] = 0; aCb[t] = 0; }
        const bool small = nmut <= 15u;
        if (small) {
            for (uint32_t m = 0; m < nmut; m++) {
                const uint32_t w = m == 0 ? m0 : (m == 1 ? m1 : rec[2 + m]);
                uint32_t x[NW];
#pragma unroll
                for (uint32_t v = 0; v < NW / 4u; v++) {
                    const uint4 xv = *((const uint4 *)(trow + (uint64_t)(w & 0x3FFFFFu) * 64) + v);
                    x[4 * v] = xv.x; x[4 * v + 1] = xv.y; x[4 * v + 2] = xv.z; x[4 * v + 3] = xv.w;
                }
                const uint32_t mi = (w >> 22) & 3u, pi = (w >> 24) & 3u;
                const uint32_t bm = (w & M_AFTER_MASK) ? 0u : 0x11111111u;   // in front of the node's first masked mutation (usher_mapper.cpp:197-200)
#pragma unroll
                for (uint32_t t = 0; t < NW; t++) {
                    const uint32_t C = (x[t] >> mi) & 0x11111111u, P = (x[t] >> pi) & 0x11111111u;
                    aP[t] += P; aC[t] += C; aN[t] += C & ~P & bm; aCb[t] += C & bm;
                }
            }
        }
#pragma unroll
        for (uint32_t t = 0; t < NW; t++) {
            const uint32_t qg = q0 + t * 8;
            if (qg >= qpad) break;
            uint32_t dpar[8];
            if (root) {
#pragma unroll
                for (int j = 0; j < 8; j++) dpar[j] = dbottom[qg + j];
            } else {
                const DT *src = d_prev + (uint64_t)qb * d_stride * SB + prow + t * 8;
                if (sizeof(DT) == 2) {
                    const uint4 v = *(const uint4 *)src;
                    dpar[0] = v.x & 0xFFFFu; dpar[1] = v.x >> 16; dpar[2] = v.y & 0xFFFFu; dpar[3] = v.y >> 16;
                    dpar[4] = v.z & 0xFFFFu; dpar[5] = v.z >> 16; dpar[6] = v.w & 0xFFFFu; dpar[7] = v.w >> 16;
                } else {
                    const uint4 v0 = *(const uint4 *)src, v1 = *((const uint4 *)src + 1);
                    dpar[0] = v0.x; dpar[1] = v0.y; dpar[2] = v0.z; dpar[3] = v0.w; dpar[4] = v1.x; dpar[5] = v1.y; dpar[6] = v1.z; dpar[7] = v1.w;
                }
            }
            uint32_t dnew[8], negs[8], comm[8];
            if (small) {
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    dnew[j] = dpar[j] + ((aP[t] >> (4 * j)) & 15u) - ((aC[t] >> (4 * j)) & 15u);
                    negs[j] = (aN[t] >> (4 * j)) & 15u;
                    comm[j] = (aCb[t] >> (4 * j)) & 15u;
                }
            } else {   // a long branch: plain sums, one dword of the row per word
                int ts[8];
#pragma unroll
                for (int j = 0; j < 8; j++) { ts[j] = 0; negs[j] = 0; comm[j] = 0; }
                for (uint32_t m = 0; m < nmut; m++) {
                    const uint32_t w = rec[2 + m];
                    const uint32_t x = trow[(uint64_t)(w & 0x3FFFFFu) * 64 + t];
                    const uint32_t mi = (w >> 22) & 3u, pi = (w >> 24) & 3u;
                    const bool before = !(w & M_AFTER_MASK);
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const uint32_t nib = (x >> (4 * j)) & 15u;
                        const uint32_t c = (nib >> mi) & 1u, pp = (nib >> pi) & 1u;
                        ts[j] += (int)pp - (int)c;
                        if (before) { comm[j] += c; negs[j] += c & ~pp & 1u; }
                    }
                }
#pragma unroll
                for (int j = 0; j < 8; j++) dnew[j] = dpar[j] + (uint32_t)ts[j];
            }
            if (!leaf) {
                DT *dst = d_cur + (uint64_t)qb * d_stride * SB + crow + t * 8;
                if (sizeof(DT) == 2)
                    *(uint4 *)dst = make_uint4((dnew[0] & 0xFFFFu) | (dnew[1] << 16), (dnew[2] & 0xFFFFu) | (dnew[3] << 16), (dnew[4] & 0xFFFFu) | (dnew[5] << 16),
                                               (dnew[6] & 0xFFFFu) | (dnew[7] << 16));
                else { *(uint4 *)dst = make_uint4(dnew[0], dnew[1], dnew[2], dnew[3]); *((uint4 *)dst + 1) = make_uint4(dnew[4], dnew[5], dnew[6], dnew[7]); }
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                uint32_t sc;
                if (root) sc = dnew[j];                                        // cost(root) = D(root), always eligible (usher_mapper.cpp:454)
                else sc = dpar[j] - negs[j] + ((comm[j] > 0 || free_internal) ? 0u : 1u);   // + 1 when not eligible (:498-502)
                // (streaming stores: the matrix is written once and never read here -- it must not evict the 3 MB of table lines
                // that every thread gathers from)
                if (qg + j < n_queries) __builtin_nontemporal_store((int32_t)sc, &scores[(uint64_t)(qg + j) * n_nodes + n]);
            }
        }
    }
}

// ====================================================================
// Phase 1: 8 samples per lane, packed 16-bit, minimum only
// ====================================================================
//
// Lane l of the wave that owns 512-sample tile T holds samples T*512 + 8*l + j,
// j = 0..7 = nibble j of the lane's dword x of a table row.  For a mutation
// with allele indices (mut, prev):
//     C = (x >> mut)  & 0x11111111      bit 4j: mut  in S_j
//     P = (x >> prev) & 0x11111111      bit 4j: prev in S_j
// and per node three 4-bit-per-sample accumulators  accP += P, accC += C,
// accN += C & ~P  (flushed every 15 mutations).  At the end of a node they are
// widened to the packed layout {sample j | sample j+4 << 16}, j = 0..3:
//     e(acc, j) = (acc >> 4j) & 0x000F000F
//     D(n)  = D(par) + e(accP) - e(accC)
//     cost  = D(par) - e(accN)          ineligible samples (common == 0) -> bit 15 set
//     best  = min(best, cost)           v_pk_min_u16
// 16-bit counters are safe because the host only takes this path when
// max_rows(sample) + max_root_path_mutations(tree) + 2 < 0x7F7F: every D / cost then stays below the
// 0x7F7F the shared upper bounds start from (bit 15 is the ineligible flag).
//
// Instruction mix.  The walk is wave-uniform, and a compiler puts every uniform computation on the
// scalar ALU -- of which a CU has ONE, shared by its four SIMDs.  The first version of this kernel
// issued 1.2 scalar instructions per vector instruction and was bound by that unit (68 % busy,
// profiles/r02a) with the vector ALUs 25 % busy.  Here the decoding of the stream words is done by
// the vector units instead, eight words at a time, one word per lane (`decode`): row offset with
// the "row is reference-everywhere for this tile" substitution folded in, shift amounts, LDS byte
// offsets of the slots; a word then costs one v_readlane per field it actually uses, its flag
// tests are single-bit scalar compares, and everything unusual (pruning records, chunk ends, headers
// touching a cold slot) is one H_RARE bit away from the fast path.
//
// Instruction costs (tools/micro/issue_rate.hip, measured on MI355X): a plain 32-bit VALU instruction
// occupies its SIMD for 2 cycles, v_pk_*_u16 and v_readlane_b32 for 4; the scalar unit retires one
// instruction per cycle per CU -- so a scalar instruction and a 4-cycle vector instruction cost the CU the
// same, on different pipes.  Hence: the packed 16-bit pairs are added and subtracted with PLAIN 32-bit
// instructions wherever no carry or borrow can cross the halves (every D and cost is a count in
// [0, 0x7F7F), see `padd` / `psub`), v_pk_min_u16 is kept for the minima only, the far tests use a biased
// subtraction instead of packed minima, and field extraction is split between the two pipes.

struct Pk4 { uint32_t v[4]; };
#ifndef UGP_GRP
#define UGP_GRP 16   // (8 until round 4: with the long-node carries out of the loop there is room for 16 rows in flight -- 106 -> 114 VGPRs;
                     //  the walk alone 1.16 -> 1.14 ms at 10M nodes, 1.44 -> 1.34 on the SARS-CoV-2 shape, 1.08 -> 0.93 with 10,000 samples)
#endif
constexpr uint32_t GRP = UGP_GRP;   // stream words per pipeline group (= unroll factor of the walk)
constexpr uint32_t DYN_HEAD = 0, DYN_TAIL = 1, DYN_ACTIVE = 32;   // dyn_ctl: tickets taken / entries pushed (one line), live work (another)
constexpr unsigned long long DYN_EXIT = (1ull << 53) - 1ull;              // entry payload "no more work" (tile, c0, c1, flag bits all set)
constexpr uint32_t CONST_ROWS = TABLE_CONST_ROWS;  // rows 0..3 of every tile's table: all samples carry A / C / G / T

__device__ __forceinline__ uint32_t ex4(uint32_t acc, int j) { return (acc >> (4 * j)) & 0x000F000Fu; }
// ... and as bytes: i = 0: the counters of samples 0, 2, 4, 6; i = 1: of samples 1, 3, 5, 7
__device__ __forceinline__ uint32_t ex8(uint32_t acc, int i) { return (acc >> (4 * i)) & 0x0F0F0F0Fu; }
// Packed pairs with plain 32-bit arithmetic: exact as long as both halves of the true result lie in [0, 0xFFFF]
// (no carry out of / borrow into the low half).  Used for D(par) + e - e' and D(par) - e, whose results are counts.
__device__ __forceinline__ uint32_t padd(uint32_t a, uint32_t b) { return a + b; }
__device__ __forceinline__ uint32_t psub(uint32_t a, uint32_t b) { return a - b; }

// LBITS: the tile's active-row bitmap (one bit per site, a.active_words dwords) lives in LDS behind the slot rows, copied there
// whenever the wave moves to another tile.  A restart of the pipelined loop is then two dependent memory round trips (stream
// words, table rows) instead of three (words, bitmap bits, rows), and the steady loop issues one load less per group.  Costs
// ~3 KB of LDS per wave (fewer resident waves); the host takes this variant when the bitmap is small enough.
// ARG (the coarse pass): next to every chunk minimum, which node set it -- the low 16 bits of its last word's stream position, per
// sample -- so that the pass needs no phase 2 (k_coarse_result maps the position back to the node).
// TIES (phase 2 on the packed path; an experiment, UGP_PHASE2_PACKED=1 -- exact, but with exact bounds and a handful of samples that
// matter its units are chains of jumps, one pipeline restart per 1.5 words, where k_ties' 64-word windows absorb a jump for free: 1.8
// against 0.3 ms per 16,384 samples): the units are the (tile, chunk) pairs that left a record and hold some sample's global minimum; a.ub
// holds the samples' global minima; the samples of the tile whose minimum lies elsewhere take no part in the far tests (and tie with
// nothing: their costs in this chunk exceed their minimum); where a node's cost equals a sample's minimum -- a rare, uniform branch -- the node
// is found from its stream position and counted for that sample (a.tie_cnt, a.tie_key: what k_ties computes one sample per lane).
// LBITS == 2 (round 5): no bitmap at all -- every mutation word fetches its site's own row.  For batches whose tiles have (almost) every
// row live -- thousands of N cells per sample: BASELINE config 5 -- the constant-row shortcut never applies, and the bitmap costs every
// restart a dependent round trip and every group a load for nothing.
// B3 (round 5): the third pruning bound (ugp_flatten.hpp "B3"): a record of the body whose two tests fail asks for a restart, and the
// restart path -- outside the pipelined loop, where loads may wait -- reads the tile's block tables (a.b3: the largest number of
// "useful" events on a root path anywhere below the node, minus those above it) and tests again with hsub replaced by that number
// plus the second hits.  The pipelined loop itself carries nothing new.
constexpr uint32_t B3_MIN_JUMP = 12;   // smaller subtrees are cheaper to walk than to ask about
template <bool STATS, int LBITS, bool ARG, bool TIES, bool B3 = false>   // STATS: per-unit accounting for tuning (UGP_STATS); off in production, it costs SGPRs
__global__ void __launch_bounds__(64, TIES ? 3 : 4) k_best8(Best8Args a) {
    extern __shared__ __attribute__((aligned(16))) u32x4 slots8[];   // the hot saved slots: [lds_slots][64] x 16 B of D (the 8 B of B per lane and slot are in registers)
    const uint32_t lane = threadIdx.x;
    const uint32_t lane16 = lane * 16u;
    // (explicit address spaces for the slot rows and the cold scratch: through a plain pointer some of these became FLAT
    // accesses, which return out of order with the row loads -- with one of them possibly in flight the compiler turns
    // every later wait into a wait for all loads)
    typedef __attribute__((address_space(3))) u32x4 lds_row;
    typedef __attribute__((address_space(1))) u32x4 glb_row;
    typedef __attribute__((address_space(1))) u32x2 glb_row2;
    auto lds_at = [&](uint32_t byte_off) -> lds_row * { return (lds_row *)((__attribute__((address_space(3))) char *)slots8 + byte_off); };
    // Persistent wave: work units u = (tile, chunk range) are pulled from 8 queues, one per XCD.
    // Queue x owns a contiguous run of tiles, so an XCD walks few tiles at a time and their
    // non-reference table rows stay resident in its 4 MiB L2 (block b is observed to land on XCD
    // b % 8; placement affects speed only).  A wave whose queue is empty steals from the others.
    // Static one-block-per-unit launches lose ~40 % here: with pruning the unit durations differ
    // by >10x and the in-order, round-robin-per-XCD workgroup dispatcher stalls behind the slow XCD.
    const uint32_t home = blockIdx.x & 7u;
    uint32_t drained = 0;   // bit x: queue x is known to be empty
    // (dyn_ctl[DYN_ACTIVE] counts the waves that have started and are not waiting: a block that has not been scheduled yet -- the
    // device may be shared with another launch -- must never be waited for)
    if (lane == 0) atomicAdd(a.dyn_ctl + DYN_ACTIVE, 1u);
    // Upper bounds (+1) of the tile this wave worked on last, kept across units: the shared copy is read and
    // written with agent-scope accesses that leave the XCD, ~40 us of wave time per exchange, so it is
    // consulted when the wave moves to another tile and every a.ub_every chunk ends, not at every chunk end.
    Pk4 ub1;
#pragma unroll
    for (int j = 0; j < 4; j++) ub1.v[j] = 0x80008000u;
    uint32_t ub_tile = 0xFFFFFFFFu;   // uniform: tile whose bounds are in ub1
    uint32_t bits_tile = 0xFFFFFFFFu; // uniform (LBITS): tile whose active-row bitmap is in LDS
    uint32_t ub_age = 0;              // uniform: chunk ends since the last exchange
    for (;;) {
    const uint64_t t_pull0 = STATS ? __builtin_amdgcn_s_memtime() : 0;
    const uint64_t tr_pull = STATS ? __builtin_amdgcn_s_memrealtime() : 0;   // (trace: the device-wide 100 MHz clock, comparable across dies)
    uint32_t tile = 0, c0 = 0, c1 = 0;   // the unit: chunks [c0, c1) of the stream for one tile
    bool unit_heavy = false;             // (STATS) the unit lies in the tile's own region
    uint32_t uflags = 2u;   // bit 0: the unit lies in its tile's own region; bit 1: nothing left (exit)
    {
        // The units of queue x are listed in a.units (k_build_units: every tile's own region first, then the rest of the ring
        // nearest first); a ticket from the queue's counter names one.
        // (every value read here goes through readfirstlane: a condition on a per-lane value is formally divergent)
        auto uload = [&](const uint32_t *p_) -> uint32_t {
            return (uint32_t)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p_, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        };
        for (uint32_t t = 0; t < 8 && (uflags & 2u); t++) {
            const uint32_t x = (home + t) & 7u;
            if (drained & (1u << x)) continue;
            const uint32_t ux = a.unit_count[x];
            // (Every ticket comes from the counter, also a wave's first.  Reserving ticket blockIdx.x / 8 for the first pull -- to spare
            // the start of a launch its storm of atomics, 25-50 us per wave in the unit trace -- was tried in round 4 and HANGS: a
            // block that is scheduled late then owns a unit nobody else may take, starts after the others have seen the live-work
            // count reach zero and released the list, cuts its unit, and its pieces land behind tickets that are already spent.)
            uint32_t v = 0xFFFFFFFFu;
            if (ux) {
                if (lane == 0) v = atomicAdd(&a.queue[x], 1u);
                v = (uint32_t)__builtin_amdgcn_readfirstlane((int)v);
            }
            if (v >= ux) { drained |= 1u << x; continue; }
            const uint4 d = a.units[(uint64_t)a.unit_base[x] + v];
            tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.x);
            c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.y);
            c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.z);
            uflags = (uint32_t)__builtin_amdgcn_readfirstlane((int)d.w) & 1u;
        }
        // Static lists drained: the shared list of split-off units (see the split in the restart path below).  Tickets, not
        // races: a wave takes the next ticket h (one atomic) and waits for entry h -- its own word, so the waiting waves never
        // crowd one address (a compare-and-swap pop did: every push woke thousands of waves into one retry loop).
        // dyn_ctl[DYN_ACTIVE] counts running waves plus entries pushed but not yet finished; the wave that brings it to zero
        // knows that nothing can be pushed any more and writes an exit mark into the entries of all tickets that are or may
        // still be waiting.  (Which wave runs an entry never matters to the results.)
        if (uflags & 2u) {
            uint32_t last = 0;
            if (lane == 0) last = atomicAdd(a.dyn_ctl + DYN_ACTIVE, 0xFFFFFFFFu);   // this wave stops running ...
            last = (uint32_t)__builtin_amdgcn_readfirstlane((int)last);
            if (last == 1u) {   // ... and was the last thing alive: release every waiter, present and future (at most one more ticket per block)
                const uint32_t tl = uload(a.dyn_ctl + DYN_TAIL);
                const unsigned long long bye = ((unsigned long long)a.dyn_epoch << 53) | DYN_EXIT;
                for (uint32_t i = tl + lane; i < min(tl + gridDim.x + 64u, a.dyn_cap); i += 64u)
                    __hip_atomic_store(a.dyn_units + i, bye, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            uint32_t h = 0;
            if (lane == 0) h = atomicAdd(a.dyn_ctl + DYN_HEAD, 1u);
            h = (uint32_t)__builtin_amdgcn_readfirstlane((int)h);
            if (h < a.dyn_cap) {
                // An entry is ONE 64-bit word {epoch:11 | own region:1 | tile:12 | c1:20 | c0:20}, written and read with single
                // relaxed agent-scope accesses: nothing to order, hence no acquire / release (on this multi-die part those
                // write back or invalidate a die's whole L2).
                uint32_t naps = 1;
                for (;;) {
                    const unsigned long long ev = __hip_atomic_load(a.dyn_units + h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)ev), hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(ev >> 32));
                    if ((hi >> 21) == a.dyn_epoch) {
                        if ((hi & 0x1FFFFFu) != 0x1FFFFFu || lo != 0xFFFFFFFFu) {   // a unit (else: the exit mark)
                            c0 = lo & 0xFFFFFu; c1 = (lo >> 20) | ((hi & 0xFFu) << 12); tile = (hi >> 8) & 0xFFFu;
                            uflags = ((hi >> 20) & 1u) | 4u;   // (bit 2: a split-off unit; statistics only)
                        }
                        break;
                    }
                    for (uint32_t i = 0; i < naps; i++) __builtin_amdgcn_s_sleep(127);   // ~3.4 us each
                    naps = min(naps + 1u, 3u);
                }
            }
        }
    }
    // (opaque to the optimizer: without it the paths above are threaded through the whole unit body -- the kernel doubled in
    // size and spilled its pipelined loop)
    tile = (uint32_t)__builtin_amdgcn_readfirstlane((int)tile); c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c0);
    c1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)c1); uflags = (uint32_t)__builtin_amdgcn_readfirstlane((int)uflags);
    asm volatile("" : "+s"(tile), "+s"(c0), "+s"(c1), "+s"(uflags));
    if (uflags & 2u) return;
    unit_heavy = (uflags & 1u) != 0;
    if (c0 >= c1) continue;
    // The dense units of the tiles' own regions are the kernel's critical path (one wave walks ~10k words; everything else
    // fits beside them): their waves get the SIMD's issue slots first, the far units run in the gaps.
    if (a.heavy_prio) { if (unit_heavy) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0); }
    // table rows through a buffer resource: address = tile base + 4*lane (VGPR) + 256*row (SGPR soffset);
    // rows 0..3 are the constant rows, site s is row s + 4
    const uint32_t n_rows = a.n_sites + CONST_ROWS;
    const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(a.table + (uint64_t)tile * n_rows * 64), 0, (int)(n_rows * 256u), 0x00020000);
    const uint32_t lane4 = lane * 4u;
    Pk4 dbot;
    {
        const uint32_t *db = a.dbottom + (uint64_t)tile * 512 + lane * 8;
#pragma unroll
        for (int j = 0; j < 4; j++) dbot.v[j] = (db[j] & 0xFFFFu) | (db[j + 4] << 16);
    }
    Pk4 irr;   // (TIES) 0x8000 per sample whose global minimum is not attained in this chunk: it takes no part in the far tests
#pragma unroll
    for (int j = 0; j < 4; j++) irr.v[j] = 0;
    if (TIES) {
        const uint4 lb = *(const uint4 *)(a.lbest + (((uint64_t)c0 * a.n_tiles + tile) * 64 + lane) * 4);
        const uint4 gb = *(const uint4 *)(a.ub + ((uint64_t)tile * 64 + lane) * 4);
        irr.v[0] = pk_min(lb.x ^ gb.x, 0x00010001u) << 15; irr.v[1] = pk_min(lb.y ^ gb.y, 0x00010001u) << 15;
        irr.v[2] = pk_min(lb.z ^ gb.z, 0x00010001u) << 15; irr.v[3] = pk_min(lb.w ^ gb.w, 0x00010001u) << 15;
    }
    Pk4 best, dcur, dpar;
    // (round 6) the SECOND smallest cost of the open chunk, per sample: a chunk whose minimum is attained by exactly one node has
    // sec > best.  Phase 2 then needs no walk for a sample whose global minimum lies in one such chunk only -- the node is the one the
    // seed descent found (k_descend records it): k_select / k_final.  Two packed instructions per node and pair of samples.
    constexpr bool UNIQ = !ARG && !TIES;
    Pk4 sec;
#pragma unroll
    for (int j = 0; j < 4; j++) { best.v[j] = 0xFFFFFFFFu; sec.v[j] = 0xFFFFFFFFu; dcur.v[j] = dbot.v[j]; dpar.v[j] = 0; }   // (dcur: the root reads D(bottom) as "the previous node's D")
    // A node with more than 15 mutation words overflows the 4-bit counters: its header carries H_SLOW, so it is walked by slow_node
    // (outside the pipelined loop), which spills the counters into these packed carries every 15 words (M_FLUSH).  They exist only
    // there: held across the pipelined loop they cost 14 vector registers that the loop needs for rows in flight.
    struct Carry { Pk4 D, N, C; uint32_t B[2]; bool flushed; };
    uint32_t accP = 0, accC = 0, accN = 0;
    Pk4 bpos;                  // (ARG) per sample: position (low 16 bits) of the node that set `best`
#pragma unroll
    for (int j = 0; j < 4; j++) bpos.v[j] = 0;
    uint32_t pos_base = 0;     // (ARG) uniform: stream position of the word at relative position 0 of the range being walked
    // best = min(best, c) per half; with ARG the halves that improve take the position p2 (both halves = the node's position)
    Pk4 tie_t;                 // (TIES) per half: 1 where the open node's cost equals the sample's global minimum
    uint32_t tie_any = 0, accU = 0;   // (TIES) accU: like accC, the node's mutations the sample does NOT share (has_unique); never flushed: bits only
    // (TIES) a tie is handled outside the pipelined loop, like a pruning jump: the node's end asks for a restart behind itself and leaves
    // what the handler needs -- its position (in a vector register, same value in every lane: see stop_v; 0xFFFFFFFF = none) and accU
    uint32_t tie_u = 0, tie_pos_v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(tie_pos_v) : "s"(0xFFFFFFFFu));
#pragma unroll
    for (int j = 0; j < 4; j++) tie_t.v[j] = 0;
    auto take_min = [&](int j, uint32_t c, uint32_t p2) {
        if (TIES) {
            tie_t.v[j] = pk_min(c ^ pk_sub(ub1.v[j], 0x00010001u), 0x00010001u) ^ 0x00010001u;
            tie_any |= tie_t.v[j];
            return;
        }
        const uint32_t nb = pk_min(best.v[j], c);
        if (UNIQ) sec.v[j] = pk_min(sec.v[j], pk_max(best.v[j], c));
        if (ARG) {
            const uint32_t t = pk_min(nb ^ best.v[j], 0x00010001u);   // 1 per half that changed
            const uint32_t msk = (t << 16) - t;                        // 0xFFFF per such half
            bpos.v[j] = (bpos.v[j] & ~msk) | (p2 & msk);
        }
        best.v[j] = nb;
    };
    // B(n, s) = the part of D at sites where the sample's set holds the reference base (second pruning bound, ugp_flatten.hpp):
    // one byte per sample (b[0]: samples 0, 2, 4, 6 of the lane, b[1]: samples 1, 3, 5, 7), saved and restored with D.
    // B(bottom) = 0: below the root every state is the reference base.
    uint32_t bcur[2] = {0, 0}, bpar[2] = {0, 0};
    uint32_t accPB = 0, accCB = 0;   // like accP / accC, over the samples whose set holds the site's reference base
    // The B halves of the hot slots stay in registers (indexed with the slot number, which is uniform): the kernel's occupancy
    // is set by its LDS, and 8 more bytes per lane and slot there cost a third of the resident waves; 32 registers cost none.
    u32xhot bs0 = 0, bs1 = 0;
    uint32_t hdr = 0;          // uniform: header of the open node
    uint32_t chunk = c0;       // uniform: chunk whose body is being walked
    // The unit ends in front of chunk `stop`: c1, or less once its second half has been handed to another wave.  Kept in a
    // vector register on purpose (the same value in every lane, read back with readfirstlane at chunk ends and restarts
    // only): the kernel sits at the scalar-register limit, and one more scalar that lives across the pipelined loop
    // sends its allocation over a cliff (500 more spill moves).
    uint32_t stop_v;
    asm volatile("v_mov_b32 %0, %1" : "=v"(stop_v) : "s"(c1));
    uint32_t mark_v;   // (same trick) the chunk the wave was in at its last look at the shared list: how fast is it getting on?
    asm volatile("v_mov_b32 %0, %1" : "=v"(mark_v) : "s"(c0));
    const uint32_t NOPW = H_TAG | H_RARE | H_NOP;
    // pruning: ub1 = (upper bound of best(s)) + 1 per sample, refreshed from / published to a.ub at chunk ends
    bool prune = false;        // uniform; only while walking the body (phase 1)
    bool pre_prune = false;    // uniform: replaying the preamble with its pruning records (one per path node)
    uint32_t body_start = 0;   // uniform: where the body walk begins (behind the subtree of a path node found far)
    bool have_info = false;    // uniform
    bool have_sinfo = false;   // uniform: a sibling record waits for the next header
    uint32_t sinfo = 0;        // uniform
    uint32_t info = 0;         // uniform: pending pruning record
    uint32_t skip_to = 0;      // uniform: restart request (0 = none)
    uint32_t pre_off = 0xFFFFFFFFu, pre_w0 = 0, pre_w1 = 0, pre_w2 = 0;   // (B3) the first three word groups of the next refill, loaded beside a third-bound question
    uint32_t cend = 0xFFFFFFFFu;   // uniform: position of the open chunk's end marker (phase 1)
    bool cend_stale = false;       // uniform: `chunk` advanced inside the pipeline and the end word did not say where the next chunk ends
    uint32_t t_mark = (uint32_t)__builtin_amdgcn_s_memtime();   // uniform: start of the unit / its last look at the shared list (low word: differences only)
    uint32_t n_split = 0;      // uniform (STATS)
    uint64_t n_skipped = 0;    // uniform (STATS): words jumped over (each jump counted up to the end of the unit's body: st_body words)
    uint32_t st_body = 0;
    uint64_t n_eval[2] = {0, 0};   // uniform (STATS): nodes evaluated in bodies / in preamble replays (and unpruned walks)
    uint32_t run_nodes = 0;    // uniform (STATS): nodes completed since the last restart
    uint64_t n_first_skip = 0; // uniform (STATS): jumps decided by the first node after a restart
    uint32_t n_cause[4] = {0, 0, 0, 0};   // uniform (STATS): restarts by cause -- jump, sibling jump, chunk end, slow header
    uint32_t n_jlen[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // uniform (STATS): jump lengths
    uint32_t n_b3[2] = {0, 0};   // uniform (STATS): third-bound tests at a restart / of which decided the jump
    auto count_jump = [&](uint32_t len, int cause) {
        n_cause[cause]++;
        n_jlen[len < 8 ? 0 : len < 16 ? 1 : len < 32 ? 2 : len < 64 ? 3 : len < 128 ? 4 : len < 512 ? 5 : len < 4096 ? 6 : 7]++;
    };
    // (the tests below use the kernel argument, not the per-lane pointer: a condition derived from `lane`
    // is formally divergent, and one such flag turned the whole walk's control flow -- jump target, open
    // header, position -- into vector registers with exec-mask branches)
    const bool can_prune = a.ub != nullptr;   // uniform
    uint32_t *ubp = a.ub + ((uint64_t)tile * 64 + lane) * 4;
    auto exchange_ub = [&]() {   // ub = min(ub, what other waves found, this chunk's minimum); racy but every value is a real cost
        if (!can_prune || TIES) return;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            uint32_t u = __hip_atomic_load(ubp + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            u = pk_min(pk_min(u, best.v[j]), pk_sub(ub1.v[j], 0x00010001u));
            __hip_atomic_store(ubp + j, u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            ub1.v[j] = pk_add(u, 0x00010001u);
        }
    };
    // A chunk's minima matter only if some sample's minimum is within its upper bound (the global minimum
    // never exceeds the bound): only then are they stored, and the chunk is appended to the tile's list of records
    // for phase 2.  Most chunks are far from the tile's samples and end without a store.
    auto chunk_has_candidate = [&]() -> bool {
        if (!can_prune || TIES) return true;
        uint32_t t = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) t |= pk_min(best.v[j], ub1.v[j]) ^ ub1.v[j];   // non-zero where best < ub + 1
        return __builtin_amdgcn_ballot_w64(t != 0) != 0;
    };
    auto chunk_end = [&]() {     // publish the chunk-local minimum, start the next chunk
        if (TIES) { chunk++; return; }
        if (chunk_has_candidate()) {
            uint4 *dst = (uint4 *)(a.lbest + (((uint64_t)chunk * a.n_tiles + tile) * 64 + lane) * 4);
            *dst = make_uint4(best.v[0], best.v[1], best.v[2], best.v[3]);
            if (ARG) *(uint4 *)(a.lpos + (((uint64_t)chunk * a.n_tiles + tile) * 64 + lane) * 4) = make_uint4(bpos.v[0], bpos.v[1], bpos.v[2], bpos.v[3]);
            if (UNIQ && a.luniq) {
                // one byte per lane: bit j + 4h set = sample j + 4h of the lane has at least two nodes at its chunk minimum (or no candidate)
                uint32_t nu = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t eq = pk_min(sec.v[j] ^ best.v[j], 0x00010001u) ^ 0x00010001u;   // 1 per half whose two smallest costs are equal
                    nu |= ((eq & 1u) << j) | ((eq >> 16) << (j + 4));
                }
                ((uint8_t *)a.luniq)[((uint64_t)chunk * a.n_tiles + tile) * 64 + lane] = (uint8_t)nu;
            }
            if (lane == 0) a.list[(uint64_t)tile * a.n_chunks + atomicAdd(&a.list_n[tile], 1u)] = chunk;   // (order is irrelevant to phase 2)
        }
        if (can_prune && !a.freeze_ub) {
            if (++ub_age >= a.ub_every) { exchange_ub(); ub_age = 0; }
            else {
#pragma unroll
                for (int j = 0; j < 4; j++)   // what this wave found itself ("no candidate" is 0xFFFF: take the minimum before the +1)
                    ub1.v[j] = pk_add(pk_min(pk_sub(ub1.v[j], 0x00010001u), best.v[j]), 0x00010001u);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) { best.v[j] = 0xFFFFFFFFu; if (UNIQ) sec.v[j] = 0xFFFFFFFFu; }
        chunk++;
    };
    // The chunks that lie wholly in front of relative position `to` (none of them walked, so none holds a candidate) are
    // closed in one step: their end markers are looked up by the lanes in parallel instead of one dependent load per chunk.
    // `body0` = stream position of the unit's first body word.  Returns the end-marker position of the chunk left open.
    auto close_empty_chunks = [&](uint32_t to, uint32_t body0) -> uint32_t {
        for (;;) {
            const uint32_t left = c1 - chunk;   // uniform
            if (!left) return 0xFFFFFFFFu;
            const uint32_t mark = lane < left ? a.chunk8_body_off[chunk + 1u + lane] - 1u - body0 : 0xFFFFFFFFu;   // end marker of chunk + lane
            const uint32_t k = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < left && to > mark));
            if (k) {
                chunk += k;
                if (can_prune && !a.freeze_ub) { ub_age += k; if (ub_age >= a.ub_every) { exchange_ub(); ub_age = 0; } }
            }
            if (k < (left < 64u ? left : 64u)) return rdlane(mark, k);
        }
    };
    // "can everything below / beside this node be skipped?": true when, for all 512 samples,
    //     D - hs > ub   or   B - hr > ub            (ub = upper bound of best(s); the two lower bounds of ugp_flatten.hpp)
    // Per half: 0x8000 + D - hs - (ub + 1) keeps bit 15 exactly when D - hs > ub; D < 0x7F7F, ub + 1 + hs <= 0x7FFF, and
    // in the second test B <= 255, hr <= 6, so no half borrows from its neighbour and plain 32-bit arithmetic is exact.
    // (b[0] holds the lane's samples 0, 2, 4, 6 as bytes, b[1] 1, 3, 5, 7; d.v[j] holds samples j and j + 4 as halves.)
    auto all_far = [&](const Pk4 &d, const uint32_t (&b)[2], uint32_t rec) -> bool {
        const uint32_t hs = (rec >> INFO_HS_SHIFT) & 0x7Fu;
        const uint32_t K = 0x80008000u - hs * 0x00010001u;
        const uint32_t hr = (rec >> INFO_HR_SHIFT) & 7u;
        uint32_t r = 0xFFFFFFFFu;
        if (hr != INFO_HR_NONE) {
            const uint32_t K2 = 0x80008000u - hr * 0x00010001u;
            const uint32_t bj[4] = {b[0] & 0x00FF00FFu, b[1] & 0x00FF00FFu, (b[0] >> 8) & 0x00FF00FFu, (b[1] >> 8) & 0x00FF00FFu};
            if (pre_prune && hs == PRE_HS_NONE) {   // preamble record whose hsub does not fit: the second bound only
#pragma unroll
                for (int j = 0; j < 4; j++) r &= psub(padd(bj[j], K2), ub1.v[j]) | (TIES ? irr.v[j] : 0u);
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) r &= psub(padd(d.v[j], K), ub1.v[j]) | psub(padd(bj[j], K2), ub1.v[j]) | (TIES ? irr.v[j] : 0u);
            }
        } else {
            if (pre_prune && hs == PRE_HS_NONE) return false;
#pragma unroll
            for (int j = 0; j < 4; j++) r &= psub(padd(d.v[j], K), ub1.v[j]) | (TIES ? irr.v[j] : 0u);
        }
        return __builtin_amdgcn_ballot_w64((r & 0x80008000u) != 0x80008000u) == 0;
    };

    // Saved-D slots beyond a.lds_slots live in a small global scratch (one access per ~1,300 words at 10M
    // nodes; 7 KB of LDS per wave lets 20 waves share a CU).  The pipelined loop never touches it -- a store
    // there would share vmcnt with the row loads and force vmcnt(0) waits: a header that reads or writes a cold
    // slot (or the root, whose parent value is D_bottom) carries H_SLOW | H_RARE, asks for a restart at its
    // own position, and the restart code walks that one node with `slow_node`, the general form of the step.
    __attribute__((address_space(1))) uint32_t *coldp = (__attribute__((address_space(1))) uint32_t *)a.cold + ((uint64_t)blockIdx.x * (a.max_slots > a.lds_slots ? a.max_slots - a.lds_slots : 0u) * 64 + lane) * 8;   // per lane and slot 32 B: D (16), B (8), unused (8)
    bool replay = false;       // uniform: restart at skip_to - 1 and walk one node with slow_node

    // ---- end of the open node (shared by the fast and the slow step); wa = LDS byte offset of the write slot,
    // or 0xFFFFFFFF with `cold_ws` >= 0 for a cold one
    // (TIES) the open node ties for some sample of the tile: which node is it -- the last one whose words begin at or in front of
    // the position, found by all lanes together in two or three steps -- and for whom
    auto tie_event = [&]() {
        const uint32_t pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)tie_pos_v);
        asm volatile("v_mov_b32 %0, %1" : "=v"(tie_pos_v) : "s"(0xFFFFFFFFu));
        if (pos == 0xFFFFFFFFu) return;
        const uint32_t apos = pos_base + pos;
        uint32_t lo = a.chunk_node_off[chunk], hi = a.chunk_node_off[chunk + 1u];   // uniform
        while (hi - lo > 1u) {
            const uint32_t st = (hi - lo + 63u) / 64u;
            const uint32_t idx = lo + lane * st;
            const bool le = idx < hi && a.node_pos8[idx] <= apos;   // (a prefix of the lanes: positions ascend with the DFS index)
            const uint32_t k = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(le));
            lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(lo + (k ? k - 1u : 0u) * st));
            hi = min(lo + st, hi);
        }
        const uint32_t key = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.rank_dfs[lo]) << 1;
#pragma unroll
        for (int j = 0; j < 4; j++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                if ((tie_t.v[j] >> (16 * h)) & 1u) {
                    const uint32_t q = tile * 512u + lane * 8u + (uint32_t)j + 4u * (uint32_t)h;
                    if (q < a.n_queries) {
                        const uint32_t hu = (tie_u >> (4 * (j + 4 * h))) & 1u;   // some mutation of the node is not the sample's
                        atomicAdd(&a.tie_cnt[q], 1u);
                        atomicMax(&a.tie_key[q], key | hu);
                    }
                }
            }
        }
    };
    auto node_end = [&](uint32_t pos, int cold_ws, Carry *cy) -> bool {   // true: a pruning jump was requested (skip_to); cy: the carries of a long node (slow_node), else nullptr
        const uint32_t p2 = ARG ? ((pos_base + pos) & 0xFFFFu) * 0x00010001u : 0u;
        // A sample is ineligible here when it shares no mutation with the branch (common == 0,
        // usher_mapper.cpp:454-455) unless the node is "free": z has bit 4j set for such samples
        // and is turned into a 0x8000 penalty on the 16-bit cost (valid costs stay below 0x8000).
        if (cy && cy->flushed) {   // node with more than 15 mutations (rare): fold the carries in
#pragma unroll
            for (int j = 0; j < 4; j++) dcur.v[j] = pk_add(pk_sub(pk_add(dpar.v[j], ex4(accP, j)), ex4(accC, j)), cy->D.v[j]);
#pragma unroll
            for (int i = 0; i < 2; i++) bcur[i] = bpar[i] + ex8(accPB, i) - ex8(accCB, i) + cy->B[i];
            if (!(hdr & H_NOSCORE)) {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t cost = pk_sub(pk_sub(dpar.v[j], ex4(accN, j)), cy->N.v[j]);
                    const uint32_t common = pk_add(ex4(accC, j), cy->C.v[j]);
                    uint32_t pen = 0;
                    if (!(hdr & H_FREE)) pen = (((common & 0xFFFFu) ? 0u : 0x8000u) | ((common >> 16) ? 0u : 0x80000000u));
                    take_min(j, cost | pen, p2);
                }
            }
        } else {
            if (!(hdr & H_SKIPD)) {
#pragma unroll
                for (int j = 0; j < 4; j++) dcur.v[j] = psub(padd(dpar.v[j], ex4(accP, j)), ex4(accC, j));
#pragma unroll
                for (int i = 0; i < 2; i++) bcur[i] = bpar[i] + ex8(accPB, i) - ex8(accCB, i);
            }
            // (round 6) A node with which NO sample of the tile shares a mutation (accC == 0 in every lane) is eligible for none of them
            // (usher_mapper.cpp:454-455: common > 0, unless the node is "free") -- its costs would all carry the penalty and lose against
            // any real candidate: one compare and a branch instead of ~40 vector instructions.  That is nearly every node a tile evaluates
            // away from its samples' lineages: the walk without pruning went 5.5 -> ?? ms per 2 048 samples.
            if (!(hdr & H_NOSCORE) && (TIES || (hdr & H_FREE) || __builtin_amdgcn_ballot_w64(accC != 0) != 0)) {
                uint32_t z = accC | (accC >> 1);
                z |= z >> 2;
                z = ~z & ((hdr & H_FREE) ? 0u : 0x11111111u);   // bit 4j: sample j shares no mutation with this branch
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const uint32_t cost = psub(dpar.v[j], ex4(accN, j));
                    const uint32_t pen = (j == 3 ? (z << 3) : (z << (15 - 4 * j))) & 0x80008000u;
                    take_min(j, cost | pen, p2);
                }
            }
        }
        if (hdr & H_STORE) {
            const u32x4 v = u32x4{dcur.v[0], dcur.v[1], dcur.v[2], dcur.v[3]};
            const u32x2 vb = u32x2{bcur[0], bcur[1]};
            if (cold_ws >= 0) { *(glb_row *)(coldp + (uint64_t)cold_ws * 512) = v; *(glb_row2 *)(coldp + (uint64_t)cold_ws * 512 + 4) = vb; }
            else {
                *lds_at(((hdr >> (H_WSLOT_SHIFT - 10)) & (63u << 10)) | lane16) = v;
                const uint32_t wsi = (hdr >> H_WSLOT_SHIFT) & HOT_MASK;
                bs0[wsi] = vb.x; bs1[wsi] = vb.y;
            }
        }
        bool tie_here = false;   // uniform
        if (TIES) {
            if (__builtin_amdgcn_ballot_w64(tie_any != 0) != 0) {
                tie_here = true;
                tie_u = accU;
                asm volatile("v_mov_b32 %0, %1" : "=v"(tie_pos_v) : "s"(pos));
            }
            tie_any = 0; accU = 0;
        }
        accP = accC = accN = accPB = accCB = 0;
        if (STATS) { run_nodes++; n_eval[pre_prune || !prune ? 1 : 0]++; }
        if (have_info) {   // this node carries a pruning record: can its whole subtree be skipped?
            have_info = false;
            if (all_far(dcur, bcur, info)) {
                if (pre_prune) {   // a path node: nothing of its subtree is needed -- end the replay, start the body behind it
                    body_start = info & INFO_JUMP_MASK;
                    skip_to = 0x7FFFFFFFu;
                    if (STATS) n_skipped += min(body_start, st_body);
                    return true;
                }
                skip_to = pos + 1 + (info & INFO_JUMP_MASK);
                if (STATS) { n_skipped += min(info & INFO_JUMP_MASK, st_body > pos + 1u ? st_body - (pos + 1u) : 0u); if (run_nodes == 1) n_first_skip++; count_jump(info & INFO_JUMP_MASK, 0); }
                return true;
            }
            if (B3 && !pre_prune) {   // both tests failed: is the third bound worth a look?  (bit 31 of skip_to: "test again at the restart")
                const uint32_t hs3 = (info >> INFO_HS_SHIFT) & 0x7Fu, hr3 = (info >> INFO_HR_SHIFT) & 7u;
                // (... only if the best the third bound can give -- no useful mutation anywhere below: hsub replaced by the second hits --
                // would decide the jump: on a sample's own lineage it never does, and every question costs a restart)
                if (hr3 != INFO_HR_NONE && hs3 > hr3 && (info & INFO_JUMP_MASK) >= B3_MIN_JUMP &&
                    all_far(dcur, bcur, (info & ~(0x7Fu << INFO_HS_SHIFT)) | (hr3 << INFO_HS_SHIFT))) { skip_to = (pos + 1u) | 0x80000000u; return true; }
            }
        }
        if (TIES && tie_here) { skip_to = pos + 1; return true; }   // (the walk goes on behind the node once the tie has been booked)
        return false;
    };
    auto flush_acc = [&](Carry &cy) {   // 15 mutations in the 4-bit counters: spill to the packed carries (slow_node only)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const uint32_t eP = ex4(accP, j), eC = ex4(accC, j), eN = ex4(accN, j);
            cy.D.v[j] = pk_sub(pk_add(cy.D.v[j], eP), eC);
            cy.N.v[j] = pk_add(cy.N.v[j], eN);
            cy.C.v[j] = pk_add(cy.C.v[j], eC);
        }
#pragma unroll
        for (int i = 0; i < 2; i++) cy.B[i] += ex8(accPB, i) - ex8(accCB, i);   // (bytes may borrow from each other here: the node's total is exact)
        accP = accC = accN = accPB = accCB = 0;
        cy.flushed = true;
    };
    // sibling record waiting at a header: can this child and the non-last siblings after it all be skipped?
    // (Round 5, tried and removed: the third bound for the run -- the tile's useful mutations anywhere in the run, measured from
    // cum_under of the header's block, which is a lower bound of the parent's count -- asked like a node's record.  Exact (model and
    // GPU tests), but a run of siblings is long and some subtree in it nearly always holds a useful mutation: 124 k questions per
    // launch, 40 k decided; restarts 404 k -> 425 k, the kernel 0.90 -> 0.94 ms; on the SARS-CoV-2 shape 201 k asked, 32 k decided,
    // 1.30 -> 1.90 ms.  A failed question costs a restart the plain test never needed.)
    auto sibling_test = [&](uint32_t pos) -> bool {
        have_sinfo = false;
        if (!all_far(dpar, bpar, sinfo)) return false;
        skip_to = pos + (sinfo & INFO_JUMP_MASK);   // the start of the parent's last child
        have_info = false;
        if (STATS) { n_skipped += min(sinfo & INFO_JUMP_MASK, st_body > pos ? st_body - pos : 0u); count_jump(sinfo & INFO_JUMP_MASK, 1); }
        return true;
    };
    // words that leave the fast path (H_RARE); true: the pipeline has to restart at skip_to
    auto rare_word = [&](uint32_t w, uint32_t pos) -> bool {
        if (w & H_INFO) {
            if (prune || pre_prune) {
                if (w & H_SIB) { sinfo = w; have_sinfo = true; }   // about the node that follows and its later siblings
                else { info = w; have_info = true; }               // about the node that follows and its descendants
            }
            return false;
        }
        if (w & H_CHUNK_END) {
            // a chunk with a candidate (or a due exchange of bounds) is closed by the restart code, which may
            // store; any other chunk just ends here, inside the pipeline
            // (also when the unit ends with this chunk because its rest has been handed to another wave)
            if (chunk_has_candidate() || ub_age + 1 >= a.ub_every || chunk + 1u >= (uint32_t)__builtin_amdgcn_readfirstlane((int)stop_v)) { skip_to = pos + 1; if (STATS) n_cause[2]++; return true; }
            ub_age++;
#pragma unroll
            for (int j = 0; j < 4; j++) best.v[j] = 0xFFFFFFFFu;
            chunk++;
            {   // the end word names the length of the chunk it opens
                const uint32_t ln = (w >> CE_LEN_SHIFT) & CE_LEN_MASK;
                if (ln) cend = pos + ln; else cend_stale = true;
            }
            return false;
        }
        if (w & H_NOP) return false;
        replay = true; skip_to = pos + 1;   // slow header (nothing of this node has been touched yet)
        if (STATS) n_cause[3]++;
        return true;
    };

    // The general step, one node at a time, outside the pipeline (restart path): any slot, D_bottom, word by word.
    // `p` = position of the node's header; returns the position behind the node (or wherever a jump leads via skip_to).
    const uint32_t *sp = nullptr;   // uniform: stream being walked (preambles or bodies), rebased to the unit
    auto slow_node = [&](uint32_t p) -> uint32_t {
        const uint32_t w = (uint32_t)__builtin_amdgcn_readfirstlane((int)sp[p]);
        hdr = w;
        const uint32_t rs = (w >> H_RSLOT_SHIFT) & 63u, ws = (w >> H_WSLOT_SHIFT) & 63u;
        if (w & H_REG) {
#pragma unroll
            for (int j = 0; j < 4; j++) dpar.v[j] = dcur.v[j];
            bpar[0] = bcur[0]; bpar[1] = bcur[1];
        } else {
            u32x4 t;   // (two loads, not one through a selected pointer: that would be a FLAT access, whose out-of-order return makes every later wait a wait for everything)
            u32x2 tb;
            if (rs >= a.lds_slots) { t = *(const glb_row *)(coldp + (uint64_t)(rs - a.lds_slots) * 512); tb = *(const glb_row2 *)(coldp + (uint64_t)(rs - a.lds_slots) * 512 + 4); }
            else { t = *lds_at(rs * 1024u + lane16); tb.x = bs0[rs & HOT_MASK]; tb.y = bs1[rs & HOT_MASK]; }
            dpar.v[0] = t.x; dpar.v[1] = t.y; dpar.v[2] = t.z; dpar.v[3] = t.w;
            bpar[0] = tb.x; bpar[1] = tb.y;
        }
        const int cold_ws = ((w & H_STORE) && ws >= a.lds_slots) ? (int)(ws - a.lds_slots) : -1;
        if (have_sinfo && sibling_test(p)) return p + 1;
        p++;
        Carry cy;
#pragma unroll
        for (int j = 0; j < 4; j++) { cy.D.v[j] = 0; cy.N.v[j] = 0; cy.C.v[j] = 0; }
        cy.B[0] = cy.B[1] = 0; cy.flushed = false;
        if (!(w & H_END)) {
            for (;;) {
                const uint32_t m = (uint32_t)__builtin_amdgcn_readfirstlane((int)sp[p]);
                const uint32_t x = __builtin_amdgcn_raw_buffer_load_b32(trsrc, lane4, ((m & 0x3FFFFFu) + CONST_ROWS) << 8, 0);
                const uint32_t mi = (m >> 22) & 3u, pi = (m >> 24) & 3u, ri = (m >> 26) & 3u;
                const uint32_t C = (x >> mi) & 0x11111111u, P = (x >> pi) & 0x11111111u, R = (x >> ri) & 0x11111111u;
                accP += P; accC += C; accN += C & ~P; accPB += P & R; accCB += C & R;
                if (TIES) accU |= ~C & 0x11111111u;
                p++;
                if (m & M_END) break;
                if (m & M_FLUSH) flush_acc(cy);
            }
        }
        node_end(p - 1, cold_ws, &cy);
        return p;
    };

    // Software pipeline over groups of 8 words, four stages deep:
    //   group g+3  stream words being fetched (8 lanes x 4 B)
    //   group g+2  "is this (tile, site) row non-reference?" bits being fetched (8-lane gather, 3 KB bitmap)
    //   group g+1  table rows in flight, one per word, into X[0..7]; a word whose row is
    //              reference-everywhere for the tile fetches the constant row of its reference base instead
    //   group g    evaluated
    // Every load is in a fixed order, which keeps the compiler's vmcnt bookkeeping exact
    // (s_waitcnt vmcnt(N) with the younger loads still in flight).
    const uint32_t *abm = a.active + (uint64_t)tile * a.active_words;
    __attribute__((address_space(3))) uint32_t *lbits = (__attribute__((address_space(3))) uint32_t *)((__attribute__((address_space(3))) char *)slots8 + a.lds_slots * 1024u);
    if (LBITS == 1 && bits_tile != tile) {   // (one coalesced copy per change of tile, in flight together with the unit's other first loads)
        for (uint32_t i = lane; i < a.active_words; i += 64u) lbits[i] = abm[i];
        bits_tile = tile;
        __syncthreads();   // (one wave per block: orders the writes before the other lanes' reads)
    }
    const uint64_t t_wave0 = STATS ? __builtin_amdgcn_s_memtime() : 0;
    const uint64_t tr_start = STATS ? __builtin_amdgcn_s_memrealtime() : 0;
    uint64_t t_restart = 0, n_restart = 0;
    body_start = 0;
    uint64_t t_pre_end = 0;
    for (int phase = 0; phase < 2; phase++) {   // 0: replay of the preamble (the root path of the unit's first node), 1: the body
        if (STATS && phase == 1) t_pre_end = __builtin_amdgcn_s_memtime();
        if (STATS) st_body = a.chunk8_body_off[c1] - a.chunk8_body_off[c0];
        sp = phase == 0 ? a.pre8 : a.stream8;
        const uint32_t begin = phase == 0 ? a.chunk8_pre_off[c0] : a.chunk8_body_off[c0];
        const uint32_t end = phase == 0 ? a.chunk8_pre_off[c0 + 1] : a.chunk8_body_off[c1];
        if (begin >= end) continue;
        const uint32_t n = end - begin;
        sp += begin;
        if (ARG || TIES) pos_base = begin;
        const uint32_t l8 = lane & (GRP - 1u);
        uint32_t lim = n;   // uniform: end of the range being walked (words behind it read as padding)
        auto load_words = [&](uint32_t off) -> uint32_t {   // words off .. off+7 in lanes 0..7 (replicated x8)
            const uint32_t i = off + l8;
            const uint32_t v = sp[i < lim ? i : lim - 1];
            return i < lim ? v : NOPW;
        };
        auto load_bits = [&](uint32_t wv) -> uint32_t {     // per lane: bitmap dword of its word's site
            const uint32_t site = (wv & H_TAG) ? 0u : (wv & 0x3FFFFFu);
            if (LBITS == 2) return 0xFFFFFFFFu;
            return LBITS == 1 ? lbits[site >> 5] : abm[site >> 5];
        };
        // Per-lane decoding of a group (lane k of every 8 holds word k): byte offset of the word's table row -- the
        // site's own row if some sample of the tile is non-reference there, else the constant row of the site's
        // reference base (headers: row 0).  The other fields of a word are extracted on the scalar side.
        auto decode = [&](uint32_t wv, uint32_t bits) -> uint32_t {
            const bool is_hdr = (wv & H_TAG) != 0;
            const bool act = !is_hdr && ((bits >> (wv & 31u)) & 1u);
            return (act ? (wv & 0x3FFFFFu) + CONST_ROWS : (is_hdr ? 0u : (wv >> 26) & 3u)) << 8;
        };
        prune = phase == 1 && can_prune;
        pre_prune = phase == 0 && can_prune && !unit_heavy && !a.no_pre_records;
        have_info = false;
        cend = phase == 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.chunk8_body_off[c0 + 1] - 1u - begin)) : 0xFFFFFFFFu;
        cend_stale = false;
        if ((prune || pre_prune) && ub_tile != tile) {   // start from what earlier waves of this tile already know
            // (one plain cached load, not four agent-scope ones that leave the die: whatever copy it finds -- the seeds at
            // worst -- was once written as the cost of a real eligible node, so it is a valid if older bound)
            const uint4 u = *(const uint4 *)ubp;
            ub1.v[0] = pk_add(u.x, 0x00010001u); ub1.v[1] = pk_add(u.y, 0x00010001u); ub1.v[2] = pk_add(u.z, 0x00010001u); ub1.v[3] = pk_add(u.w, 0x00010001u);
            ub_tile = tile;
            ub_age = 0;
        }
        // One stream word of the group being evaluated: w0v = the group's words (lane k = word k),
        // x = the lane's dword of the word's table row.  Returns true when the pipeline has to restart at
        // skip_to (chunk end that stores, pruning jump, slow header).
        auto step = [&](uint32_t w0v, int k, uint32_t x, uint32_t pos) -> bool {
            const uint32_t w = rdlane(w0v, k);
            if (w & H_TAG) {
                if (w & H_RARE) return rare_word(w, pos);
                hdr = w;
                if (w & H_REG) {
#pragma unroll
                    for (int j = 0; j < 4; j++) dpar.v[j] = dcur.v[j];
                    bpar[0] = bcur[0]; bpar[1] = bcur[1];
                } else {
                    const u32x4 t = *lds_at((w & (63u << H_RSLOT_SHIFT)) | lane16);
                    const uint32_t rsi = (w >> H_RSLOT_SHIFT) & HOT_MASK;
                    dpar.v[0] = t.x; dpar.v[1] = t.y; dpar.v[2] = t.z; dpar.v[3] = t.w;
                    bpar[0] = bs0[rsi]; bpar[1] = bs1[rsi];
                }
                if (have_sinfo && sibling_test(pos)) return true;
                if (!(w & H_END)) return false;
            } else {
                const uint32_t mi = (w >> 22) & 3u, pi = (w >> 24) & 3u, ri = (w >> 26) & 3u;
                const uint32_t C = (x >> mi) & 0x11111111u, P = (x >> pi) & 0x11111111u, R = (x >> ri) & 0x11111111u;
                accP += P; accC += C; accN += C & ~P; accPB += P & R; accCB += C & R;
                if (TIES) accU |= ~C & 0x11111111u;
                if (!(w & M_END)) return false;   // (a node with an M_FLUSH word never gets here: its header is H_SLOW)
            }
            return node_end(pos, -1, nullptr);
        };
        uint32_t off = 0;
        if (B3) pre_off = 0xFFFFFFFFu;   // (words asked for in another unit or phase are not this one's)
        if (phase == 1 && body_start) {   // the replay ended at a path node whose subtree is not needed: close the chunks in front of its end
            off = min(body_start, n);
            cend = close_empty_chunks(off, begin);   // (nothing has been walked yet: no chunk in front of `off` holds a candidate)
        }
        bool cautious = false;   // uniform: the last run was cut short by a jump inside its first group
        while (off < lim) {
            // (re)fill the pipeline at `off`
            const uint64_t t_r0 = STATS ? __builtin_amdgcn_s_memtime() : 0;
            uint32_t w0, w1, w2;
            if (B3 && pre_off == off) { w0 = pre_w0; w1 = pre_w1; w2 = pre_w2; }   // (already here: asked for beside the third bound's table loads)
            else { w0 = load_words(off); w1 = load_words(off + GRP); w2 = load_words(off + 2 * GRP); }
            if (B3) pre_off = 0xFFFFFFFFu;
            // (the decoded row offsets of the NEXT group, o1, are carried around the loop rather than its active-row bits: the
            // decode then sits at the bottom of an iteration, behind the loads it depends on and in front of nothing -- at the
            // top of the next one the compiler's wait for those bits was a wait for every load in flight, rows included)
            const uint32_t bb0 = a.refill_all_rows ? 0xFFFFFFFFu : load_bits(w0);   // (experiment: no bitmap round trip in front of the first rows)
            const uint32_t bb1 = load_bits(w1);
            const uint32_t o0 = decode(w0, bb0);
            uint32_t o1 = decode(w1, bb1);
            uint32_t X[GRP];
#pragma unroll
            for (int k = 0; k < (int)GRP; k++) X[k] = __builtin_amdgcn_raw_buffer_load_b32(trsrc, lane4, rdlane(o0, k), 0);
            skip_to = 0;
            if (STATS) { t_restart += __builtin_amdgcn_s_memtime() - t_r0 + (X[0] & 0u); n_restart++; run_nodes = 0; }
            bool first = true;   // uniform: still inside the first group of this run
            bool hit = false;    // uniform: a step asked for a restart
            // (Tried in round 4 and removed: taking a short pruning jump -- a third of all jumps lead to a word that is already in
            // registers -- without a restart, by overwriting the skipped words with padding.  213k of 690k restarts per launch went
            // away and the launch did not get faster: a short jump's restart hits the lines its words came from, it was never one of
            // the expensive ones, and the extra code in the unrolled steps cost 6 % by itself.)
            if (cautious) {
                // Sparse regime (runs of a few words between jumps): evaluate the first group before
                // anything is requested for the second one, so a jump does not leave eight dead row loads
                // ahead of the next refill in the in-order return queue.
#pragma unroll
                for (int k = 0; k < (int)GRP; k++) {
                    if (step(w0, k, X[k], off + k)) { hit = true; break; }
                }
                if (!hit) {   // the run goes on: bring the pipeline to its steady state one group further (words and bits before the rows)
                    const uint32_t w3 = load_words(off + 3 * GRP);
                    const uint32_t b2 = load_bits(w2);
#pragma unroll
                    for (int k = 0; k < (int)GRP; k++) X[k] = __builtin_amdgcn_raw_buffer_load_b32(trsrc, lane4, rdlane(o1, k), 0);
                    o1 = decode(w2, b2);
                    w0 = w1; w1 = w2; w2 = w3;
                    off += GRP;
                    first = false;
                }
            }
            if (!hit)
            for (; off < lim; off += GRP) {
                const uint32_t w3 = load_words(off + 3 * GRP);
                const uint32_t b2 = load_bits(w2);
#pragma unroll
                for (int k = 0; k < (int)GRP; k++) {
                    if (step(w0, k, X[k], off + k)) { hit = true; break; }
                    X[k] = __builtin_amdgcn_raw_buffer_load_b32(trsrc, lane4, rdlane(o1, k), 0);
                }
                if (hit) break;
                o1 = decode(w2, b2);
                w0 = w1; w1 = w2; w2 = w3;
                first = false;
            }
            if (!hit) break;   // walked to the end of the range
            cautious = first;
            if (replay) {
                // the node whose header sits at skip_to - 1 needs the general step
                replay = false;
                const uint32_t p = skip_to - 1u;
                skip_to = 0;
                const uint32_t q = slow_node(p);
                __builtin_amdgcn_s_waitcnt(0);   // (a cold slot written here may be read by the very next node)
                if (!skip_to) { off = q; continue; }
            }
            // (also behind the general step: a node walked by slow_node may have asked too)
            if (B3 && (skip_to & 0x80000000u)) {
                // Third bound for the record in `info` (its node's D / B are still in dcur / bcur): descendants in the words behind
                // p = the node's last word.  M = the maximum of cum_over over their blocks; U = cum_under of the node's own block; hU <= M - U.
                skip_to &= 0x7FFFFFFFu;
                const uint32_t p = skip_to - 1u, J = info & INFO_JUMP_MASK;
                const uint32_t P = begin + p;
                const B3Dev *b3 = a.b3;
                uint32_t q0 = (P + 1u) >> B3_BLOCK_SHIFT, q1 = (P + J) >> B3_BLOCK_SHIFT;
                const uint32_t nb = b3->n_blocks;
                // The refill's first round trip -- the stream words -- does not wait for the answer: both places it can start from are
                // requested now, beside the table loads (a question then costs three round trips, like any restart, not four).
                const uint32_t sa0 = load_words(p + 1u), sa1 = load_words(p + 1u + GRP), sa2 = load_words(p + 1u + 2 * GRP);
                const uint32_t sb0 = load_words(p + 1u + J), sb1 = load_words(p + 1u + J + GRP), sb2 = load_words(p + 1u + J + 2 * GRP);
                // exact maximum over [q0, q1]: at each level the (up to 63) entries left and right of the whole 64-groups, the groups
                // themselves one level up -- at most seven loads per lane, all in flight together
                uint32_t mv = 0;
                {
                    const uint8_t *arr = b3->over + (uint64_t)tile * nb;
#pragma unroll
                    for (int lv = 0; lv < 4; lv++) {
                        if (lv == 3 || q1 - q0 < 64u) { for (uint32_t i = q0 + lane; i <= q1; i += 64u) mv = max(mv, (uint32_t)arr[i]); break; }
                        const uint32_t le = q0 | 63u, rs = q1 & ~63u;
                        if (q0 + lane <= le) mv = max(mv, (uint32_t)arr[q0 + lane]);
                        if (rs + lane <= q1) mv = max(mv, (uint32_t)arr[rs + lane]);
                        q0 = (q0 >> 6) + 1u; q1 = (q1 >> 6);
                        if (q0 >= q1) break;   // (no whole group between the two ends)
                        q1 -= 1u;
                        arr = lv == 0 ? b3->l1 + (uint64_t)tile * b3->n_l1 : lv == 1 ? b3->l2 + (uint64_t)tile * b3->n_l2 : b3->l3 + (uint64_t)tile * b3->n_l3;
                    }
                }
                const uint32_t un = b3->under[(uint64_t)tile * nb + (P >> B3_BLOCK_SHIFT)];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) mv = max(mv, (uint32_t)__shfl_xor((int)mv, o));
                const uint32_t M = (uint32_t)__builtin_amdgcn_readfirstlane((int)mv), U = (uint32_t)__builtin_amdgcn_readfirstlane((int)un);
                const uint32_t hs3 = (info >> INFO_HS_SHIFT) & 0x7Fu, hr3 = (info >> INFO_HR_SHIFT) & 7u;
                const uint32_t hu = M > U ? M - U : 0u;
                const uint32_t jb = STATS ? min(7u, (uint32_t)max(0, 28 - (int)__builtin_clz(J | 1u))) : 0u;   // (J < 16, < 32, ... >= 1024)
                const uint32_t gb = STATS ? min(hs3 - hr3, 4u) - 1u : 0u;   // (hsub - hsec of the record: 1, 2, 3, 4 and more)
                if (STATS) { n_b3[0]++; if (lane == 0) { atomicAdd((unsigned long long *)a.stats + 72 + jb, 1ull); atomicAdd((unsigned long long *)a.stats + 88 + gb, 1ull); } }
                pre_off = p + 1u; pre_w0 = sa0; pre_w1 = sa1; pre_w2 = sa2;
                if (M != 255u && hu + hr3 < hs3) {   // (255: some block's count is saturated -- no bound)
                    const uint32_t rec3 = (info & ~(0x7Fu << INFO_HS_SHIFT)) | ((hu + hr3) << INFO_HS_SHIFT);
                    if (all_far(dcur, bcur, rec3)) {
                        skip_to = p + 1u + J;
                        pre_off = skip_to; pre_w0 = sb0; pre_w1 = sb1; pre_w2 = sb2;
                        if (STATS) { n_b3[1]++; n_skipped += min(J, st_body > p + 1u ? st_body - (p + 1u) : 0u); count_jump(J, 0); if (lane == 0) { atomicAdd((unsigned long long *)a.stats + 80 + jb, 1ull); atomicAdd((unsigned long long *)a.stats + 92 + gb, 1ull); } }
                    }
                }
            }
            if (TIES) tie_event();   // (a node that tied asked for this restart, or did so on top of a jump)
            // restart request: close every chunk whose end marker lies before the new position
            // (the end-marker position of the open chunk is kept in a register: loading it here put a memory round
            // trip, behind every load still in flight, in front of each refill)
            if (cend_stale) {
                cend_stale = false;
                cend = chunk < c1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(a.chunk8_body_off[chunk + 1] - 1u - begin)) : 0xFFFFFFFFu;
            }
            if (chunk < c1 && skip_to > cend) {   // the open chunk ends (it may hold a candidate); those behind it, up to the new position, are empty
                chunk_end();
                cend = close_empty_chunks(skip_to, begin);
            }
            // Units are cut while they run: how long one takes is known only by walking it (most of a far unit is jumped over,
            // the part next to the tile's samples is not), so they start out large, and a wave that has been on one for
            // a.split_cycles while others wait for work (tickets taken minus entries pushed, read now and then, never waited for) hands the
            // second half of what is left to the shared list.  That half starts at a chunk boundary: every chunk has a preamble.
            const uint32_t stop = (uint32_t)__builtin_amdgcn_readfirstlane((int)stop_v);
            if (chunk >= stop) break;   // (the rest of the unit belongs to another wave now)
            if (phase == 1 && stop - chunk >= 2u && (uint32_t)__builtin_amdgcn_s_memtime() - t_mark > (unit_heavy ? a.split_heavy : a.split_cycles)) {
                // how many waves wait for an entry: tickets taken - entries pushed (one look per threshold, and only from
                // units that have been running that long)
                // Only a unit that is getting on slowly is worth cutting: one that has closed many chunks since the last look is
                // jumping over most of what it was given -- its second half would cost the other wave a replay of the root path
                // to find out the same.
                const unsigned long long pr = __hip_atomic_load((const unsigned long long *)(a.dyn_ctl + DYN_HEAD), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const uint32_t hd = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)pr), tl = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(pr >> 32));
                const uint32_t adv = chunk - (uint32_t)__builtin_amdgcn_readfirstlane((int)mark_v);
                asm volatile("v_mov_b32 %0, %1" : "=v"(mark_v) : "s"(chunk));
                if (hd > tl && adv <= a.split_dense) {
                    // As many pieces as waves wait (round 4; at most 63, each at least one chunk, the wave keeps the first): the unit
                    // trace of a launch showed half of the waves idle from 60 % of its span on while a handful of dense units were
                    // halved once per look -- a unit with 500 us of work left reached the idle waves in log2 steps of 170 us.  Lane i
                    // writes entry i.
                    const uint32_t R = stop - chunk;
                    // (a.split_many = the fewest chunks a piece may have: a one-chunk piece costs its wave 105 us -- the fixed part of a
                    // unit and a walk without the bounds its neighbours found -- where the same chunk takes 47 us inside its unit)
                    // (the units of a tile's own region are dense -- a chunk there is 50-130 us of walking, as much as a piece's fixed
                    // part -- and short: they have their own, smaller minimum, the high half of the argument)
                    const uint32_t sm = unit_heavy ? (a.split_many >> 16) : (a.split_many & 0xFFFFu);
                    const uint32_t cap_p = R / max(sm, 1u) > 1u ? R / max(sm, 1u) - 1u : 1u;
                    const uint32_t pieces = a.split_many ? min(min(hd - tl, 63u), cap_p) : 1u;
                    const uint32_t per = a.split_many ? max(1u, R / (pieces + 1u)) : R / 2u;   // (one piece: the second half, rounded down)
                    uint32_t slot = 0xFFFFFFFFu;
                    if (lane == 0) {
                        atomicAdd(a.dyn_ctl + DYN_ACTIVE, pieces);   // the entries count as live work from before they can be seen
                        slot = atomicAdd(a.dyn_ctl + DYN_TAIL, pieces);
                    }
                    slot = (uint32_t)__builtin_amdgcn_readfirstlane((int)slot);
                    const uint32_t n_ok = slot < a.dyn_cap ? min(pieces, a.dyn_cap - slot) : 0u;   // (list full: the rest is not pushed)
                    if (n_ok) {
                        if (lane < n_ok) {   // piece i = the i-th run of `per` chunks counted back from the unit's end
                            const uint32_t pc1 = stop - lane * per, pc0 = pc1 - per;
                            const unsigned long long ev = (unsigned long long)pc0 | ((unsigned long long)pc1 << 20) | ((unsigned long long)tile << 40) |
                                                          ((unsigned long long)(unit_heavy ? 1u : 0u) << 52) | ((unsigned long long)a.dyn_epoch << 53);
                            __hip_atomic_store(a.dyn_units + slot + lane, ev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        const uint32_t new_stop = stop - n_ok * per;
                        asm volatile("v_mov_b32 %0, %1" : "=v"(stop_v) : "s"(new_stop));
                        if (STATS) n_split += n_ok;
                    }
                    if (n_ok < pieces && lane == 0) atomicAdd(a.dyn_ctl + DYN_ACTIVE, 0u - (pieces - n_ok));
                }
                t_mark = (uint32_t)__builtin_amdgcn_s_memtime();
            }
            off = skip_to;
        }
    }
    if (STATS && a.trace && lane == 0) {   // one record per unit: who ran what, when (UGP_TRACE; tools/analysis/unit_trace.py)
        const unsigned long long i = atomicAdd((unsigned long long *)a.trace, 1ull);
        if (i < a.trace_cap) {
            unsigned long long *r = (unsigned long long *)a.trace + 8 + i * 6;
            r[0] = ((unsigned long long)blockIdx.x << 32) | (tile << 4) | (unit_heavy ? 1u : 0u) | ((uflags & 4u) ? 2u : 0u);
            r[1] = ((unsigned long long)c0 << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)stop_v);
            r[2] = tr_pull; r[3] = tr_start; r[4] = __builtin_amdgcn_s_memrealtime();
            r[5] = ((unsigned long long)n_restart << 32) | ((unsigned long long)(n_split & 0xFFu) << 24) | (c1 & 0xFFFFFFu);
        }
    }
    if (STATS && !a.trace && lane == 0) {   // debug accounting, one update per unit
        const unsigned long long tw = __builtin_amdgcn_s_memtime() - t_wave0;
        unsigned long long *st = (unsigned long long *)a.stats;
        atomicAdd(st + 0, (unsigned long long)n_skipped);
        atomicAdd(st + 26, (unsigned long long)n_first_skip);
        atomicAdd(st + 1, (unsigned long long)n_restart);
        atomicAdd(st + 2, (unsigned long long)t_restart);
        atomicAdd(st + 3, tw);
        atomicAdd(st + (unit_heavy ? 27 : 28), tw);   // wave cycles inside / outside the tiles' own regions
        atomicAdd(st + (unit_heavy ? 29 : 30), 1ull);
        atomicAdd(st + 31, (unsigned long long)n_split);
        for (int i = 0; i < 4; i++) atomicAdd(st + 48 + 2 * i + (unit_heavy ? 0 : 1), (unsigned long long)n_cause[i]);
        for (int i = 0; i < 8; i++) atomicAdd(st + 56 + i, (unsigned long long)n_jlen[i]);
        if (B3) { atomicAdd(st + 64, (unsigned long long)n_b3[0]); atomicAdd(st + 65, (unsigned long long)n_b3[1]); }
        atomicAdd(st + 66, (unsigned long long)n_eval[0]); atomicAdd(st + 67, (unsigned long long)n_eval[1]);
        if (!unit_heavy) {   // what the preamble records decided for this unit
            const uint32_t body_words = a.chunk8_body_off[c1] - a.chunk8_body_off[c0];
            const int cls = body_start >= body_words ? 0 : (body_start ? 1 : 2);
            atomicAdd(st + 32 + 2 * cls, 1ull);
            atomicAdd(st + 33 + 2 * cls, tw);
            if (cls == 0) {   // a unit that ended in its preamble: where did its time go?
                atomicAdd(st + 38, (unsigned long long)(t_wave0 - t_pull0));      // pulling the unit
                atomicAdd(st + 39, (unsigned long long)(t_pre_end - t_wave0));    // replaying the preamble
                atomicAdd(st + 40, (unsigned long long)(t_wave0 + tw - t_pre_end));   // closing the chunks
            }
        }
        atomicMax(st + 4, tw);
        atomicAdd(st + 5 + min(tw >> 22, 15ull), 1ull);   // histogram of unit durations, 4.2M-cycle bins
    }
    }   // next work unit
}

// The coarse pass needs no phase 2: per sample the minimum over the recorded chunks (ineligible nodes carry bit 15 and lose
// against any eligible one; the root always is), the smallest chunk that attains it, and the node that set that chunk's minimum --
// k_best8<ARG> stored the low 16 bits of its stream position (chunks are shorter than 2^16 words), node_pos8 maps a position back
// to the node.  Any node of minimal cost will do: the result seeds the sort and the descent, never an answer.
// Block = one 512-sample tile, thread = one dword of its 1 KB records (two samples).
__global__ void k_coarse_result(const uint32_t *__restrict__ lbest, const uint32_t *__restrict__ lpos, const uint32_t *__restrict__ list,
                                const uint32_t *__restrict__ list_n, uint32_t n_chunks, uint32_t n_tiles, uint32_t n_queries,
                                const uint32_t *__restrict__ chunk_node_off, const uint32_t *__restrict__ chunk8_body_off,
                                const uint32_t *__restrict__ node_pos8, const uint32_t *__restrict__ dfs2bfs, ugp_result *__restrict__ out) {
    const uint32_t tile = blockIdx.x, i = tile * 256 + threadIdx.x;
    const uint32_t per_chunk = n_tiles * 256;
    const uint32_t n = list_n[tile];
    const uint32_t *l = list + (uint64_t)tile * n_chunks;
    uint32_t mlo = 0xFFFFu, mhi = 0xFFFFu, clo = 0, chi = 0;
    for (uint32_t e = 0; e < n; e++) {
        const uint32_t c = l[e];
        const uint32_t v = lbest[(uint64_t)c * per_chunk + i];
        const uint32_t lo = v & 0xFFFFu, hi = v >> 16;
        if (lo < mlo || (lo == mlo && c < clo)) { mlo = lo; clo = c; }
        if (hi < mhi || (hi == mhi && c < chi)) { mhi = hi; chi = c; }
    }
    auto node_of = [&](uint32_t c, uint32_t p16) -> uint32_t {   // DFS index of the node of chunk c whose words hold the position
        const uint32_t start = chunk8_body_off[c];
        const uint32_t pos = start + ((p16 - start) & 0xFFFFu);
        uint32_t lo = chunk_node_off[c], hi = chunk_node_off[c + 1];   // the last d in [lo, hi) with node_pos8[d] <= pos
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) / 2; if (node_pos8[mid] <= pos) lo = mid; else hi = mid; }
        return lo;
    };
    const uint32_t plo = lpos[(uint64_t)clo * per_chunk + i] & 0xFFFFu, phi = lpos[(uint64_t)chi * per_chunk + i] >> 16;
    // dword (lane l, j): samples 8l + j and 8l + j + 4 of the tile
    const uint32_t l64 = threadIdx.x >> 2, j = threadIdx.x & 3u;
    const uint32_t q0 = tile * 512 + l64 * 8 + j;
    if (q0 < n_queries) out[q0] = ugp_result{(int32_t)mlo, 1u, dfs2bfs[node_of(clo, plo)], 0u};
    if (q0 + 4 < n_queries) out[q0 + 4] = ugp_result{(int32_t)mhi, 1u, dfs2bfs[node_of(chi, phi)], 0u};
}

// Global minimum per sample over the chunk-local minima that were recorded: the list of a tile is cut into
// gridDim.y slices (partial minima in `part`), then k_gbest2 folds the slices.  Block = (tile, slice), thread =
// one dword of the tile's 1 KB record.
__global__ void k_gbest(const uint32_t *__restrict__ lbest, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n,
                        uint32_t n_chunks, uint32_t n_tiles, uint32_t *__restrict__ part /* [gridDim.y][n_tiles*256] */,
                        uint32_t *__restrict__ part_cnt /* or null: in how many chunks of the slice the slice minimum is attained, per half, capped at 3 */) {
    // (one-wave blocks, four per tile: see k_descend -- a CU next to the other batches' walks has room for one more wave)
    const uint32_t tile = blockIdx.x >> 2, i = blockIdx.x * 64 + threadIdx.x;
    const uint32_t per_chunk = n_tiles * 256;
    const uint32_t n = list_n[tile];
    const uint32_t e0 = (uint32_t)((uint64_t)blockIdx.y * n / gridDim.y), e1 = (uint32_t)((uint64_t)(blockIdx.y + 1) * n / gridDim.y);
    const uint32_t *l = list + (uint64_t)tile * n_chunks;
    if (!part_cnt) {
        uint32_t m = 0xFFFFFFFFu;
        for (uint32_t e = e0; e < e1; e++) m = pk_min(m, lbest[(uint64_t)l[e] * per_chunk + i]);
        part[(uint64_t)blockIdx.y * per_chunk + i] = m;
        return;
    }
    uint32_t ml = 0xFFFFu, mh = 0xFFFFu, cl = 0, ch = 0;
    for (uint32_t e = e0; e < e1; e++) {
        const uint32_t x = lbest[(uint64_t)l[e] * per_chunk + i], xl = x & 0xFFFFu, xh = x >> 16;
        cl = xl < ml ? 1u : (xl == ml ? min(cl + 1u, 3u) : cl); ml = min(ml, xl);
        ch = xh < mh ? 1u : (xh == mh ? min(ch + 1u, 3u) : ch); mh = min(mh, xh);
    }
    part[(uint64_t)blockIdx.y * per_chunk + i] = ml | (mh << 16);
    part_cnt[(uint64_t)blockIdx.y * per_chunk + i] = cl | (ch << 16);
}
__global__ void k_gbest2(const uint32_t *__restrict__ part, uint32_t n_slices, uint32_t per_chunk,
                         uint32_t *__restrict__ gbest /* [n_tiles][64][4] packed */, const uint32_t *__restrict__ part_cnt, uint32_t *__restrict__ gcnt /* both or neither */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per_chunk) return;
    // (eight independent loads in flight per thread: one after the other the 64 slices were 64 round trips, 21 us of a batch's chain)
    if (!gcnt) {
        uint32_t m8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) m8[j] = 0xFFFFFFFFu;
        uint32_t k = 0;
        for (; k + 8 <= n_slices; k += 8) {
#pragma unroll
            for (int j = 0; j < 8; j++) m8[j] = pk_min(m8[j], part[(uint64_t)(k + j) * per_chunk + i]);
        }
        uint32_t m = 0xFFFFFFFFu;
        for (; k < n_slices; k++) m = pk_min(m, part[(uint64_t)k * per_chunk + i]);
#pragma unroll
        for (int j = 0; j < 8; j++) m = pk_min(m, m8[j]);
        gbest[i] = m;
        return;
    }
    uint32_t ml = 0xFFFFu, mh = 0xFFFFu, cl = 0, ch = 0;
    auto fold = [&](uint32_t x, uint32_t c) {
        const uint32_t xl = x & 0xFFFFu, xh = x >> 16, kl = c & 0xFFFFu, kh = c >> 16;
        cl = xl < ml ? kl : (xl == ml ? min(cl + kl, 3u) : cl); ml = min(ml, xl);
        ch = xh < mh ? kh : (xh == mh ? min(ch + kh, 3u) : ch); mh = min(mh, xh);
    };
    uint32_t k = 0;
    for (; k + 8 <= n_slices; k += 8) {
        uint32_t x8[8], c8[8];
#pragma unroll
        for (int j = 0; j < 8; j++) { x8[j] = part[(uint64_t)(k + j) * per_chunk + i]; c8[j] = part_cnt[(uint64_t)(k + j) * per_chunk + i]; }
#pragma unroll
        for (int j = 0; j < 8; j++) fold(x8[j], c8[j]);
    }
    for (; k < n_slices; k++) fold(part[(uint64_t)k * per_chunk + i], part_cnt[(uint64_t)k * per_chunk + i]);
    gbest[i] = ml | (mh << 16);
    gcnt[i] = cl | (ch << 16);
}

__device__ __forceinline__ uint32_t pk_lookup(const uint32_t *packed, uint32_t tile512, uint32_t within) {
    // sample `within` (0..511) of a tile: lane = within >> 3, nibble j = within & 7 ->
    // dword (j & 3), half (j >> 2)
    const uint32_t l = within >> 3, j = within & 7u;
    const uint32_t w = packed[((uint64_t)tile512 * 64 + l) * 4 + (j & 3u)];
    return (j >> 2) ? (w >> 16) : (w & 0xFFFFu);
}

// One thread per (chunk, 64-sample tile): does any of its samples attain its
// global minimum in this chunk?  If so the pair becomes a phase-2 work item.
__global__ void k_select(const uint32_t *__restrict__ lbest, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n,
                         const uint32_t *__restrict__ gbest, uint32_t n_chunks, uint32_t n_tiles, uint32_t n_queries,
                         uint32_t *__restrict__ items, uint32_t *__restrict__ n_items, uint32_t cap,
                         // (round 6; all three or none) A sample needs no walk of this chunk when its global minimum is attained in this ONE
                         // chunk (gcnt == 1), by ONE node of it (its bit in luniq clear: k_best8's second-smallest cost), and the seed descent
                         // names a node of exactly that cost (dres == gbest): that node is then the only optimal placement (k_final).
                         const uint32_t *__restrict__ gcnt, const uint32_t *__restrict__ luniq, const uint32_t *__restrict__ dres) {
    // blocks (tile, 0..gridDim.y-1) share one tile; their threads stride over its (recorded chunk, 64-sample sub-tile) pairs
    const uint32_t tile = blockIdx.x;
    const uint32_t n = list_n[tile], n_t64 = n_tiles * 8;
    const uint32_t *l = list + (uint64_t)tile * n_chunks;
    for (uint32_t p = blockIdx.y * blockDim.x + threadIdx.x; p < n * 8u; p += gridDim.y * blockDim.x) {
        const uint32_t c = l[p >> 3], t8 = p & 7u;
        const uint32_t t64 = tile * 8 + t8;
        if ((uint64_t)t64 * 64 >= n_queries) continue;
        // the 64 samples of t64 are lanes (t64&7)*8 .. +8 of tile t64>>3: 32 consecutive dwords
        const uint64_t off = ((uint64_t)tile * 64 + t8 * 8) * 4;
        const uint4 *lb = (const uint4 *)(lbest + (uint64_t)c * n_tiles * 256 + off);   // (16-byte loads, all sixteen in flight together)
        const uint4 *gb = (const uint4 *)(gbest + off);
        bool hit = false;
        if (gcnt) {
            // the sub-tile's eight lanes' bytes of the record's "more than one node at the chunk minimum" flags (bit j + 4h: sample j + 4h of the lane)
            const uint2 nu2 = *(const uint2 *)((const uint8_t *)luniq + ((uint64_t)c * n_tiles + tile) * 64 + t8 * 8u);
            const uint4 *gc = (const uint4 *)(gcnt + off), *dr = (const uint4 *)(dres + off);
#pragma unroll 4
            for (int v = 0; v < 8; v++) {   // v = lane within the sub-tile; dword w = samples w (low half) and w + 4 (high half) of that lane (four lanes' loads in flight together)
                const uint4 l4 = lb[v], g4 = gb[v], c4 = gc[v], d4 = dr[v];
                const uint32_t nu = ((v < 4 ? nu2.x : nu2.y) >> (8 * (v & 3))) & 0xFFu;
                const uint32_t ls[4] = {l4.x, l4.y, l4.z, l4.w}, gs[4] = {g4.x, g4.y, g4.z, g4.w};
                const uint32_t cs[4] = {c4.x, c4.y, c4.z, c4.w}, ds[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const uint32_t q0 = t64 * 64 + (uint32_t)v * 8 + (uint32_t)w;
                    const uint32_t x = ls[w] ^ gs[w];
                    const bool res_lo = (cs[w] & 0xFFFFu) == 1u && !((nu >> w) & 1u) && ((ds[w] ^ gs[w]) & 0xFFFFu) == 0;
                    const bool res_hi = (cs[w] >> 16) == 1u && !((nu >> (w + 4)) & 1u) && ((ds[w] ^ gs[w]) >> 16) == 0;
                    if ((x & 0xFFFFu) == 0 && q0 < n_queries && !res_lo) hit = true;
                    if ((x >> 16) == 0 && q0 + 4 < n_queries && !res_hi) hit = true;
                }
            }
        } else {
        uint4 xl[8], xg[8];
#pragma unroll
        for (int v = 0; v < 8; v++) { xl[v] = lb[v]; xg[v] = gb[v]; }
#pragma unroll
        for (int v = 0; v < 8; v++) {
            const uint32_t xs[4] = {xl[v].x ^ xg[v].x, xl[v].y ^ xg[v].y, xl[v].z ^ xg[v].z, xl[v].w ^ xg[v].w};
#pragma unroll
            for (int w = 0; w < 4; w++) {
                const uint32_t k = (uint32_t)(v * 4 + w), x = xs[w];
                const uint32_t q0 = t64 * 64 + (k >> 2) * 8 + (k & 3u);   // low half: nibble (k&3); high half: +4
                if ((x & 0xFFFFu) == 0 && q0 < n_queries) hit = true;
                if ((x >> 16) == 0 && q0 + 4 < n_queries) hit = true;
            }
        }
        }
        if (hit) {
            const uint32_t idx = atomicAdd(n_items, 1u);
            if (idx < cap) items[idx] = c * n_t64 + t64;
        }
    }
}

// Phase 2 on the packed path: the records (tile, chunk) in which some sample of the tile attains its global minimum become the work
// units of k_best8<TIES>, appended to the list of one of its eight queues (tile mod 8: a queue's tiles stay on one XCD).  One wave
// per record: lane l compares its four dwords (eight samples) of the record with the tile's minima.
// info: [0..7] first entry of each queue's list, [8..15] entries (zeroed before), [16..23] queue heads, [32..] the walk's counters.
__global__ void __launch_bounds__(256) k_select8(const uint32_t *__restrict__ lbest, const uint32_t *__restrict__ list, const uint32_t *__restrict__ list_n,
                                                 const uint32_t *__restrict__ gbest, uint32_t n_chunks, uint32_t n_tiles, uint4 *__restrict__ units,
                                                 uint32_t *__restrict__ info) {
    const uint32_t tile = blockIdx.x, lane = threadIdx.x & 63u;
    const uint32_t n = list_n[tile];
    const uint32_t *l = list + (uint64_t)tile * n_chunks;
    const uint4 gb = *(const uint4 *)(gbest + ((uint64_t)tile * 64 + lane) * 4);
    const uint32_t x = tile & 7u, cap_q = ((n_tiles + 7u) / 8u) * n_chunks;
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 8) info[threadIdx.x] = threadIdx.x * cap_q;
    for (uint32_t e = blockIdx.y * 4u + (threadIdx.x >> 6); e < n; e += gridDim.y * 4u) {
        const uint32_t c = l[e];
        const uint4 lb = *(const uint4 *)(lbest + (((uint64_t)c * n_tiles + tile) * 64 + lane) * 4);
        const uint32_t t = pk_min(lb.x ^ gb.x, 0x00010001u) & pk_min(lb.y ^ gb.y, 0x00010001u) & pk_min(lb.z ^ gb.z, 0x00010001u) &
                           pk_min(lb.w ^ gb.w, 0x00010001u);   // a half is 0 where some pair of halves is equal
        if (__builtin_amdgcn_ballot_w64(t != 0x00010001u) != 0 && lane == 0) {
            const uint32_t idx = atomicAdd(&info[8 + x], 1u);
            units[(uint64_t)x * cap_q + idx] = make_uint4(tile, c, c + 1u, 0u);
        }
    }
}

// Phase 2: re-walk only the selected (chunk, 64-sample tile) pairs, one sample
// per lane, counting the nodes that attain the sample's global minimum and
// keeping the reference's winner among them (usher_mapper.cpp:476-497).
template <bool LIST>
__global__ void __launch_bounds__(64) k_ties(PlaceArgs a, const uint32_t *__restrict__ lbest, const uint32_t *__restrict__ gbest,
                                             const uint32_t *__restrict__ items, const uint32_t *__restrict__ n_items,
                                             uint32_t cap, uint32_t n_t64 /* 8 * n_tiles512, as k_select encodes */,
                                             uint32_t *__restrict__ cnt_out, uint32_t *__restrict__ key_out,
                                             const uint32_t *__restrict__ rank2bfs, const uint32_t *__restrict__ order /* slot -> sample, or nullptr */) {
    extern __shared__ __attribute__((aligned(16))) uint32_t slots[];
    __shared__ __attribute__((aligned(16))) uint32_t rowbuf[64 * 8];
    const uint32_t lane = threadIdx.x;
    const uint32_t n = min(*n_items, cap);
    // Blocks go round robin over the eight XCDs, each with its own L2; k_select appends the (up to eight) 64-sample sub-tiles of one
    // (tile, chunk) record next to each other, and they read the same stream windows and 32-byte pieces of the same 256-byte table
    // rows.  So the blocks of one XCD share a contiguous eighth of the list (round 5; before: item i on XCD i mod 8).  Measured: next
    // to nothing -- 437 -> 426 MB from HBM per launch, L2 hits 8.6 -> 10.7 %: few records have relevant samples in more than one
    // sub-tile, so neighbouring items mostly name different chunks.
    const uint32_t nx = min(8u, gridDim.x), xcd = blockIdx.x % nx, seg = (n + nx - 1u) / nx, on_xcd = (gridDim.x - xcd + nx - 1u) / nx;
    const uint32_t it_end = min(n, (xcd + 1u) * seg);
    for (uint32_t it = xcd * seg + blockIdx.x / nx; it < it_end; it += on_xcd) {
        const uint32_t item = items[it];
        const uint32_t c = item / n_t64, t64 = item % n_t64;
        const uint32_t q = t64 * 64 + lane;
        const uint32_t want = (q < a.n_queries) ? pk_lookup(gbest, q >> 9, q & 511u) : 0xFFFFFFFFu;
        const bool relevant = q < a.n_queries && pk_lookup(lbest + (uint64_t)c * (n_t64 / 8u) * 256u, q >> 9, q & 511u) == want;
        const uint32_t list_q = (LIST && q < a.n_queries) ? (order ? order[q] : q) : 0u;
        const uint32_t skipn = (a.skip && q < a.n_queries) ? a.skip[order ? order[q] : q] : 0xFFFFFFFFu;   // (extended search: one node left out for this sample)
        WalkOut r = walk_ties<LIST>(a, slots, rowbuf, t64, c, lane, want, relevant, rank2bfs, list_q, skipn);
        if (r.cnt) {
            atomicAdd(&cnt_out[q], r.cnt);
            atomicMax(&key_out[q], r.key);
        }
    }
}

// An extended search that leaves one node x out for a sample (ugp_place_opts::skip_node: matUtils uncertainty never maps a sample
// onto its own node, uncertainty.cpp:216) on the packed path: phase 1 runs as always -- with bounds that never relied on x (seeds
// that skip it, no tightening from the chunk minima, which do include it: Best8Args::freeze_ub) -- and the ONE chunk minimum that
// may be x's is then recomputed for that sample without x: the chunk that holds x, walked one sample per lane like phase 2.
// Phase 2 then sees minima of the search it was asked for.  One wave per sample slot; only the slot's own lane stores (a 16-bit
// half of the packed record).  A record of a chunk that phase 1 did not list is written too and never read.
__global__ void __launch_bounds__(64) k_fix_skip(PlaceArgs a, uint32_t *__restrict__ lbest, const uint32_t *__restrict__ skip_chunk, uint32_t n_tiles512,
                                                  const uint32_t *__restrict__ rank2bfs, const uint32_t *__restrict__ order) {
    extern __shared__ __attribute__((aligned(16))) uint32_t slots[];
    __shared__ __attribute__((aligned(16))) uint32_t rowbuf[64 * 8];
    const uint32_t slot = blockIdx.x, lane = threadIdx.x;
    const uint32_t qi = order ? order[slot] : slot;
    const uint32_t x = a.skip[qi], c = skip_chunk[qi];   // uniform
    if (x == 0xFFFFFFFFu || c == 0xFFFFFFFFu) return;
    const WalkOut r = walk_ties<false, true>(a, slots, rowbuf, slot >> 6, c, lane, 0u, true, rank2bfs, 0u, x);   // (the other lanes walk along for their own samples: unused)
    if (lane == (slot & 63u)) {
        const uint32_t within = slot & 511u, l = within >> 3, j = within & 7u;
        uint16_t *rec = (uint16_t *)(lbest + (((uint64_t)c * n_tiles512 + (slot >> 9)) * 64 + l) * 4 + (j & 3u));
        rec[j >> 2] = (uint16_t)r.best;
    }
}

__global__ void k_final(const uint32_t *__restrict__ gbest, const uint32_t *__restrict__ cnt, const uint32_t *__restrict__ key,
                        const uint32_t *__restrict__ rank2bfs, uint32_t n_queries, ugp_result *__restrict__ out,
                        const uint32_t *__restrict__ order /* slot -> sample, or nullptr */,
                        const uint32_t *__restrict__ dnode, const uint32_t *__restrict__ refined /* both or neither: the seed descent's node and its cost, by slot */) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n_queries) return;
    ugp_result r;
    r.best_set_difference = (int32_t)pk_lookup(gbest, q >> 9, q & 511u);
    r.num_best = cnt[q];
    if (dnode && r.num_best == 0) {
        // no chunk was walked for this sample: k_select found its minimum attained once, by one node, at the cost of the descent's node.
        // (The condition is checked again here; a sample that fails it keeps num_best = 0, which no caller can mistake for an answer.)
        const uint32_t dn = dnode[q];
        if (dn != 0xFFFFFFFFu && refined[q] == (uint32_t)r.best_set_difference) { r.num_best = 1u; r.best_j = dn >> 1; r.best_has_unique = dn & 1u; }
        else { r.best_j = 0xFFFFFFFFu; r.best_has_unique = 0u; }
        out[order ? order[q] : q] = r;
        return;
    }
    r.best_j = rank2bfs[key[q] >> 1];
    r.best_has_unique = key[q] & 1u;
    out[order ? order[q] : q] = r;
}

// Locality sort of the samples (speed only): key = DFS rank of the sample's best node in a coarse
// top-of-the-tree MAT.  Samples that land in the same region of the tree share tiles, so the
// lower-bound pruning of k_best8 can skip the rest of the tree for the whole tile.
__global__ void k_sort_keys(const ugp_result *__restrict__ coarse_res, const uint32_t *__restrict__ coarse2dfs, uint32_t n,
                            uint32_t *__restrict__ keys, uint32_t *__restrict__ idx) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    keys[q] = coarse2dfs[coarse_res[q].best_j];
    idx[q] = q;
}
__global__ void k_invert(const uint32_t *__restrict__ order, uint32_t n, uint32_t *__restrict__ slot_of) {
    const uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n) slot_of[order[s]] = s;
}

// Seed the per-sample upper bounds of k_best8 with the samples' best cost in the coarse MAT.  The coarse
// MAT is the top of the tree with every kept node's own mutations, so a node that is eligible there is
// eligible in the full tree with the same cost (a kept node that is a leaf only in the coarse tree is
// eligible there iff common > 0, which makes an internal node eligible too): the value is a real cost.
// The unused sample slots of the last tile must not hold the tile back (all 512 lanes have to be far for a jump): left
// alone they are 240 copies of a sample identical to the reference, an outlier that sits at the root.  They get D(bottom) =
// pad_d (a count no real cost undercuts within 16 bits, and larger than any hsub) and the bound 0: the first test, D - hsub > 0,
// then holds for them everywhere.  Their results are never read.
__global__ void k_seed_ub(const ugp_result *__restrict__ coarse_res, const uint32_t *__restrict__ order, uint32_t n_queries,
                          uint32_t n_words, uint32_t *__restrict__ ub, const uint32_t *__restrict__ refined, uint32_t *__restrict__ dbottom,
                          uint32_t pad_d, const uint32_t *__restrict__ skip, const uint32_t *__restrict__ coarse2bfs,
                          const uint32_t *__restrict__ dnode, uint32_t *__restrict__ dres /* both or neither: packed like ub -- refined[slot] where the descent names a node of that cost, else 0xFFFF */) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;   // word (tile*64 + lane)*4 + jj holds samples jj and jj+4 of the lane
    if (i >= n_words) return;
    const uint32_t slot = (i >> 2) * 8 + (i & 3u);
    if (dres) {
        auto dv = [&](uint32_t q) -> uint32_t { return (q < n_queries && refined && dnode[q] != 0xFFFFFFFFu) ? min(refined[q], 0xFFFFu) : 0xFFFFu; };
        dres[i] = dv(slot) | (dv(slot + 4) << 16);
    }
    auto val = [&](uint32_t q) -> uint32_t {
        if (q >= n_queries) { if (dbottom) dbottom[q] = pad_d; return dbottom ? 0u : 0x7F7Fu; }
        const int32_t b = coarse_res[order[q]].best_set_difference;
        uint32_t v = b < 0 || b > 0x7F7F ? 0x7F7Fu : (uint32_t)b;
        // (a search that leaves one node out for this sample, ugp_place_opts::skip_node: the coarse best cost is a bound only when
        // some other node attains it -- unknown when the coarse winner is that very node)
        // (without the coarse -> caller index map nothing is known about the winner: no seed for such a sample)
        if (skip && skip[order[q]] != 0xFFFFFFFFu && coarse_res[order[q]].best_j != 0xFFFFFFFFu &&
            (!coarse2bfs || coarse2bfs[coarse_res[order[q]].best_j] == skip[order[q]])) v = 0x7F7Fu;
        return refined ? min(v, refined[q]) : v;   // (the descent below: also the cost of a real eligible node)
    };
    ub[i] = val(slot) | (val(slot + 4) << 16);
}

// Tighter seeds: a best-first descent from the sample's best node in the coarse MAT.  The coarse MAT ends ~1,000 nodes above
// the leaves, so its best cost still counts every mutation the sample shares with the rest of its lineage (8 on average, up
// to 20, at 10M nodes); with bounds that loose the second pruning bound and the preamble records of k_best8 decide late
// (seeded with the exact answers the kernel is 40 % faster).  One wave per sample: a small frontier of (node, D) pairs lives
// in LDS; the entry with the smallest D is expanded -- the lanes evaluate its children (64 per round) in the full tree:
// D(child), cost(child), eligibility, exactly as the walk kernels do -- the smallest eligible cost seen is kept, and every
// internal child whose D does not exceed its parent's joins the frontier.  Along the sample's own lineage D never grows
// (every mutation there is one the sample has), siblings that merely tie (no mutations) are dead ends one level further
// down, so the frontier stays tiny.  Every value recorded is the cost of a real eligible node: the result is a valid upper
// bound of best(s) whatever the search misses; it only has to be good, not exact.
// The search starts at the coarse best node AND at its nearest ancestors: in a tree with much homoplasy the coarse winner is
// now and then (4 % of the benchmark's samples) a sibling branch that happens to share one mutation with the sample, one or
// two levels below the point where the sample's own lineage leaves the coarse MAT.
// Needs the children of a node to be contiguous in BFS index (child_begin[j] + 1 .. child_begin[j + 1]).
constexpr uint32_t DESC_FRONTIER = 32, DESC_MAX_EXPANSIONS = 96, DESC_UP = 2;
// node_pair[j] = {child_begin[j], rec_off[j]} (one 8-byte load per node; entry N closes the last child range).
// The search is a chain of dependent loads (children range -> record -> table rows), a few per expansion: the frontier
// keeps each entry's children range so that an expansion starts with the children's pairs, a record's first words are
// fetched together before its length is known (the stream is padded), and the rows of a node's mutations are in flight together.
// A sample is served by G lanes: 16 (four samples per wave: a quarter of the waves, all resident at once, for the same
// chain per sample) when nodes rarely have more children than that, a whole wave when the tree has large polytomies
// (the SARS-CoV-2-shaped benchmark tree: 16 lanes cost 2.5x there).
// (Round 5, tried and removed: a second table {first record word, first mutation word} per node, loaded with the pair, so that the row
// of a one-mutation node is requested in the second round trip instead of the third: 8 more bytes per node, k_descend 242 -> 245 us --
// the chain of an expansion is not what bounds it; the four samples of a wave wait for the slowest, and the rounds of children do.)
// (one-wave blocks since round 4: next to the persistent walks of the other batches -- 118 VGPRs, a quarter of a SIMD's register file
// per wave -- a CU usually has room for ONE more wave, not for the four of a 256-thread block, which then waits for a whole CU)
constexpr uint32_t DESC_BLOCK = 64;
template <uint32_t G>
__global__ void __launch_bounds__(DESC_BLOCK) k_descend(const ugp_result *__restrict__ coarse_res, const uint32_t *__restrict__ order, uint32_t n_queries,
                          const uint32_t *__restrict__ coarse2bfs, const uint2 *__restrict__ node_pair,
                          const uint32_t *__restrict__ parent, const uint32_t *__restrict__ stream, const uint32_t *__restrict__ table, uint32_t n_sites,
                          uint32_t *__restrict__ refined, uint32_t max_expansions, int slack, const uint32_t *__restrict__ skip,
                          uint32_t *__restrict__ dnode /* or null: [n_queries] by slot -- (BFS index << 1 | has_unique) of a node whose cost is refined[slot], UINT32_MAX: none */) {
    constexpr uint32_t NG = DESC_BLOCK / G;   // samples per block
    __shared__ uint32_t f_node[NG][DESC_FRONTIER], f_cb[NG][DESC_FRONTIER], f_ce[NG][DESC_FRONTIER];
    __shared__ int f_d[NG][DESC_FRONTIER];
    const uint32_t g = threadIdx.x / G, gl = threadIdx.x % G;       // group (sample) within the block, lane within the group
    const uint32_t gsh = (threadIdx.x & 63u) / G * G;                // bit offset of the group in a wave-wide ballot
    const uint32_t slot = blockIdx.x * NG + g;
    ugp_result r;
    r.best_set_difference = -1; r.best_j = 0xFFFFFFFFu; r.num_best = 0; r.best_has_unique = 0;
    if (slot < n_queries) r = coarse_res[order ? order[slot] : slot];
    const bool alive = slot < n_queries && r.best_j != 0xFFFFFFFFu && r.best_set_difference >= 0 && r.best_set_difference <= 0x7F7F;
    const uint32_t *trow = table + ((uint64_t)(slot >> 9) * (n_sites + TABLE_CONST_ROWS) + TABLE_CONST_ROWS) * 64 + ((slot & 511u) >> 3);
    const uint32_t sh = (slot & 7u) * 4u;
    // one node for one sample: sum of delta over all words, sum of min(delta, 0) and shared mutations before the first masked one
    auto eval = [&](uint32_t rec_at, int &dsum, int &neg, uint32_t &common, uint32_t &w0, bool &excluded) {
        const uint32_t *rec = stream + rec_at;
        w0 = rec[0];
        excluded = (rec[1] & KEY_EXCLUDED) != 0;   // rewritten since the tree was flattened: its cost is no bound of best(s)
        const uint32_t m0 = rec[2], m1 = rec[3];   // (fetched with w0; real only if the record is that long)
        const uint32_t nw = w0 & 0xFFFFu;
        dsum = 0; neg = 0; common = 0;
        auto one = [&](uint32_t w, uint32_t S) {
            const int c = (int)((S >> ((w >> 22) & 3u)) & 1u), p = (int)((S >> ((w >> 24) & 3u)) & 1u);
            const int d = p - c;
            dsum += d;
            if (!(w & M_AFTER_MASK)) { neg += min(d, 0); common += (uint32_t)c; }
        };
        if (nw >= 1) {
            const uint32_t r0 = trow[(uint64_t)(m0 & 0x3FFFFFu) * 64];
            const uint32_t r1 = nw >= 2 ? trow[(uint64_t)(m1 & 0x3FFFFFu) * 64] : 0u;
            one(m0, (r0 >> sh) & 15u);
            if (nw >= 2) one(m1, (r1 >> sh) & 15u);
            for (uint32_t k = 2; k < nw; k++) { const uint32_t w = rec[2 + k]; one(w, (trow[(uint64_t)(w & 0x3FFFFFu) * 64] >> sh) & 15u); }
        }
    };
    int best = alive ? r.best_set_difference : 0x7F7F;
    // (round 6) which node attains `best`: cost << 34 | BFS index << 1 | has_unique, smallest first -- for phase 2's samples whose minimum
    // is attained by one node only (k_select): that node is then this one, and no chunk is walked for it
    unsigned long long bkey = ~0ull;
    // one node left out for this sample (ugp_place_opts::skip_node): its cost is never a bound; the search still passes through it
    const uint32_t skipn = (skip && slot < n_queries) ? skip[order ? order[slot] : slot] : 0xFFFFFFFFu;
    uint32_t n_f = 0;   // uniform within the group: frontier entries
    uint32_t start[DESC_UP + 1];   // the start nodes (an ancestor's expansion must not enter the next one again)
#pragma unroll
    for (uint32_t i = 0; i <= DESC_UP; i++) start[i] = 0xFFFFFFFFu;
    if (alive) {   // (every lane of the group does the same loads: broadcasts)
        uint32_t node = coarse2bfs[r.best_j];
        start[0] = node;
        uint2 pr = node_pair[node];
        uint32_t ce = node_pair[node + 1].x;
        int dsum, neg; uint32_t common, w0; bool excl;
        eval(pr.y, dsum, neg, common, w0, excl);
        // cost(node) = D(parent) + neg = best  ->  D(parent) = best - neg, D(node) = D(parent) + dsum;  the root's cost is its D
        int D = (w0 & F_ROOT) ? best : best - neg + dsum;
        if (node == skipn) best = 0x7F7F;   // (D above is derived from the node's true cost; as a bound it does not count)
        else {
            // the coarse winner is a node of the full tree with the same mutations, the same cost and eligible (k_seed_ub); has_unique as
            // walk_ties computes it: a masked mutation, or a mutation the sample does not share; the root never
            const uint32_t hu0 = (w0 & F_ROOT) ? 0u : (((w0 & F_MASKED) || common != (w0 & 0xFFFFu)) ? 1u : 0u);
            if (!excl) bkey = ((unsigned long long)(uint32_t)best << 34) | ((unsigned long long)node << 1) | hu0;
        }
        if (gl == 0) { f_node[g][0] = node; f_d[g][0] = D; f_cb[g][0] = pr.x + 1u; f_ce[g][0] = ce + 1u; }
        n_f = 1;
        for (uint32_t up = 0; up < DESC_UP && node != 0; up++) {   // D(ancestor) = D(child) - (sum of the child's deltas)
            D -= dsum;
            node = parent[node];
            start[up + 1] = node;
            pr = node_pair[node];
            ce = node_pair[node + 1].x;
            if (gl == 0) { f_node[g][n_f] = node; f_d[g][n_f] = D; f_cb[g][n_f] = pr.x + 1u; f_ce[g][n_f] = ce + 1u; }
            n_f++;
            eval(pr.y, dsum, neg, common, w0, excl);
        }
    }
    for (uint32_t it = 0; it < max_expansions; it++) {
        const bool act = n_f != 0;   // this group still has something to expand
        if (__builtin_amdgcn_ballot_w64(act) == 0) break;
        // pop the entry with the smallest D (G = 16: two candidates per lane, DESC_FRONTIER = 2 x 16)
        unsigned long long kk = ~0ull;
        if (act && gl < n_f) kk = ((unsigned long long)(uint32_t)f_d[g][gl] << 32) | gl;
        if (G < DESC_FRONTIER && act && gl + G < n_f) { const unsigned long long k2 = ((unsigned long long)(uint32_t)f_d[g][gl + G] << 32) | (gl + G); kk = k2 < kk ? k2 : kk; }
#pragma unroll
        for (int o = (int)G / 2; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(kk, o, (int)G); kk = other < kk ? other : kk; }
        const uint32_t e = act ? ((uint32_t)kk & 31u) : 0u;
        const int D = f_d[g][e];
        // (the entry with the smallest D lies more than `slack` above the best cost seen: nothing left in the frontier is on the
        // sample's lineage any more -- there D and the best cost fall together -- and the search ends)
        const bool go = act && D <= best + slack;
        if (act && !go) n_f = 0;
        const uint32_t cb = go ? f_cb[g][e] : 0u, ce = go ? f_ce[g][e] : 0u;   // children: BFS indices [cb, ce)
        if (go) {
            n_f--;
            if (gl == 0 && e != n_f) { f_node[g][e] = f_node[g][n_f]; f_d[g][e] = f_d[g][n_f]; f_cb[g][e] = f_cb[g][n_f]; f_ce[g][e] = f_ce[g][n_f]; }
        }
        for (uint32_t c0 = cb;; c0 += G) {
            const bool more = c0 < ce;
            if (__builtin_amdgcn_ballot_w64(more) == 0) break;
            const uint32_t c = c0 + gl;
            int cost = 0x7FFFFFFF, dc = 0;
            unsigned long long ck = ~0ull;
            bool push = false;
            uint32_t ccb = 0, cce = 0;
            if (more && c < ce) {
                const uint2 pr = node_pair[c];
                cce = node_pair[c + 1].x + 1u;
                ccb = pr.x + 1u;
                int dsum, neg; uint32_t common, w0; bool excl;
                eval(pr.y, dsum, neg, common, w0, excl);
                const bool leaf = (w0 & F_LEAF) != 0, masked = (w0 & F_MASKED) != 0;
                if (!masked && !excl && c != skipn && (common > 0 || (!leaf && (w0 & 0xFFFFu) == 0))) {
                    cost = D + neg;
                    ck = ((unsigned long long)(uint32_t)cost << 34) | ((unsigned long long)c << 1) | (common != (w0 & 0xFFFFu) ? 1u : 0u);
                }
                dc = D + dsum;
                // follow a child whose D does not grow -- or grows by one while it shares a mutation with the sample (the sample's
                // lineage passing a node of which it lacks one mutation; a sibling branch shares one only by homoplasy)
                push = !leaf && (dc <= D || (common > 0 && dc <= D + 1));
#pragma unroll
                for (uint32_t i = 0; i < DESC_UP; i++) push = push && c != start[i];   // (already in the frontier)
            }
#pragma unroll
            for (int o = (int)G / 2; o > 0; o >>= 1) cost = min(cost, __shfl_xor(cost, o, (int)G));
            if (dnode) {
#pragma unroll
                for (int o = (int)G / 2; o > 0; o >>= 1) { const unsigned long long other = __shfl_xor(ck, o, (int)G); ck = other < ck ? other : ck; }
                bkey = ck < bkey ? ck : bkey;
            }
            best = min(best, cost);
            const unsigned long long pm = (__builtin_amdgcn_ballot_w64(push) >> gsh) & (G == 64 ? ~0ull : ((1ull << (G & 63u)) - 1ull));
            if (push) {
                const uint32_t at = n_f + (uint32_t)__builtin_popcountll(pm & ((1ull << gl) - 1ull));
                if (at < DESC_FRONTIER) { f_node[g][at] = c; f_d[g][at] = dc; f_cb[g][at] = ccb; f_ce[g][at] = cce; }
            }
            n_f = min(n_f + (uint32_t)__builtin_popcountll(pm), DESC_FRONTIER);
        }
    }
    if (slot < n_queries && gl == 0) {
        refined[slot] = alive ? (uint32_t)max(0, min(best, 0x7F7F)) : 0x7F7Fu;
        // (valid only when the key's cost IS the value stored above: a cost outside [0, 0x7F7F) is clamped there and names no node)
        if (dnode) dnode[slot] = (alive && bkey != ~0ull && (int)(bkey >> 34) == best && best >= 0 && best < 0x7F7F) ? (uint32_t)(bkey & 0xFFFFFFFFull) : 0xFFFFFFFFu;
    }
}

// Where in the chunk order do a tile's own samples sit?  keys_sorted[q] = DFS rank of the coarse best node of
// the q-th sample in tile order; the region runs from the first sample's chunk to a few chunks past the
// last one's (the coarse node's own subtree).  Scheduling hint only.
__global__ void k_tile_ranges(const uint32_t *__restrict__ keys_sorted, uint32_t n_queries, uint32_t n_tiles,
                              const uint32_t *__restrict__ chunk_node_off, uint32_t n_chunks, uint32_t align,
                              uint32_t *__restrict__ hstart, uint32_t *__restrict__ hlen) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_tiles) return;
    const uint32_t q0 = t * 512u, q1 = min(n_queries, q0 + 512u) - 1u;
    if (q0 >= n_queries) { hstart[t] = 0; hlen[t] = 0; return; }
    auto chunk_of = [&](uint32_t d) {   // last c with chunk_node_off[c] <= d
        uint32_t lo = 0, hi = n_chunks;
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (chunk_node_off[mid] <= d) lo = mid; else hi = mid; }
        return lo;
    };
    const uint32_t c_first = chunk_of(keys_sorted[q0]);
    const uint32_t c_last = min(n_chunks - 1u, chunk_of(keys_sorted[q1]) + 3u);
    // snapped to the unit grid, so the units of all tiles cover the same chunk ranges
    const uint32_t h0 = c_first / align * align;
    const uint32_t h1 = min(n_chunks, (c_last / align + 1u) * align);
    hstart[t] = h0;
    hlen[t] = h1 - h0;
}

// The work units of k_best8, one list per queue (= XCD; queue x owns the tiles [x T / 8, (x + 1) T / 8)).
// Units of a tile: its own region H = [h0, h0 + hl) (the chunks its samples sit in, from the locality sort; hl = 0
// without it) cannot be pruned and is dense work -- heavy_chunks chunks per unit, and every tile's H comes first in the
// list.  The rest of the ring, A = [h0 + hl, n_chunks) ascending and B = [0, h0) descending, is mostly jumped over; its
// units are listed nearest first (rank k of every tile before rank k + 1 of any, alternating between A and B; with
// light_order 1 one tile after the other, each nearest first), since the ones next to the region are pruned least and
// would otherwise be the tail of the kernel.  A unit far from the region almost always ends in the replay of its
// preamble or after a few jumps, so what it costs is the replay, not its length: the units grow with the distance --
// `unit_chunks` chunks for the first `grow_every` units of a side, twice that for the next `grow_every`, ... up to
// `unit_max` (grow_every = 0: all the same size).  Entry = {tile, c0, c1, flags (bit 0: own region)}; c0 == c1: filler.
// One block per queue; thread t lists the tiles tlo + t, tlo + t + 256, ...
__global__ void __launch_bounds__(256) k_build_units(const uint32_t *__restrict__ hstart, const uint32_t *__restrict__ hlen, uint32_t n_tiles, uint32_t n_chunks,
                                                     uint32_t unit_chunks, uint32_t heavy_chunks, uint32_t grow_every, uint32_t unit_max, uint32_t light_order,
                                                     uint32_t per_tile_cap, uint4 *__restrict__ units, uint32_t *__restrict__ unit_base, uint32_t *__restrict__ unit_count,
                                                     uint32_t *__restrict__ dyn_ctl) {
    __shared__ uint32_t s_heavy[512], s_light[512], s_hoff[512], s_loff[512];   // (at most 4,096 tiles per launch: 512 per queue)
    __shared__ uint32_t s_max_light, s_heavy_total;
    const uint32_t x = blockIdx.x, tlo = x * n_tiles / 8u, thi = (x + 1u) * n_tiles / 8u, T = thi - tlo;
    const uint32_t U = max(unit_chunks, 1u), HU = max(heavy_chunks, 1u), UM = max(unit_max, U);
    auto len_of = [&](uint32_t i) -> uint32_t {   // length of the i-th unit of a side
        if (!grow_every) return U;
        const uint32_t sh = min(i / grow_every, 16u);
        return (uint32_t)min((uint64_t)U << sh, (uint64_t)UM);
    };
    // units a side of `chunks` chunks is cut into: by growth step, not unit by unit (every thread of the block runs this for every
    // tile of its queue: unit by unit it was 70-100 us of a step's dependent chain)
    auto count_side = [&](uint32_t chunks) -> uint32_t {
        if (!chunks) return 0u;
        if (!grow_every) return (chunks + U - 1u) / U;
        uint32_t n = 0;
        uint64_t done = 0;
        for (uint32_t g = 0;; g++) {
            const uint64_t L = min((uint64_t)U << min(g, 16u), (uint64_t)UM);
            const uint64_t left = chunks - done;
            if (L == UM || g >= 16u || (uint64_t)grow_every * L >= left) return n + (uint32_t)((left + L - 1u) / L);
            n += grow_every; done += (uint64_t)grow_every * L;
        }
    };
    for (uint32_t t = threadIdx.x; t < T; t += blockDim.x) {
        const uint32_t h0 = hstart ? hstart[tlo + t] : 0u, hl = hstart ? hlen[tlo + t] : 0u;
        s_heavy[t] = (hl + HU - 1u) / HU;
        s_light[t] = count_side(n_chunks - h0 - hl) + count_side(h0);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t ho = 0, lo = 0, ml = 0;
        for (uint32_t t = 0; t < T; t++) { s_hoff[t] = ho; s_loff[t] = lo; ho += s_heavy[t]; lo += s_light[t]; ml = max(ml, s_light[t]); }
        s_heavy_total = ho; s_max_light = ml;
        unit_base[x] = tlo * per_tile_cap;
        if (x == 0 && dyn_ctl) { dyn_ctl[DYN_HEAD] = 0; dyn_ctl[DYN_TAIL] = 0; dyn_ctl[DYN_ACTIVE] = 0; }
        unit_count[x] = ho + (light_order == 1u ? lo : ml * T);
    }
    __syncthreads();
    uint4 *out = units + (uint64_t)tlo * per_tile_cap;
    // own-region units: one thread each (a tile's run of them is found by bisection over the offsets); the units outside are
    // a short sequential list per tile
    for (uint32_t i = threadIdx.x; i < s_heavy_total; i += blockDim.x) {
        uint32_t lo = 0, hi = T;   // the last t with s_hoff[t] <= i
        while (hi - lo > 1u) { const uint32_t mid = (lo + hi) / 2u; if (s_hoff[mid] <= i) lo = mid; else hi = mid; }
        const uint32_t tile = tlo + lo, r = i - s_hoff[lo];
        const uint32_t h0 = hstart[tile], hl = hlen[tile];
        const uint32_t c0 = h0 + r * HU;
        out[i] = make_uint4(tile, c0, min(c0 + HU, h0 + hl), 1u);
    }
    // the units outside: one thread per (tile, rank) -- the k-th unit of a tile is the (k / 2)-th of alternating sides while both
    // last, then the rest of the longer one; where the i-th unit of a side begins is a sum over the growth steps in front of it
    // (a tile's list used to be written by ONE thread, ~150 dependent iterations: 100 us of a step's dependent chain)
    auto start_of = [&](uint32_t i) -> uint64_t {   // chunks covered by the units 0 .. i-1 of a side
        if (!grow_every) return (uint64_t)i * U;
        uint64_t sum = 0;
        for (uint32_t g = 0; g * grow_every < i; g++) {
            const uint32_t cnt = min(grow_every, i - g * grow_every);
            sum += (uint64_t)cnt * min((uint64_t)U << min(g, 16u), (uint64_t)UM);
        }
        return sum;
    };
    const uint32_t n_rank_max = light_order == 1u ? 0u : s_max_light;
    for (uint32_t t = 0; t < T; t++) {
        const uint32_t tile = tlo + t;
        const uint32_t h0 = hstart ? hstart[tile] : 0u, hl = hstart ? hlen[tile] : 0u;
        const uint32_t lenA = n_chunks - h0 - hl, lenB = h0;
        const uint32_t nA = count_side(lenA), nB = count_side(lenB), m = min(nA, nB);
        const uint32_t n_rank = light_order == 1u ? nA + nB : n_rank_max;   // (rank-major lists are padded with fillers to the longest tile)
        for (uint32_t k = threadIdx.x; k < n_rank; k += blockDim.x) {
            uint32_t c0 = 0, c1 = 0;
            if (k < nA + nB) {
                const bool side_a = k < 2u * m ? !(k & 1u) : nA > nB;
                const uint32_t i = k < 2u * m ? k / 2u : k - m;
                const uint64_t before = start_of(i);
                if (side_a) { c0 = h0 + hl + (uint32_t)min(before, (uint64_t)lenA); c1 = (uint32_t)min((uint64_t)c0 + len_of(i), (uint64_t)n_chunks); }
                else { c1 = lenB - (uint32_t)min(before, (uint64_t)lenB); c0 = c1 - min(len_of(i), c1); }
            }
            const uint32_t at = s_heavy_total + (light_order == 1u ? s_loff[t] + k : k * T + t);
            out[at] = make_uint4(tile, c0, c1, 0u);
        }
    }
}

__global__ void k_extract_best(const ugp_result *__restrict__ res, uint32_t n, int32_t *__restrict__ best) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) best[q] = res[q].best_set_difference;
}

// ---------------------------------------------------------------- launchers

hipError_t launch_extract_best(const ugp_result *res, uint32_t n, int32_t *best, hipStream_t s) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_extract_best, dim3((n + 255) / 256), dim3(256), 0, s, res, n, best);
    return hipGetLastError();
}

hipError_t launch_tile_ranges(const uint32_t *keys_sorted, uint32_t n_queries, uint32_t n_tiles512, const uint32_t *chunk_node_off,
                              uint32_t n_chunks, uint32_t align, uint32_t *hstart, uint32_t *hlen, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_ranges, dim3((n_tiles512 + 63) / 64), dim3(64), 0, s, keys_sorted, n_queries, n_tiles512, chunk_node_off,
                       n_chunks, align, hstart, hlen);
    return hipGetLastError();
}

hipError_t launch_build_units(const uint32_t *hstart, const uint32_t *hlen, uint32_t n_tiles512, uint32_t n_chunks, uint32_t unit_chunks, uint32_t heavy_chunks,
                              uint32_t grow_every, uint32_t unit_max, uint32_t light_order, uint32_t per_tile_cap, void *units, uint32_t *unit_base,
                              uint32_t *unit_count, uint32_t *dyn_ctl, hipStream_t s) {
    // (one-wave blocks unless the lists are long: next to the persistent walks of the batches in front a CU has room for one more
    // wave, and a four-wave block waits for a CU to drain -- 105 us for the coarse pass's eight tiny lists in the pipelined trace)
    hipLaunchKernelGGL(k_build_units, dim3(8), dim3(n_tiles512 <= 512 ? 64 : 256), 0, s, hstart, hlen, n_tiles512, n_chunks, unit_chunks, heavy_chunks, grow_every, unit_max, light_order,
                       per_tile_cap, (uint4 *)units, unit_base, unit_count, dyn_ctl);
    return hipGetLastError();
}

hipError_t launch_seed_ub(const ugp_result *coarse_res, const uint32_t *order, uint32_t n_queries, uint32_t n_tiles512, uint32_t *ub,
                          const uint32_t *refined, uint32_t *dbottom, uint32_t pad_d, const uint32_t *skip, const uint32_t *coarse2bfs,
                          const uint32_t *dnode, uint32_t *dres, hipStream_t s) {
    const uint32_t n_words = n_tiles512 * 256;
    hipLaunchKernelGGL(k_seed_ub, dim3((n_words + 63) / 64), dim3(64), 0, s, coarse_res, order, n_queries, n_words, ub, refined, dbottom, pad_d,
                       skip, coarse2bfs, dnode, (dnode && refined) ? dres : nullptr);
    return hipGetLastError();
}

hipError_t launch_descend(const ugp_result *coarse_res, const uint32_t *order, uint32_t n_queries, const uint32_t *coarse2bfs,
                          const uint32_t *node_pair, const uint32_t *parent, const uint32_t *stream, const uint32_t *table,
                          uint32_t n_sites, uint32_t *refined, bool wide, uint32_t max_expansions, int slack, const uint32_t *skip, uint32_t *dnode, hipStream_t s) {
    if (!n_queries) return hipSuccess;
    const uint32_t max_exp = max_expansions ? max_expansions : DESC_MAX_EXPANSIONS;   // (tuning)
    // (slack, measured at 10M nodes: 0 costs the main walk 60 %, 1..5 are alike, none is 8 % more descent)
    if (wide)
        hipLaunchKernelGGL(k_descend<64>, dim3((n_queries + DESC_BLOCK / 64 - 1) / (DESC_BLOCK / 64)), dim3(DESC_BLOCK), 0, s, coarse_res, order, n_queries, coarse2bfs, (const uint2 *)node_pair, parent,
                           stream, table, n_sites, refined, max_exp, slack, skip, dnode);
    else
        hipLaunchKernelGGL(k_descend<16>, dim3((n_queries + DESC_BLOCK / 16 - 1) / (DESC_BLOCK / 16)), dim3(DESC_BLOCK), 0, s, coarse_res, order, n_queries, coarse2bfs, (const uint2 *)node_pair, parent,
                           stream, table, n_sites, refined, max_exp, slack, skip, dnode);
    return hipGetLastError();
}

// The locality sort without the device radix sort (round 4; it had been k_sort_keys + five rocprim kernels + k_invert, ~95 us of a
// batch's chain).  The keys take few distinct values -- the nodes of the coarse tree, 4,882 at 10M nodes -- so a counting sort does:
// histogram over the bins (coarse_bin[c] = position of coarse node c among the coarse nodes in depth-first order; one thread per
// sample, global atomics on `bins`, zeroed by the caller), exclusive scan of the bins (one block), scatter (one thread per sample
// again).  Samples of one bin come out in the order the scatter's atomics fall: the order is a scheduling hint, results never
// depend on it.  (One block doing all three steps in LDS was tried first: 16 serial rounds of dependent loads per thread, slower
// than the radix sort.)
constexpr uint32_t LSORT_MAX_BINS = 15360;   // (60 KB of LDS for the scan: within the 64 KB a block gets without asking)
__global__ void k_lsort_hist(const ugp_result *__restrict__ coarse_res, const uint32_t *__restrict__ coarse_bin, uint32_t n, uint32_t *__restrict__ bins) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) atomicAdd(&bins[coarse_bin[coarse_res[q].best_j]], 1u);
}
// (one wave: next to the persistent walks of the batches in front a CU has room for one more wave, and a four-wave block waits for a
// whole CU to drain -- 60 us on average, 906 us at worst in the round-4 trace, for 8 us of work.  Every lane owns a run of bins; the
// partial sums are scanned across the lanes with DPP-free shuffles.)
__global__ void __launch_bounds__(64) k_lsort_scan(uint32_t *__restrict__ bins, uint32_t n_bins) {
    extern __shared__ uint32_t lb[];   // [n_bins]: the counters are staged through LDS -- every global access coalesced and independent
    const uint32_t t = threadIdx.x;
    for (uint32_t b = t; b < n_bins; b += 64u) lb[b] = bins[b];
    __syncthreads();
    const uint32_t per = (n_bins + 63u) / 64u, b0 = min(t * per, n_bins), b1 = min(b0 + per, n_bins);   // every lane owns a run of bins
    uint32_t sum = 0;
    for (uint32_t b = b0; b < b1; b++) sum += lb[b];
    uint32_t inc = sum;   // inclusive scan over the 64 lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const uint32_t v = __shfl_up(inc, o); if ((int)t >= o) inc += v; }
    uint32_t run = inc - sum;
    for (uint32_t b = b0; b < b1; b++) { const uint32_t c = lb[b]; lb[b] = run; run += c; }
    __syncthreads();
    for (uint32_t b = t; b < n_bins; b += 64u) bins[b] = lb[b];
}
__global__ void k_lsort_scatter(const ugp_result *__restrict__ coarse_res, const uint32_t *__restrict__ coarse2dfs, const uint32_t *__restrict__ coarse_bin,
                                uint32_t n, uint32_t *__restrict__ bins, uint32_t *__restrict__ keys_sorted, uint32_t *__restrict__ order,
                                uint32_t *__restrict__ slot_of) {
    const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= n) return;
    const uint32_t c = coarse_res[q].best_j;
    const uint32_t pos = atomicAdd(&bins[coarse_bin[c]], 1u);
    order[pos] = q; keys_sorted[pos] = coarse2dfs[c]; slot_of[q] = pos;
}

hipError_t launch_locality_sort(const ugp_result *coarse_res, const uint32_t *coarse2dfs, uint32_t n, uint32_t *keys,
                                uint32_t *keys_sorted, uint32_t *idx, uint32_t *order, uint32_t *slot_of, void *temp,
                                size_t *temp_bytes, const uint32_t *coarse_bin, uint32_t n_bins, uint32_t *bins, hipStream_t s) {
    if (temp && coarse_bin && bins && n_bins && n_bins <= LSORT_MAX_BINS) {
        hipError_t e = hipMemsetAsync(bins, 0, (size_t)n_bins * sizeof(uint32_t), s);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(k_lsort_hist, dim3((n + 63) / 64), dim3(64), 0, s, coarse_res, coarse_bin, n, bins);
        hipLaunchKernelGGL(k_lsort_scan, dim3(1), dim3(64), (size_t)n_bins * sizeof(uint32_t), s, bins, n_bins);
        hipLaunchKernelGGL(k_lsort_scatter, dim3((n + 63) / 64), dim3(64), 0, s, coarse_res, coarse2dfs, coarse_bin, n, bins, keys_sorted, order, slot_of);
        return hipGetLastError();
    }
    if (!temp) {   // size query
        return rocprim::radix_sort_pairs(nullptr, *temp_bytes, keys, keys_sorted, idx, order, (size_t)n, 0u, 32u, s);
    }
    hipLaunchKernelGGL(k_sort_keys, dim3((n + 63) / 64), dim3(64), 0, s, coarse_res, coarse2dfs, n, keys, idx);
    hipError_t e = rocprim::radix_sort_pairs(temp, *temp_bytes, keys, keys_sorted, idx, order, (size_t)n, 0u, 32u, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_invert, dim3((n + 63) / 64), dim3(64), 0, s, order, n, slot_of);
    return hipGetLastError();
}

hipError_t launch_fill_table(uint32_t *table, const uint8_t *site_ref, uint32_t n_sites, uint64_t total_dwords,
                             hipStream_t s) {
    if (total_dwords == 0) return hipSuccess;
    uint64_t blocks = (total_dwords / 4 + 63) / 64;   // (the table is a whole number of 64-dword rows; one-wave blocks: see k_descend)
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(k_fill_table, dim3((uint32_t)blocks), dim3(64), 0, s, table, site_ref, n_sites + TABLE_CONST_ROWS, total_dwords);
    return hipGetLastError();
}

hipError_t launch_build_tiles(uint32_t *table, uint32_t *active, uint32_t active_words, uint32_t n_tiles512, const uint64_t *ent_off, uint32_t q0,
                              const uint32_t *order, uint32_t nq, const int32_t *pos, const uint8_t *ref, const uint8_t *nuc, const uint8_t *is_missing,
                              const int32_t *pos2site, const int32_t *site_pos, const uint8_t *site_ref, uint32_t n_sites, uint32_t max_pos,
                              uint32_t *dbottom, const unsigned long long *err, hipStream_t s) {
    if (!n_tiles512 || !n_sites) return hipSuccess;
    hipLaunchKernelGGL(k_build_tiles, dim3((n_sites + TB_SITES - 1) / TB_SITES, n_tiles512), dim3(256), 0, s, table, active, active_words, ent_off, q0, order, nq,
                       pos, ref, nuc, is_missing, pos2site, site_pos, site_ref, n_sites, max_pos, err);
    hipLaunchKernelGGL(k_row_counts, dim3((nq + 3) / 4), dim3(256), 0, s, ent_off, q0, order, nq, ref, nuc, is_missing, dbottom, err);
    return hipGetLastError();
}

hipError_t launch_scatter(uint32_t *table, uint32_t *dbottom, const int32_t *pos, const uint8_t *ref,
                          const uint8_t *nuc, const uint8_t *is_missing, const uint32_t *ent_q,
                          const int32_t *pos2site, uint32_t max_pos, uint32_t n_sites, uint64_t n_ent, uint32_t q_base,
                          uint32_t *active, uint32_t active_words, const uint32_t *slot_of, const unsigned long long *err, uint32_t *useful, uint32_t useful_words,
                          hipStream_t s) {
    if (n_ent == 0) return hipSuccess;
    constexpr int K = 4;
    const uint64_t blocks = (n_ent + 64 * K - 1) / (64 * K);
    hipLaunchKernelGGL(k_scatter_rows<K>, dim3((uint32_t)blocks), dim3(64), 0, s, table, dbottom, pos, ref, nuc,
                       is_missing, ent_q, pos2site, max_pos, n_sites, n_ent, q_base, active, active_words, slot_of, err, useful, useful_words);
    return hipGetLastError();
}

// the rows listed in row_list (their number is on the device: n_listed), restricted to the samples [q_base, q_base + n_q)
hipError_t launch_scatter_list(uint32_t *table, uint32_t *dbottom, const int32_t *pos, const uint8_t *ref, const uint8_t *nuc, const uint8_t *is_missing,
                               const uint32_t *ent_q, const int32_t *pos2site, uint32_t max_pos, uint32_t n_sites, uint32_t q_base, uint32_t n_q,
                               uint32_t *active, uint32_t active_words, const uint32_t *slot_of, const uint32_t *row_list, const uint32_t *n_listed,
                               const unsigned long long *err, uint32_t *useful, uint32_t useful_words, hipStream_t s) {
    hipLaunchKernelGGL(k_scatter_entries, dim3(8192), dim3(64), 0, s, table, dbottom, pos, ref, nuc, is_missing, ent_q, pos2site, max_pos, n_sites,
                       (uint64_t)0, q_base, active, active_words, slot_of, row_list, n_listed, n_q, err, useful, useful_words);
    return hipGetLastError();
}

hipError_t launch_nmask_build(const uint64_t *ent_off, uint32_t n_queries, const int32_t *pos, const uint8_t *is_missing, const int32_t *pos2site,
                              uint32_t max_pos, uint32_t words, uint32_t *nmask, uint32_t *plain_rows, uint32_t *n_plain, hipStream_t s) {
    if (!n_queries) return hipSuccess;
    hipLaunchKernelGGL(k_nmask_build, dim3(n_queries), dim3(256), (size_t)words * 4, s, ent_off, pos, is_missing, pos2site, max_pos, words, nmask,
                       plain_rows, n_plain);
    return hipGetLastError();
}

hipError_t launch_ntiles(uint32_t *table, uint32_t *active, uint32_t active_words, uint32_t n_tiles512, const uint32_t *nmask, uint32_t words,
                         const uint32_t *order, uint32_t q0, uint32_t nq, const uint8_t *site_ref, uint32_t n_sites, hipStream_t s) {
    if (!n_tiles512 || !n_sites) return hipSuccess;
    hipLaunchKernelGGL(k_ntiles, dim3((words + 15) / 16, n_tiles512), dim3(256), 0, s, table, active, active_words, nmask, words, order, q0, nq, site_ref,
                       n_sites);
    return hipGetLastError();
}

hipError_t launch_place(const PlaceArgs &a, int mode, uint32_t max_slots, hipStream_t s) {
    const uint32_t blocks = a.n_tiles * a.n_groups;
    const size_t lds = (size_t)max_slots * 64 * sizeof(uint32_t);
    if (mode == 0) hipLaunchKernelGGL((k_place<0, false>), dim3(blocks), dim3(64), lds, s, a);
    else if (mode == 1) hipLaunchKernelGGL((k_place<1, false>), dim3(blocks), dim3(64), lds, s, a);
    else if (mode == 2) hipLaunchKernelGGL((k_place<2, false>), dim3(blocks), dim3(64), lds, s, a);
    else if (mode == 4) hipLaunchKernelGGL((k_place<0, true>), dim3(blocks), dim3(64), lds, s, a);
    else hipLaunchKernelGGL((k_place<2, true>), dim3(blocks), dim3(64), lds, s, a);
    return hipGetLastError();
}

// One launch per level; level_off: host array of the breadth-first level boundaries.  d16: D fits 16 bits.
hipError_t launch_scores_levels(const uint32_t *node_pair, const uint32_t *parent, const uint32_t *stream, const uint32_t *table, uint32_t n_sites,
                                const uint32_t *dbottom, const uint32_t *level_off, uint32_t n_levels, void *d_a, void *d_b, bool d16, uint32_t d_stride,
                                uint32_t qpad, uint32_t n_queries, uint64_t n_nodes, int32_t *scores, uint32_t block, hipStream_t s) {
    const uint32_t bs = block ? block : 1024u;   // (tuning: threads per block)
    for (uint32_t l = 0; l < n_levels; l++) {
        const uint32_t b = level_off[l], e = level_off[l + 1], pb = l ? level_off[l - 1] : 0u;
        if (e <= b) continue;
        const uint32_t bx = (e - b + bs - 1u) / bs;
        // narrow levels: the sample groups spread over blockIdx.y so that the top of the tree does not run on a handful of waves
        uint32_t by = 1;
        while (by < (qpad + SCORES_SB - 1u) / SCORES_SB && (uint64_t)bx * by < 2048u) by *= 2;
        void *prev = (l & 1u) ? d_a : d_b, *cur = (l & 1u) ? d_b : d_a;
        if (d16)
            hipLaunchKernelGGL(k_scores_level<uint16_t>, dim3(bx, by), dim3(bs), 0, s, (const uint2 *)node_pair, parent, stream, table, n_sites, dbottom, b, e, pb,
                               (const uint16_t *)prev, (uint16_t *)cur, d_stride, qpad, n_queries, n_nodes, scores);
        else
            hipLaunchKernelGGL(k_scores_level<uint32_t>, dim3(bx, by), dim3(bs), 0, s, (const uint2 *)node_pair, parent, stream, table, n_sites, dbottom, b, e, pb,
                               (const uint32_t *)prev, (uint32_t *)cur, d_stride, qpad, n_queries, n_nodes, scores);
    }
    return hipGetLastError();
}

hipError_t launch_merge(const uint32_t *part_best, const uint32_t *part_cnt, const uint32_t *part_key,
                        const uint32_t *rank2bfs, uint32_t n_groups, uint32_t n_queries, ugp_result *out,
                        hipStream_t s) {
    if (n_queries == 0) return hipSuccess;
    hipLaunchKernelGGL(k_merge, dim3((n_queries + 255) / 256), dim3(256), 0, s, part_best, part_cnt, part_key,
                       rank2bfs, n_groups, n_queries, out);
    return hipGetLastError();
}

// Resident one-wave blocks of k_best8 per CU for a given dynamic LDS size, on the current device.
// variant: 0 = the main walk, 1 = with the active-row bitmap in LDS, 2 = the coarse pass (records which node set each minimum) --
// their register and LDS needs differ, and the persistent grid and its cold-slot scratch are sized from this number.
hipError_t best8_occupancy(size_t lds_bytes, int variant, int *per_cu) {
    if (variant == 1) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_best8<false, 1, false, false>, 64, lds_bytes);
    if (variant == 2) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_best8<false, 0, true, false>, 64, lds_bytes);
    if (variant == 5) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_best8<false, 0, false, false, true>, 64, lds_bytes);
    if (variant == 6) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_best8<false, 2, false, false, true>, 64, lds_bytes);
    if (variant == 3) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_best8<false, 2, false, false>, 64, lds_bytes);
    if (variant == 4) return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_best8<false, 2, true, false>, 64, lds_bytes);
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(per_cu, k_best8<false, 0, false, false>, 64, lds_bytes);
}

// Persistent grid of `blocks` one-wave workgroups (the caller sizes a.cold for exactly that many).
hipError_t launch_best8(const Best8Args &a, uint32_t blocks, hipStream_t s) {
    const size_t lds = (size_t)a.lds_slots * 64 * 16;   // the D rows of the hot slots (their B halves live in registers)
#ifdef UGP_EXPERIMENTS   // (the statistics build of the walk and phase 2 as a mode of it exist only in libusher_amd_exp.so)
    if (a.tie_cnt) { hipLaunchKernelGGL((k_best8<false, 0, false, true>), dim3(blocks), dim3(64), lds, s, a); return hipGetLastError(); }   // (phase 2)
    if (a.stats && !a.lpos && a.b3) { hipLaunchKernelGGL((k_best8<true, 0, false, false, true>), dim3(blocks), dim3(64), lds, s, a); return hipGetLastError(); }
    if (a.stats && !a.lpos) { hipLaunchKernelGGL((k_best8<true, 0, false, false>), dim3(blocks), dim3(64), lds, s, a); return hipGetLastError(); }
#endif
    if (a.b3 && !a.lpos && a.lds_bits != 1) {   // the main walk with the third bound (not with the bitmap in LDS: that variant is for long launches)
        if (a.lds_bits == 2) hipLaunchKernelGGL((k_best8<false, 2, false, false, true>), dim3(blocks), dim3(64), lds, s, a);
        else hipLaunchKernelGGL((k_best8<false, 0, false, false, true>), dim3(blocks), dim3(64), lds, s, a);
        return hipGetLastError();
    }
    if (a.lpos && a.lds_bits == 2) hipLaunchKernelGGL((k_best8<false, 2, true, false>), dim3(blocks), dim3(64), lds, s, a);   // (the coarse pass of a batch whose rows are all live)
    else if (a.lpos) hipLaunchKernelGGL((k_best8<false, 0, true, false>), dim3(blocks), dim3(64), lds, s, a);   // (the coarse pass; no statistics there)
    else if (a.lds_bits == 2) hipLaunchKernelGGL((k_best8<false, 2, false, false>), dim3(blocks), dim3(64), lds, s, a);
    else if (a.lds_bits) hipLaunchKernelGGL((k_best8<false, 1, false, false>), dim3(blocks), dim3(64), lds + (((size_t)a.active_words * 4 + 15) & ~(size_t)15), s, a);
    else hipLaunchKernelGGL((k_best8<false, 0, false, false>), dim3(blocks), dim3(64), lds, s, a);
    return hipGetLastError();
}

hipError_t launch_coarse_result(const uint32_t *lbest, const uint32_t *lpos, const uint32_t *list, const uint32_t *list_n, uint32_t n_chunks, uint32_t n_tiles512,
                                uint32_t n_queries, const uint32_t *chunk_node_off, const uint32_t *chunk8_body_off, const uint32_t *node_pos8,
                                const uint32_t *dfs2bfs, ugp_result *out, hipStream_t s) {
    hipLaunchKernelGGL(k_coarse_result, dim3(n_tiles512), dim3(256), 0, s, lbest, lpos, list, list_n, n_chunks, n_tiles512, n_queries, chunk_node_off,
                       chunk8_body_off, node_pos8, dfs2bfs, out);
    return hipGetLastError();
}

#ifdef UGP_EXPERIMENTS
// Phase 2 as a mode of the packed walk (k_best8<TIES>): b = the arguments of the phase-1 walk it follows; units: room for n_chunks entries per
// tile, info: 128 dwords, blocks: the walk's grid (its cold-slot scratch is reused).
hipError_t launch_phase2_packed(const Best8Args &b1, const uint32_t *list, const uint32_t *list_n, uint32_t *gbest_part, uint32_t *gbest, uint32_t n_tiles512,
                                void *units, uint32_t *info, uint32_t *cnt, uint32_t *key, const uint32_t *node_pos8, const uint32_t *rank_dfs,
                                const uint32_t *chunk_node_off, const uint32_t *rank2bfs, uint32_t n_queries, ugp_result *out, const uint32_t *order,
                                uint32_t blocks, hipStream_t s) {
    const uint32_t per_chunk = n_tiles512 * 256;
    const uint32_t slices = std::min<uint32_t>(GBEST_SLICES, b1.n_chunks);
    hipLaunchKernelGGL(k_gbest, dim3(n_tiles512 * 4, slices), dim3(64), 0, s, b1.lbest, list, list_n, b1.n_chunks, n_tiles512, gbest_part, (uint32_t *)nullptr);
    hipLaunchKernelGGL(k_gbest2, dim3((per_chunk + 255) / 256), dim3(256), 0, s, gbest_part, slices, per_chunk, gbest, (const uint32_t *)nullptr, (uint32_t *)nullptr);
    hipError_t e = hipMemsetAsync(info, 0, 128 * sizeof(uint32_t), s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_select8, dim3(n_tiles512, n_tiles512 < 256 ? 8 : 1), dim3(256), 0, s, b1.lbest, list, list_n, gbest, b1.n_chunks, n_tiles512,
                       (uint4 *)units, info);
    Best8Args b = b1;
    b.ub = gbest;
    b.units = (const uint4 *)units; b.unit_base = info; b.unit_count = info + 8; b.queue = info + 16; b.dyn_ctl = info + 32;
    b.dyn_units = nullptr; b.dyn_cap = 0; b.split_cycles = b.split_heavy = 0xFFFFFFFFu;   // (units of one chunk: nothing to cut, nothing to wait for)
    b.lpos = nullptr; b.stats = nullptr; b.trace = nullptr; b.lds_bits = 0; b.heavy_prio = 0; b.ub_every = 0x7FFFFFFFu;
    b.tie_cnt = cnt; b.tie_key = key; b.node_pos8 = node_pos8; b.rank_dfs = rank_dfs; b.chunk_node_off = chunk_node_off; b.n_queries = n_queries;
    e = launch_best8(b, blocks, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_final, dim3((n_queries + 255) / 256), dim3(256), 0, s, gbest, cnt, key, rank2bfs, n_queries, out, order, (const uint32_t *)nullptr, (const uint32_t *)nullptr);
    return hipGetLastError();
}
#endif

hipError_t launch_fix_skip(const PlaceArgs &a, uint32_t *lbest, const uint32_t *skip_chunk, uint32_t n_tiles512, const uint32_t *rank2bfs, const uint32_t *order,
                           uint32_t max_slots, hipStream_t s) {
    if (!a.n_queries || !a.skip) return hipSuccess;
    hipLaunchKernelGGL(k_fix_skip, dim3(a.n_queries), dim3(64), (size_t)max_slots * 64 * sizeof(uint32_t), s, a, lbest, skip_chunk, n_tiles512, rank2bfs, order);
    return hipGetLastError();
}

hipError_t launch_phase2(const PlaceArgs &a, const uint32_t *lbest, const uint32_t *list, const uint32_t *list_n, uint32_t *gbest_part,
                         uint32_t *gbest, uint32_t n_tiles512,
                         uint32_t *items, uint32_t *n_items, uint32_t cap, uint32_t *cnt, uint32_t *key,
                         const uint32_t *rank2bfs, const uint32_t *rank2out, ugp_result *out, const uint32_t *order, uint32_t max_slots, bool lists,
                         const Phase2Uniq *u, hipStream_t s) {
    const uint32_t per_chunk = n_tiles512 * 256;
    const uint32_t slices = std::min<uint32_t>(GBEST_SLICES, a.n_chunks);
    const bool uq = u && u->luniq && u->dnode && u->refined && u->dres && u->gcnt && u->gcnt_part && !lists && !rank2out;
    hipLaunchKernelGGL(k_gbest, dim3(n_tiles512 * 4, slices), dim3(64), 0, s, lbest, list, list_n, a.n_chunks, n_tiles512, gbest_part, uq ? u->gcnt_part : nullptr);
    hipLaunchKernelGGL(k_gbest2, dim3((per_chunk + 63) / 64), dim3(64), 0, s, gbest_part, slices, per_chunk, gbest, uq ? u->gcnt_part : nullptr, uq ? u->gcnt : nullptr);
    const uint64_t pairs = (uint64_t)a.n_chunks * n_tiles512 * 8;
    hipLaunchKernelGGL(k_select, dim3(n_tiles512, n_tiles512 < 256 ? 32 : 4), dim3(64), 0, s, lbest, list, list_n, gbest, a.n_chunks, n_tiles512, a.n_queries, items,
                       n_items, cap, uq ? u->gcnt : nullptr, uq ? u->luniq : nullptr, uq ? u->dres : nullptr);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const size_t lds = (size_t)max_slots * 64 * sizeof(uint32_t);
    uint32_t blocks = 256 * 32;   // latency-bound walk: as many waves as a CU holds (3.3 KB of LDS each)
    if ((uint64_t)blocks > pairs) blocks = (uint32_t)pairs;
    if (lists) hipLaunchKernelGGL(k_ties<true>, dim3(blocks), dim3(64), lds, s, a, lbest, gbest, items, n_items, cap, n_tiles512 * 8, cnt, key, rank2bfs, order);
    else hipLaunchKernelGGL(k_ties<false>, dim3(blocks), dim3(64), lds, s, a, lbest, gbest, items, n_items, cap, n_tiles512 * 8, cnt, key, rank2bfs, order);
    hipLaunchKernelGGL(k_final, dim3((a.n_queries + 63) / 64), dim3(64), 0, s, gbest, cnt, key, rank2out ? rank2out : rank2bfs, a.n_queries, out, order,
                       uq ? u->dnode : nullptr, uq ? u->refined : nullptr);
    return hipGetLastError();
}

}  // namespace ugp
