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
// Mapping to the hardware.  Lanes = samples: one 64-lane wavefront owns a
// "tile" of 64 query samples and walks a contiguous range of the tree's DFS
// record stream.  The stream is wave-uniform: 64 dwords at a time are loaded
// coalesced into one VGPR and handed to the scalar unit with v_readlane, so
// all control flow (record decode, slot numbers, loop counts) runs on SGPRs.
// The only per-lane memory traffic is one coalesced 32-byte row of the tile's
// 4-bit allele table per tree mutation (8 lanes share a dword) and the D stack
// in LDS (one 256-byte row per saved ancestor; depth <= log2 N by
// construction, see ugp_flatten.cpp).  Integer work only; no MFMA.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugp_kernels.hpp"

namespace ugp {

// ----------------------------------------------------------- allele tiles

// table[tile][site][8 dwords]: 64 nibbles, nibble l = allele set of sample
// (tile*64 + l) at that site.  Initialised to the reference base everywhere
// (a sample without a VCF row at a position carries the reference allele,
// usher_mapper.cpp:244, 301, 425).
__global__ void k_fill_table(uint32_t *__restrict__ table, const uint8_t *__restrict__ site_ref,
                             uint32_t n_sites, uint64_t total_dwords) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (; i < total_dwords; i += stride) {
        uint32_t site = (uint32_t)((i >> 3) % n_sites);
        table[i] = (uint32_t)site_ref[site] * 0x11111111u;
    }
}

// One thread per VCF row: overwrite the sample's nibble at tree sites and count
// D_bottom = #{non-missing rows whose allele set excludes the reference base}
// (usher_mapper.cpp:292-388 with an empty ancestral list).
__global__ void k_scatter_entries(uint32_t *__restrict__ table, uint32_t *__restrict__ dbottom,
                                  const int32_t *__restrict__ pos, const uint8_t *__restrict__ ref,
                                  const uint8_t *__restrict__ nuc, const uint8_t *__restrict__ is_missing,
                                  const uint32_t *__restrict__ ent_q, const int32_t *__restrict__ pos2site,
                                  uint32_t max_pos, uint32_t n_sites, uint64_t n_ent, uint32_t q_base) {
    uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_ent) return;
    const uint32_t q = ent_q[e] - q_base;   // sample index within this launch's tiles
    const int32_t p = pos[e];
    const uint32_t r = ref[e];
    const uint32_t miss = is_missing[e];
    const uint32_t a = miss ? 15u : (uint32_t)nuc[e];
    if (!miss && (a & r) == 0) atomicAdd(&dbottom[q], 1u);
    if (p < 0 || (uint32_t)p > max_pos) return;
    const int32_t site = pos2site[p];
    if (site < 0) return;
    const uint32_t tile = q >> 6, lane = q & 63;
    uint32_t *w = table + ((uint64_t)tile * n_sites + (uint32_t)site) * 8 + (lane >> 3);
    const uint32_t sh = (lane & 7) * 4;
    atomicXor(w, ((r ^ a) & 15u) << sh);   // nibble was r (k_fill_table); rows are unique per (sample, position)
}

// ------------------------------------------------------------ stream reader

struct Reader {
    const uint32_t *p;
    uint32_t base, end;   // dword offsets; uniform
    uint32_t buf;         // lane l holds p[base + l]
    uint32_t cur;         // uniform
    __device__ __forceinline__ void init(const uint32_t *ptr, uint32_t begin, uint32_t end_, uint32_t lane) {
        p = ptr; base = begin; end = end_; cur = 0;
        buf = (base + lane < end) ? p[base + lane] : 0u;
    }
    __device__ __forceinline__ bool done() const { return base + cur >= end; }
    __device__ __forceinline__ uint32_t next(uint32_t lane) {
        if (cur == 64) {
            base += 64; cur = 0;
            buf = (base + lane < end) ? p[base + lane] : 0u;
        }
        uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)buf, (int)cur);
        cur++;
        return w;
    }
};

// -------------------------------------------------------------- the kernel

template <int MODE>   // 0: best placement  1: per-node scores  2: collect tied nodes
__global__ void __launch_bounds__(64) k_place(PlaceArgs a) {
    extern __shared__ __attribute__((aligned(16))) uint32_t slots[];   // [max_slots][64]
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
    if (c0 >= c1) {
        if (MODE == 0) {
            const uint64_t o = ((uint64_t)tile * a.n_groups + g) * 64 + lane;
            a.part_best[o] = 0x7fffffffu; a.part_cnt[o] = 0; a.part_key[o] = 0;
        }
        return;
    }
    const uint32_t *tab = a.table + (uint64_t)tile * a.n_sites * 8 + (lane >> 3);
    const uint32_t sh = (lane & 7u) * 4u;
    const uint32_t q = tile * 64 + lane;
    const uint32_t dbot = a.dbottom[q];

    uint32_t best = 0x7fffffffu, cnt = 0, bkey = 0;
    uint32_t want_best = 0;
    if (MODE == 2) want_best = (q < a.n_queries) ? (uint32_t)a.best_in[q] : 0xffffffffu;
    uint32_t dcur = 0;
    uint32_t node_idx = a.chunk_node_off[c0];   // DFS index of the next body record (MODE 1)
    (void)node_idx;

    for (int phase = 0; phase < 2; phase++) {
        Reader rd;
        if (phase == 0) rd.init(a.pre_stream, a.chunk_pre_off[c0], a.chunk_pre_off[c0 + 1], lane);
        else rd.init(a.stream, a.chunk_body_off[c0], a.chunk_body_off[c1], lane);
        while (!rd.done()) {
            const uint32_t w0 = rd.next(lane);
            const uint32_t key = rd.next(lane);
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
                const uint32_t site = w & 0x3FFFFFu, mi = (w >> 22) & 3u, pi = (w >> 24) & 3u;
                const uint32_t x = tab[(uint64_t)site * 8];
                const uint32_t nib = (x >> sh) & 15u;
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
                if (MODE == 0) {
                    const uint32_t k = key | hu;
                    if (elig) {
                        if (cost < best) { best = cost; cnt = 1; bkey = k; }
                        else if (cost == best) { cnt++; bkey = max(bkey, k); }
                    }
                } else if (MODE == 1) {
                    const uint32_t bfs = a.dfs2bfs[node_idx];
                    if (q < a.n_queries) a.scores[(uint64_t)q * a.n_nodes + bfs] = (int32_t)(cost + (elig ? 0u : 1u));
                    node_idx++;
                } else {
                    const uint32_t bfs = a.dfs2bfs[node_idx];
                    if (elig && cost == want_best) {
                        const uint32_t i = atomicAdd(&a.tie_count[q], 1u);
                        if (i < a.tie_cap) {
                            a.tie_j[(uint64_t)q * a.tie_cap + i] = bfs;
                            a.tie_hu[(uint64_t)q * a.tie_cap + i] = (uint8_t)hu;
                        }
                    }
                    node_idx++;
                }
            }
        }
    }
    if (MODE == 0) {
        const uint64_t o = ((uint64_t)tile * a.n_groups + g) * 64 + lane;
        a.part_best[o] = best; a.part_cnt[o] = cnt; a.part_key[o] = bkey;
    }
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

// ---------------------------------------------------------------- launchers

hipError_t launch_fill_table(uint32_t *table, const uint8_t *site_ref, uint32_t n_sites, uint64_t total_dwords,
                             hipStream_t s) {
    if (total_dwords == 0) return hipSuccess;
    uint64_t blocks = (total_dwords + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(k_fill_table, dim3((uint32_t)blocks), dim3(256), 0, s, table, site_ref, n_sites, total_dwords);
    return hipGetLastError();
}

hipError_t launch_scatter(uint32_t *table, uint32_t *dbottom, const int32_t *pos, const uint8_t *ref,
                          const uint8_t *nuc, const uint8_t *is_missing, const uint32_t *ent_q,
                          const int32_t *pos2site, uint32_t max_pos, uint32_t n_sites, uint64_t n_ent, uint32_t q_base,
                          hipStream_t s) {
    if (n_ent == 0) return hipSuccess;
    uint64_t blocks = (n_ent + 255) / 256;
    hipLaunchKernelGGL(k_scatter_entries, dim3((uint32_t)blocks), dim3(256), 0, s, table, dbottom, pos, ref, nuc,
                       is_missing, ent_q, pos2site, max_pos, n_sites, n_ent, q_base);
    return hipGetLastError();
}

hipError_t launch_place(const PlaceArgs &a, int mode, uint32_t max_slots, hipStream_t s) {
    const uint32_t blocks = a.n_tiles * a.n_groups;
    const size_t lds = (size_t)max_slots * 64 * sizeof(uint32_t);
    if (mode == 0) hipLaunchKernelGGL(k_place<0>, dim3(blocks), dim3(64), lds, s, a);
    else if (mode == 1) hipLaunchKernelGGL(k_place<1>, dim3(blocks), dim3(64), lds, s, a);
    else hipLaunchKernelGGL(k_place<2>, dim3(blocks), dim3(64), lds, s, a);
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

}  // namespace ugp
