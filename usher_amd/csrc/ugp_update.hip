// ugp_update.hip -- gfx950 kernels of the add mode: scoring of the nodes created or rewritten since the tree was flattened
// against a batch of pending samples, and the exclusion of rewritten nodes from the flattened tree's candidate set.
// See ugp_update.hpp.  Integer work, lanes = samples, record data wave-uniform.
#include "ugp_update.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <climits>

namespace ugp {

__global__ void k_or_words(uint32_t *__restrict__ stream, const uint32_t *__restrict__ pos, uint32_t n, uint32_t bits) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && pos[i] != 0xFFFFFFFFu) atomicOr(&stream[pos[i]], bits);   // (a node may be listed twice in one call)
}

hipError_t launch_or_words(uint32_t *stream, const uint32_t *pos, uint32_t n, uint32_t bits, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_or_words, dim3((n + 255) / 256), dim3(256), 0, s, stream, pos, n, bits);
    return hipGetLastError();
}

// Tie rank of an extended search: caller indices 0..n-1 sorted (stably) by a 64-bit key, worst first; rank2out[r] = caller index of
// rank r, rank_bfs[BFS index of that node] = r (to_bfs: caller index -> BFS index, or null = the same).
__global__ void k_iota(uint32_t *__restrict__ v, uint32_t n) { const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) v[i] = i; }
__global__ void k_rank_scatter(const uint32_t *__restrict__ rank2out, const uint32_t *__restrict__ to_bfs, uint32_t n, uint32_t *__restrict__ rank_bfs) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) { const uint32_t k = rank2out[r]; rank_bfs[to_bfs ? to_bfs[k] : k] = r; }
}
hipError_t launch_rank_sort(void *temp, size_t *temp_bytes, const uint64_t *keys, uint64_t *keys_out, uint32_t *iota, uint32_t *rank2out, uint32_t n,
                            const uint32_t *to_bfs, uint32_t *rank_bfs, hipStream_t s) {
    if (!temp) return rocprim::radix_sort_pairs(nullptr, *temp_bytes, keys, keys_out, iota, rank2out, (size_t)n, 0u, 64u, s);
    hipLaunchKernelGGL(k_iota, dim3((n + 255) / 256), dim3(256), 0, s, iota, n);
    hipError_t e = rocprim::radix_sort_pairs(temp, *temp_bytes, keys, keys_out, iota, rank2out, (size_t)n, 0u, 64u, s);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_rank_scatter, dim3((n + 255) / 256), dim3(256), 0, s, rank2out, to_bfs, n, rank_bfs);
    return hipGetLastError();
}

// A node mask shared by all samples of one extended search (ugp_place_opts::node_mask) as a temporary exclusion: the words of every
// node the mask leaves out get (set != 0) or lose the "no candidate" bit.  i = index in this tree, map_j[i] = its index in the mask's
// tree (the coarse tree of the locality pre-pass: its node's index in the full tree; null: the same).  The root is left alone.
__global__ void k_mask_words(const uint8_t *__restrict__ mask, const uint32_t *__restrict__ map_j, uint32_t n, const uint32_t *__restrict__ hdr8,
                             const uint32_t *__restrict__ rec, const uint32_t *__restrict__ post, uint32_t *__restrict__ stream8, uint32_t *__restrict__ stream,
                             uint32_t *__restrict__ stream_t, uint32_t bit8, int set) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0 || i >= n || mask[map_j ? map_j[i] : i]) return;
    // (a node that carries a masked mutation has that bit in its packed header already -- it is never eligible -- and keeps it)
    const bool own8 = (stream[rec[i]] & (1u << 31)) != 0;   // F_MASKED of the 32-bit record
    if (own8) hdr8 = nullptr;
    if (!hdr8) {
        if (set) { stream[rec[i] + 1u] |= KEY_EXCLUDED; if (post[i] != 0xFFFFFFFFu) stream_t[post[i] + 1u] |= KEY_EXCLUDED; }
        else { stream[rec[i] + 1u] &= ~KEY_EXCLUDED; if (post[i] != 0xFFFFFFFFu) stream_t[post[i] + 1u] &= ~KEY_EXCLUDED; }
        return;
    }
    if (set) {
        if (hdr8[i] != 0xFFFFFFFFu) stream8[hdr8[i]] |= bit8;
        stream[rec[i] + 1u] |= KEY_EXCLUDED;
        if (post[i] != 0xFFFFFFFFu) stream_t[post[i] + 1u] |= KEY_EXCLUDED;
    } else {
        if (hdr8[i] != 0xFFFFFFFFu) stream8[hdr8[i]] &= ~bit8;
        stream[rec[i] + 1u] &= ~KEY_EXCLUDED;
        if (post[i] != 0xFFFFFFFFu) stream_t[post[i] + 1u] &= ~KEY_EXCLUDED;
    }
}

hipError_t launch_mask_words(const uint8_t *mask, const uint32_t *map_j, uint32_t n, const uint32_t *hdr8, const uint32_t *rec, const uint32_t *post,
                             uint32_t *stream8, uint32_t *stream, uint32_t *stream_t, uint32_t bit8, bool set, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_mask_words, dim3((n + 255) / 256), dim3(256), 0, s, mask, map_j, n, hdr8, rec, post, stream8, stream, stream_t, bit8, set ? 1 : 0);
    return hipGetLastError();
}

// dense[pos][q] = allele mask of the sample's row at pos (15 for a missing call); D(bottom) = rows whose set excludes the
// reference base (usher_mapper.cpp:292-388 with an empty ancestral list).  Rows are unique per (sample, position).
__global__ void k_dense_scatter(uint8_t *__restrict__ dense, uint32_t n_pos, uint32_t qpad, int32_t *__restrict__ dbot, const int32_t *__restrict__ pos,
                                const uint8_t *__restrict__ ref, const uint8_t *__restrict__ nuc, const uint8_t *__restrict__ is_missing,
                                const uint32_t *__restrict__ ent_q, uint64_t n_ent) {
    const uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_ent) return;
    const int32_t p = pos[e];
    const uint32_t q = ent_q[e];
    const uint32_t a = is_missing[e] ? 15u : (uint32_t)nuc[e];
    if (!is_missing[e] && (a & ref[e]) == 0) atomicAdd(&dbot[q], 1);
    if (p >= 0 && (uint32_t)p < n_pos) dense[(uint64_t)p * qpad + q] = (uint8_t)a;
}

hipError_t launch_dense_scatter(uint8_t *dense, uint32_t n_pos, uint32_t qpad, int32_t *dbot, const int32_t *pos, const uint8_t *ref, const uint8_t *nuc,
                                const uint8_t *is_missing, const uint32_t *ent_q, uint64_t n_ent, hipStream_t s) {
    if (!n_ent) return hipSuccess;
    hipLaunchKernelGGL(k_dense_scatter, dim3((uint32_t)((n_ent + 255) / 256)), dim3(256), 0, s, dense, n_pos, qpad, dbot, pos, ref, nuc, is_missing, ent_q, n_ent);
    return hipGetLastError();
}

// One wave = 64 samples x a run of records.  For record x and sample s (closed form, SURVEY 8a / DESIGN 2):
//   D(parent) = D(bottom) + sum over the positions where the parent's state is not the reference base of
//               ([state not in S] - [ref not in S]);   cost = D(parent) + sum over own mutations of min(delta, 0)
//   eligible  = common > 0 or (internal and no mutations);   has_unique = masked or common != #mutations
// The entries of a record are read through scalar loads (the record index is wave-uniform); a lane's share is one byte of the
// dense table per entry -- 64 consecutive bytes per wave.
template <int PASS>
__global__ void __launch_bounds__(64) k_touched(TouchedArgs a, uint32_t T_RECS_PER_BLOCK) {
    const uint32_t lane = threadIdx.x;
    const uint32_t q = a.q0 + blockIdx.y * 64u + lane;
    const bool in = q < a.q1;
    const uint32_t qc = in ? q : a.q0;   // (lanes past the end read a valid column and write nothing)
    const uint32_t id_b = a.id0 + blockIdx.x * T_RECS_PER_BLOCK, id_e = min(a.id1, id_b + T_RECS_PER_BLOCK);
    const int dbot = a.dbot[qc];
    const int want = PASS == 2 ? a.best[qc] : 0;
    int lmin = INT_MAX;
    for (uint32_t id = id_b; id < id_e; id++) {
        if (!a.alive[id]) continue;
        const TouchedRec r = a.rec[id];
        const TouchedEnt *e = a.ent + r.ent_off;
        // (round 6) eight entries at a time: their table bytes are independent loads, all in flight together -- one after the other
        // (a scalar load of the entry, then the byte it names, then the next entry) a record of ~40 entries was ~40 round trips, and the
        // driver waits for this kernel once per round of 64 insertions (0.43 ms, 0.67 s per 100 000 insertions)
        int D = dbot, neg = 0;
        uint32_t common = 0;
        const uint32_t n_all = r.n_path + r.n_own;
        for (uint32_t k0 = 0; k0 < n_all; k0 += 8u) {
            uint32_t rows[8], bits8[8];
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t kk = min(k0 + (uint32_t)j, n_all - 1u);
                const int32_t p = e[kk].pos;
                bits8[j] = e[kk].bits;
                rows[j] = (uint32_t)p < a.n_pos ? (uint32_t)a.dense[(uint64_t)(uint32_t)p * a.qpad + qc] : 0u;
            }
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t k = k0 + (uint32_t)j;
                if (k >= n_all) break;
                const uint32_t bits = bits8[j], al = bits & 15u, pv = (bits >> 8) & 15u, rf = (bits >> 16) & 15u;
                const uint32_t sp = rows[j] ? rows[j] : rf;
                if (k < r.n_path) D += (int)((sp & al) == 0) - (int)((sp & rf) == 0);   // the parent's state where it is not the reference base
                else {                                                                   // the node's own mutations
                    const int c = (sp & al) != 0, pr = (sp & pv) != 0;
                    common += (uint32_t)c;
                    neg += min(pr - c, 0);
                }
            }
        }
        const bool masked = (r.flags & T_MASKED) != 0;
        const uint32_t num_mut = r.n_own + (masked ? 1u : 0u);
        const int cost = D + neg;
        const bool elig = common > 0 || (!(r.flags & T_LEAF) && num_mut == 0);
        if (PASS == 1) { if (elig) lmin = min(lmin, cost); }
        else if (in && elig && cost == want) {
            const uint32_t i = atomicAdd(&a.cnt[q], 1u);
            if (i < a.cap) { a.ids[(uint64_t)q * a.cap + i] = id; a.hu[(uint64_t)q * a.cap + i] = (masked || common != num_mut) ? 1 : 0; }
        }
    }
    if (PASS == 1 && in && lmin != INT_MAX) atomicMin(&a.best[q], lmin);
}

// Between the passes: where the minimum fell below the cost the sample's list was built for, the list starts over.
__global__ void k_touched_reset(const int32_t *__restrict__ best, int32_t *__restrict__ list_best, uint32_t *__restrict__ cnt, uint32_t q0, uint32_t q1) {
    const uint32_t q = q0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= q1) return;
    if (best[q] < list_best[q]) { list_best[q] = best[q]; cnt[q] = 0; }
}

hipError_t launch_touched(const TouchedArgs &a, int32_t *list_best, hipStream_t s) {
    if (a.id1 <= a.id0 || a.q1 <= a.q0) return hipSuccess;
    // Records per wave.  The driver WAITS for these kernels once per round of insertions, and a wave takes its records one after the
    // other (five rounds of loads each): 16 per wave were 0.3 ms per round; the few hundred records of a round spread over more waves
    // cost more atomics on best[] and nothing else.  (UGP_TOUCHED_RECS: 1..64, for measurements.)
    static const uint32_t T_RECS_PER_BLOCK = [] { const char *e = getenv("UGP_TOUCHED_RECS"); const int v = e ? atoi(e) : 2; return (uint32_t)std::min(64, std::max(1, v)); }();
    const dim3 grid((a.id1 - a.id0 + T_RECS_PER_BLOCK - 1) / T_RECS_PER_BLOCK, (a.q1 - a.q0 + 63) / 64);
    hipLaunchKernelGGL(k_touched<1>, grid, dim3(64), 0, s, a, T_RECS_PER_BLOCK);
    hipLaunchKernelGGL(k_touched_reset, dim3((a.q1 - a.q0 + 255) / 256), dim3(256), 0, s, a.best, list_best, a.cnt, a.q0, a.q1);
    hipLaunchKernelGGL(k_touched<2>, grid, dim3(64), 0, s, a, T_RECS_PER_BLOCK);
    return hipGetLastError();
}

__global__ void k_scores_mask(int32_t *__restrict__ scores, uint64_t n_queries, uint64_t n_nodes, const uint8_t *__restrict__ mask) {
    for (uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_nodes; j += (uint64_t)gridDim.x * blockDim.x) {
        if (mask[j]) continue;
        for (uint64_t q = 0; q < n_queries; q++) scores[q * n_nodes + j] = 0;
    }
}
__global__ void k_scores_skip(int32_t *__restrict__ scores, uint64_t n_queries, uint64_t n_nodes, const uint32_t *__restrict__ skip) {
    const uint64_t q = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n_queries && skip[q] != 0xFFFFFFFFu) scores[q * n_nodes + skip[q]] = 0;
}
hipError_t launch_scores_mask(int32_t *scores, uint64_t n_queries, uint64_t n_nodes, const uint8_t *mask, const uint32_t *skip, hipStream_t s) {
    if (!n_queries || !n_nodes) return hipSuccess;
    if (mask) hipLaunchKernelGGL(k_scores_mask, dim3((uint32_t)std::min<uint64_t>((n_nodes + 255) / 256, 65536)), dim3(256), 0, s, scores, n_queries, n_nodes, mask);
    if (skip) hipLaunchKernelGGL(k_scores_skip, dim3((uint32_t)((n_queries + 255) / 256)), dim3(256), 0, s, scores, n_queries, n_nodes, skip);
    return hipGetLastError();
}

__global__ void k_copy_words(uint32_t *__restrict__ dst, const uint32_t *__restrict__ src, uint64_t n) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
hipError_t launch_copy_words(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t s) {
    if (!n) return hipSuccess;
    hipLaunchKernelGGL(k_copy_words, dim3((uint32_t)std::min<uint64_t>((n + 63) / 64, 4096)), dim3(64), 0, s, dst, src, n);
    return hipGetLastError();
}

}  // namespace ugp
