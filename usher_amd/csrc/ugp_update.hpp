// ugp_update.hpp -- device side of the add mode (default `usher`: every placement edits the tree, usher_common.cpp:652-765).
// The flattened tree on the device stays the one ugp_mat_create built; what an insertion changes is kept beside it:
//   * the nodes created or rewritten since ("touched": the new sample leaf, the new internal node, the split best_node) as
//     RECORDS -- own mutations plus the state of the parent along its root path -- scored against a batch of pending samples
//     by k_touched (closed form of mapper2_body, usher_mapper.cpp:167-504, one record x 64 samples per wave);
//   * the flattened nodes among them are taken out of the candidate set of every later search by setting one bit in their
//     words of the three record streams (k_or_words), so that a search of the flattened tree is exact over the nodes that are
//     still what they were, and the records are exact over the rest.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ugp {

constexpr uint32_t KEY_EXCLUDED = 1u << 31;   // bit 31 of a record's tie-key word (w1) in `stream` / `stream_t`: not a candidate any more
constexpr uint32_t T_LEAF = 1u, T_MASKED = 2u;   // record flags (ugp_touched::flags)

struct TouchedRec { uint32_t ent_off, n_path, n_own, flags; };   // entries [ent_off, ent_off + n_path + n_own)
struct TouchedEnt { int32_t pos; uint32_t bits; };                // bits = allele | prev << 8 | ref << 16

struct TouchedArgs {
    const TouchedRec *rec; const TouchedEnt *ent; const uint8_t *alive;
    uint32_t id0, id1;            // records [id0, id1)
    const uint8_t *dense;         // [n_pos][qpad]: allele mask of sample q at pos, 0 = no row (the reference base)
    uint32_t n_pos, qpad;
    const int32_t *dbot;          // [qpad] D(bottom) of the batch's samples
    uint32_t q0, q1;              // samples [q0, q1)
    int32_t *best;                // [qpad] running minimum over the eligible records seen so far (INT32_MAX: none)
    uint32_t *cnt;                // [qpad] records in the list (may exceed cap: the list is truncated, the count is true)
    uint32_t *ids; uint8_t *hu;   // [qpad][cap] the records that attain best[q], and their has_unique
    uint32_t cap;
};

hipError_t launch_or_words(uint32_t *stream, const uint32_t *pos, uint32_t n, uint32_t bits, hipStream_t s);
// tie rank of an extended search by a stable device radix sort of 64-bit keys (temp == nullptr: only *temp_bytes is filled)
hipError_t launch_rank_sort(void *temp, size_t *temp_bytes, const uint64_t *keys, uint64_t *keys_out, uint32_t *iota, uint32_t *rank2out, uint32_t n,
                            const uint32_t *to_bfs, uint32_t *rank_bfs, hipStream_t s);
// a node mask as a temporary exclusion (every node has one thread; each node's words are its own: plain read-modify-write)
hipError_t launch_mask_words(const uint8_t *mask, const uint32_t *map_j, uint32_t n, const uint32_t *hdr8, const uint32_t *rec, const uint32_t *post,
                             uint32_t *stream8, uint32_t *stream, uint32_t *stream_t, uint32_t bit8 /* H_NOSCORE */, bool set, hipStream_t s);
// rows of the batch into the dense table (zeroed by the caller) and D(bottom): one thread per row
hipError_t launch_dense_scatter(uint8_t *dense, uint32_t n_pos, uint32_t qpad, int32_t *dbot, const int32_t *pos, const uint8_t *ref, const uint8_t *nuc,
                                const uint8_t *is_missing, const uint32_t *ent_q, uint64_t n_ent, hipStream_t s);
// dst[i] = src[i], i < n: a copy by a kernel on the stream (dst may be pinned host memory: see ugp_place_batch_async)
hipError_t launch_copy_words(uint32_t *dst, const uint32_t *src, uint64_t n, hipStream_t s);
// per-node scores of an extended search that came from the unrestricted level-by-level kernel: the nodes its mask does not admit
// and each sample's excluded node read 0 ("not scored"), as the one-sample-per-lane kernel leaves them
hipError_t launch_scores_mask(int32_t *scores, uint64_t n_queries, uint64_t n_nodes, const uint8_t *mask /* by BFS index, or null */,
                              const uint32_t *skip /* [n_queries] BFS index or UINT32_MAX, or null */, hipStream_t s);
// pass 1 (minimum), the list reset where the minimum fell (list_best = the cost the list belongs to), pass 2 (append)
hipError_t launch_touched(const TouchedArgs &a, int32_t *list_best, hipStream_t s);

}  // namespace ugp
