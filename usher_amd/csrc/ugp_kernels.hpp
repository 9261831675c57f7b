// ugp_kernels.hpp -- launch interface between the C-ABI layer and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugp_flatten.hpp"
#include "usher_amd.h"

namespace ugp {

constexpr uint32_t TABLE_CONST_ROWS = 4;   // rows 0..3 of every tile of the allele table: all samples carry A / C / G / T

struct PlaceArgs {
    const uint32_t *stream, *pre_stream;
    const uint32_t *chunk_body_off, *chunk_pre_off, *chunk_node_off;
    const uint32_t *stream_t, *chunk_t_off;   // tie stream (phase 2)
    const uint32_t *table;     // [ceil(n_tiles/8)][4 + n_sites][64]  (512-sample tile layout behind 4 constant rows)
    const uint32_t *dbottom;   // [ceil(n_tiles/8)*512]
    uint32_t n_sites, n_chunks, n_groups, n_tiles, n_queries;
    // MODE 0
    uint32_t *part_best, *part_cnt, *part_key;   // [n_tiles][n_groups][64]
    // MODE 1 / 2
    const uint32_t *dfs2bfs;
    uint64_t n_nodes;
    int32_t *scores;           // [n_queries][n_nodes]
    const int32_t *best_in;    // [n_queries]
    uint32_t *tie_count;       // [n_tiles*64]
    uint32_t *tie_j;           // [n_queries][tie_cap]
    uint8_t *tie_hu;
    uint32_t tie_cap;
    // extended search (ugp_place_batch_ex / ugp_tied_nodes_ex: the other callers of mapper2_body), all optional
    const uint8_t *node_mask;    // [n_nodes] by BFS index: 0 = not a candidate
    const uint32_t *skip;        // [n_queries] BFS index of one node excluded for that sample (UINT32_MAX = none)
    const uint32_t *alt_rank;    // [n_nodes] by BFS index: tie rank replacing the stream's (n_leaves, bfs_j) rank
    const uint32_t *out_index;   // [n_nodes] BFS index -> index reported to the caller (its node order)
};

struct Best8Args {
    const uint32_t *stream8, *pre8;
    const uint32_t *chunk8_body_off, *chunk8_pre_off;   // [n_chunks+1]
    const uint32_t *table;     // [n_tiles][4 + n_sites][64]
    const uint32_t *dbottom;   // [n_tiles*512]
    uint32_t n_sites, n_chunks, n_groups, n_tiles;   // n_tiles = 512-sample tiles
    uint32_t max_slots;
    uint32_t lds_slots;        // saved-D slots kept in LDS (1 KB each per wave); the colder ones live in `cold`
    uint32_t *cold;            // [resident waves][max_slots - lds_slots][64][4]
    const uint32_t *active;    // [n_tiles][active_words] bit per site: some sample of the tile is not reference there
    uint32_t active_words;
    uint32_t *lbest;           // [n_chunks][n_tiles][64][4] packed u16 pairs; a record exists only for the chunks in `list`
    uint32_t *luniq;           // (main walk, round 6) or null: [n_chunks][n_tiles][64] BYTES beside every lbest record -- byte l, bit j + 4h set: sample j + 4h of lane l
                               // has MORE than one node at its chunk minimum (or no candidate); clear = the minimum is attained by exactly one node
    uint32_t *lpos;            // (coarse pass) or null: same layout as lbest -- low 16 bits of the stream position of the node that set each chunk minimum
    uint32_t *list, *list_n;   // [n_tiles][n_chunks] chunks of each tile that left a record, [n_tiles] their number (zeroed before the launch)
    uint32_t *queue;           // [8] work-queue heads, one per XCD, zeroed before the launch
    const uint4 *units;        // work units {tile, c0, c1, flags}, one list per queue (k_build_units)
    const uint32_t *unit_base, *unit_count;    // [8] first entry / number of entries of each queue's list
    uint32_t *dyn_ctl;         // head, tail, active and waiting waves of the shared list of split-off units (one 128-byte line each)
    unsigned long long *dyn_units;   // [dyn_cap] {epoch:11 | own region:1 | tile:12 | c1:20 | c0:20}: an entry counts once it carries this launch's epoch
    uint32_t dyn_cap, dyn_epoch;
    uint32_t lds_bits;         // 1: the kernel variant that keeps the tile's active-row bitmap in LDS; 2: the variant without a bitmap (every word fetches its own row)
    uint32_t no_pre_records;   // the preamble replay ignores its pruning records (units longer than their jump field reaches)
    uint32_t split_heavy;      // the same for the units of the tiles' own regions (dense: both halves are real work)
    uint32_t split_dense;      // ... and only a unit that closed at most this many chunks since its last look is cut
    uint32_t split_many;       // != 0: a cut hands out as many pieces as waves wait (up to 63), each of at least this many chunks; 0: one half
    uint32_t split_cycles;     // a unit running longer than this hands half of its remainder to the shared list when waves wait for work (0xFFFFFFFF: never)
    uint32_t ub_every;         // exchange the shared upper bounds at every ub_every-th chunk end
    uint32_t freeze_ub;        // the bounds stay what they were seeded with (a search that leaves a node out per sample: the chunk minima include it)
    uint32_t refill_all_rows;  // experiment (UGP_REFILL_ALL): a refill fetches the real row of every word of its first group
    uint32_t heavy_prio;       // raise the wave priority while a unit of a tile's own region is walked
    uint64_t *trace;           // optional (with stats): [0] = records written, records of 6 words from [8] on; trace_cap = room for that many
    uint64_t trace_cap;
    uint64_t *stats;           // optional: [0] += stream words skipped by pruning (debug / bench)
    uint32_t *ub;              // [n_tiles][64][4] packed upper bounds of best(s) shared by the waves of a tile; nullptr = no pruning
    // phase 2 on the packed path (k_best8<..., TIES>): ub = the samples' global minima; every node whose cost equals a sample's is counted
    uint32_t *tie_cnt, *tie_key;       // [n_queries] number of such nodes / largest (rank << 1 | has_unique) among them; null: not this mode
    const uint32_t *node_pos8;         // [n_nodes + 1] by DFS index: stream position where the node's words begin
    const uint32_t *rank_dfs;          // [n_nodes] by DFS index: tie rank
    const uint32_t *chunk_node_off;    // [n_chunks + 1] first DFS index of every chunk
    uint32_t n_queries;
    const struct B3Dev *b3;            // third pruning bound (ugp_bound3.hpp): the tile tables of this launch, or null (the kernel variant without it)
};

// hstart / hlen: [n_tiles] or null: per tile, the first chunk of the region its own samples sit in and its length in chunks
hipError_t launch_build_units(const uint32_t *hstart, const uint32_t *hlen, uint32_t n_tiles512, uint32_t n_chunks, uint32_t unit_chunks, uint32_t heavy_chunks,
                              uint32_t grow_every, uint32_t unit_max, uint32_t light_order, uint32_t per_tile_cap, void *units, uint32_t *unit_base,
                              uint32_t *unit_count, uint32_t *dyn_ctl, hipStream_t s);
constexpr uint32_t SCORES_SB = 32;   // samples per step of k_scores_level (a multiple of 32): the D arrays are laid out in blocks of this many
// -p by levels of the breadth-first expansion (see k_scores_level): d_a / d_b = two D arrays of max level width x qpad (rounded up to SCORES_SB) entries
hipError_t launch_scores_levels(const uint32_t *node_pair, const uint32_t *parent, const uint32_t *stream, const uint32_t *table, uint32_t n_sites,
                                const uint32_t *dbottom, const uint32_t *level_off, uint32_t n_levels, void *d_a, void *d_b, bool d16, uint32_t d_stride,
                                uint32_t qpad, uint32_t n_queries, uint64_t n_nodes, int32_t *scores, uint32_t block /* 0: default */, hipStream_t s);
hipError_t best8_occupancy(size_t lds_bytes, int variant /* 0 main, 1 LDS bitmap, 2 coarse pass, 3 / 4 main / coarse without a bitmap */, int *per_cu);
hipError_t launch_best8(const Best8Args &a, uint32_t blocks, hipStream_t s);
// a.n_tiles = number of 64-sample tiles; lbest/gbest in the packed 512-tile layout
constexpr uint32_t GBEST_SLICES = 64;   // chunk-axis slices of the global-minimum reduction
// gbest_part: [GBEST_SLICES][n_tiles512*256] scratch
hipError_t launch_coarse_result(const uint32_t *lbest, const uint32_t *lpos, const uint32_t *list, const uint32_t *list_n, uint32_t n_chunks, uint32_t n_tiles512,
                                uint32_t n_queries, const uint32_t *chunk_node_off, const uint32_t *chunk8_body_off, const uint32_t *node_pos8,
                                const uint32_t *dfs2bfs, ugp_result *out, hipStream_t s);
hipError_t launch_phase2_packed(const Best8Args &b1, const uint32_t *list, const uint32_t *list_n, uint32_t *gbest_part, uint32_t *gbest, uint32_t n_tiles512,
                                void *units, uint32_t *info, uint32_t *cnt, uint32_t *key, const uint32_t *node_pos8, const uint32_t *rank_dfs,
                                const uint32_t *chunk_node_off, const uint32_t *rank2bfs, uint32_t n_queries, ugp_result *out, const uint32_t *order,
                                uint32_t blocks, hipStream_t s);
// Phase 2 without a walk for the samples whose minimum is attained by ONE node (round 6): pass 2 of the reference visits the tied nodes
// only (usher_common.cpp:416-449); a sample whose global minimum lies in one recorded chunk (gcnt), at one node of it (luniq, written by
// k_best8), at the cost of the node the seed descent found (dnode / refined by sorted slot; dres = the same cost packed like gbest), is
// answered from that node -- k_select leaves it out of the (chunk, sub-tile) pairs k_ties walks, k_final takes the node.
struct Phase2Uniq {
    const uint32_t *luniq, *dnode, *refined, *dres;
    uint32_t *gcnt /* [n_tiles512 * 256] */, *gcnt_part /* [GBEST_SLICES][n_tiles512 * 256] */;
};
hipError_t launch_phase2(const PlaceArgs &a, const uint32_t *lbest, const uint32_t *list, const uint32_t *list_n, uint32_t *gbest_part,
                         uint32_t *gbest, uint32_t n_tiles512,
                         uint32_t *items, uint32_t *n_items, uint32_t cap, uint32_t *cnt, uint32_t *key,
                         const uint32_t *rank2bfs, const uint32_t *rank2out /* tie key -> index reported (extended searches), or null = rank2bfs */,
                         ugp_result *out, const uint32_t *order, uint32_t max_slots, bool lists /* also a.tie_* */,
                         const Phase2Uniq *uniq /* or null */, hipStream_t s);

// row checks of k_rows_prepare: *err = (row << 3) | kind of the first offending row, ~0 when clean
enum { ROWS_UNSORTED = 1, ROWS_BAD_REF = 2, ROWS_BAD_MASK = 3, ROWS_REF_MISMATCH = 4 };
hipError_t launch_rows_prepare(const uint64_t *ent_off, uint32_t n_queries, uint64_t n_ent, const int32_t *pos, const uint8_t *ref,
                               const uint8_t *nuc, const uint8_t *is_missing, const int32_t *pos2site, const uint8_t *site_ref,
                               uint32_t max_pos, uint32_t n_sites, uint32_t *ent_q, unsigned long long *err, hipStream_t s);
hipError_t launch_fill_table(uint32_t *table, const uint8_t *site_ref, uint32_t n_sites, uint64_t total_dwords,
                             hipStream_t s);
// the whole tile build (fill + scatter + row counts) for batches with many rows per sample: (tile, 128-site block) in LDS
hipError_t launch_build_tiles(uint32_t *table, uint32_t *active, uint32_t active_words, uint32_t n_tiles512, const uint64_t *ent_off, uint32_t q0,
                              const uint32_t *order /* slot -> sample of the sub-batch, or null */, uint32_t nq, const int32_t *pos, const uint8_t *ref,
                              const uint8_t *nuc, const uint8_t *is_missing, const int32_t *pos2site, const int32_t *site_pos, const uint8_t *site_ref,
                              uint32_t n_sites, uint32_t max_pos, uint32_t *dbottom, const unsigned long long *err /* k_rows_prepare's verdict, or null */,
                              hipStream_t s);
hipError_t launch_scatter(uint32_t *table, uint32_t *dbottom, const int32_t *pos, const uint8_t *ref,
                          const uint8_t *nuc, const uint8_t *is_missing, const uint32_t *ent_q,
                          const int32_t *pos2site, uint32_t max_pos, uint32_t n_sites, uint64_t n_ent, uint32_t q_base,
                          uint32_t *active, uint32_t active_words, const uint32_t *slot_of, const unsigned long long *err,
                          uint32_t *useful /* third bound: [tiles][useful_words] nibble per site, or null */, uint32_t useful_words, hipStream_t s);
// tiles of batches with many missing rows: N bits per (sample, site) built once per query set, tiles transposed from them
hipError_t launch_nmask_build(const uint64_t *ent_off, uint32_t n_queries, const int32_t *pos, const uint8_t *is_missing, const int32_t *pos2site,
                              uint32_t max_pos, uint32_t words, uint32_t *nmask, uint32_t *plain_rows /* or null */, uint32_t *n_plain, hipStream_t s);
hipError_t launch_ntiles(uint32_t *table, uint32_t *active, uint32_t active_words, uint32_t n_tiles512, const uint32_t *nmask, uint32_t words,
                         const uint32_t *order, uint32_t q0, uint32_t nq, const uint8_t *site_ref, uint32_t n_sites, hipStream_t s);
hipError_t launch_scatter_list(uint32_t *table, uint32_t *dbottom, const int32_t *pos, const uint8_t *ref, const uint8_t *nuc, const uint8_t *is_missing,
                               const uint32_t *ent_q, const int32_t *pos2site, uint32_t max_pos, uint32_t n_sites, uint32_t q_base, uint32_t n_q,
                               uint32_t *active, uint32_t active_words, const uint32_t *slot_of, const uint32_t *row_list, const uint32_t *n_listed,
                               const unsigned long long *err, uint32_t *useful, uint32_t useful_words, hipStream_t s);
// locality sort (see k_sort_keys); temp == nullptr: only *temp_bytes is filled
hipError_t launch_tile_ranges(const uint32_t *keys_sorted, uint32_t n_queries, uint32_t n_tiles512, const uint32_t *chunk_node_off,
                              uint32_t n_chunks, uint32_t align, uint32_t *hstart, uint32_t *hlen, hipStream_t s);
// greedy descent from the coarse best node: refined[slot] = smallest cost of an eligible node met (a valid upper bound of best(s))
hipError_t launch_descend(const ugp_result *coarse_res, const uint32_t *order, uint32_t n_queries, const uint32_t *coarse2bfs,
                          const uint32_t *node_pair /* [n_nodes + 1][2]: child_begin, rec_off */, const uint32_t *parent, const uint32_t *stream,
                          const uint32_t *table, uint32_t n_sites, uint32_t *refined, bool wide /* a whole wave per sample: trees with large polytomies */,
                          uint32_t max_expansions /* 0: default */, int slack, const uint32_t *skip /* [n_queries] by sample, or null: BFS index of a node whose cost is no bound for that sample */,
                          uint32_t *dnode /* or null: [n_queries] by sorted slot, (BFS index << 1 | has_unique) of a node whose cost is refined[slot]; UINT32_MAX: none.  Trees of < 2^31 nodes */,
                          hipStream_t s);
hipError_t launch_seed_ub(const ugp_result *coarse_res, const uint32_t *order, uint32_t n_queries, uint32_t n_tiles512, uint32_t *ub,
                          const uint32_t *refined /* [n_queries] by sorted slot, or null */,
                          uint32_t *dbottom /* or null: D(bottom) of the unused slots of the last tile is set to pad_d, their bound to 0 */, uint32_t pad_d,
                          const uint32_t *skip /* as launch_descend */, const uint32_t *coarse2bfs,
                          const uint32_t *dnode /* as launch_descend, or null */, uint32_t *dres /* [n_tiles512 * 256] packed like ub: the descent's cost where it names a node, else 0xFFFF */,
                          hipStream_t s);
// skip_node on the packed path: recompute, without the sample's excluded node, the minimum of the chunk that holds it (k_fix_skip)
hipError_t launch_fix_skip(const PlaceArgs &a, uint32_t *lbest, const uint32_t *skip_chunk /* [n_queries] by sample */, uint32_t n_tiles512, const uint32_t *rank2bfs,
                           const uint32_t *order, uint32_t max_slots, hipStream_t s);
// coarse_bin / n_bins / bins: position of every coarse node among the coarse nodes in depth-first order and n_bins counters, or
// null -- with them the samples are sorted by a counting sort (k_lsort_*: three small launches) instead of the device radix sort
hipError_t launch_locality_sort(const ugp_result *coarse_res, const uint32_t *coarse2dfs, uint32_t n, uint32_t *keys,
                                uint32_t *keys_sorted, uint32_t *idx, uint32_t *order, uint32_t *slot_of, void *temp,
                                size_t *temp_bytes, const uint32_t *coarse_bin, uint32_t n_bins, uint32_t *bins, hipStream_t s);
hipError_t launch_extract_best(const ugp_result *res, uint32_t n, int32_t *best, hipStream_t s);
hipError_t launch_place(const PlaceArgs &a, int mode, uint32_t max_slots, hipStream_t s);   // mode 0/1/2, +4: extended (a.node_mask etc.)
hipError_t launch_merge(const uint32_t *part_best, const uint32_t *part_cnt, const uint32_t *part_key,
                        const uint32_t *rank2bfs, uint32_t n_groups, uint32_t n_queries, ugp_result *out,
                        hipStream_t s);

}  // namespace ugp
