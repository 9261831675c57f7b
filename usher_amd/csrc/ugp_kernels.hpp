// ugp_kernels.hpp -- launch interface between the C-ABI layer and the kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ugp_flatten.hpp"
#include "usher_amd.h"

namespace ugp {

struct PlaceArgs {
    const uint32_t *stream, *pre_stream;
    const uint32_t *chunk_body_off, *chunk_pre_off, *chunk_node_off;
    const uint32_t *table;     // [n_tiles][n_sites][8]
    const uint32_t *dbottom;   // [n_tiles*64]
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
};

hipError_t launch_fill_table(uint32_t *table, const uint8_t *site_ref, uint32_t n_sites, uint64_t total_dwords,
                             hipStream_t s);
hipError_t launch_scatter(uint32_t *table, uint32_t *dbottom, const int32_t *pos, const uint8_t *ref,
                          const uint8_t *nuc, const uint8_t *is_missing, const uint32_t *ent_q,
                          const int32_t *pos2site, uint32_t max_pos, uint32_t n_sites, uint64_t n_ent, uint32_t q_base,
                          hipStream_t s);
hipError_t launch_place(const PlaceArgs &a, int mode, uint32_t max_slots, hipStream_t s);
hipError_t launch_merge(const uint32_t *part_best, const uint32_t *part_cnt, const uint32_t *part_key,
                        const uint32_t *rank2bfs, uint32_t n_groups, uint32_t n_queries, ugp_result *out,
                        hipStream_t s);

}  // namespace ugp
