// driver.hpp -- the usher-compatible placement driver (host side of the drop-in):
// option validation, per-sample bookkeeping, tree update and every output file
// of the reference's usher_common() (src/usher_common.cpp:7-1073), with the
// node x sample search delegated to a placement backend (the GPU library).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "mat.hpp"
#include "usher_amd.h"

namespace uh {

struct Options {   // the 22 flags of src/usher.cpp:47-86
    std::string vcf, tree, outdir = ".", load_mat, save_mat;
    bool sort1 = false, sort2 = false, sort3 = false, reverse_sort = false;
    bool collapse_tree = false, collapse_output_tree = false;
    uint32_t max_uncertainty = 1000000, max_parsimony = 1000000;
    bool write_uncondensed = false;
    size_t subtrees_size = 0, subtrees_single = 0;
    bool print_scores = false;
    uint32_t max_trees = 1;
    bool retain_branch_len = false, no_add = false, detailed_clades = false;
    uint32_t threads = 0;
    int device = 0;   // extension: HIP device ordinal (--device)
    std::string devices;   // extension: device list to shard the samples across (--devices 0-7); parsed by the backend
};

// The search block of usher_common.cpp:342-449 for a batch of samples on a
// static tree.  Signatures mirror include/usher_amd.h; `ctx` is backend state.
struct Backend {
    void *ctx = nullptr;
    int (*place)(void *ctx, const ugp_tree_desc *, uint64_t tree_version, const ugp_queries *, ugp_result *) = nullptr;
    int (*scores)(void *ctx, const ugp_tree_desc *, uint64_t tree_version, const ugp_queries *, int32_t *) = nullptr;
    int (*ties)(void *ctx, const ugp_tree_desc *, uint64_t tree_version, const ugp_queries *, uint32_t cap, uint32_t *,
                uint8_t *, uint32_t *) = nullptr;
    const char *(*last_error)(void *ctx) = nullptr;
    // mapper_body (usher_mapper.cpp:6-161) for all VCF sites of a `-t` build: ugp_fitch_sankoff.
    // `fitch` computes and keeps the result in the backend, `fitch_get` copies it out and releases it.
    int (*fitch)(void *ctx, uint64_t n_nodes, const uint32_t *parent, const ugp_sites *sites, uint64_t *n_out) = nullptr;
    int (*fitch_get)(void *ctx, uint32_t *site, uint32_t *node, uint8_t *par_nuc, uint8_t *mut_nuc) = nullptr;
    // Add mode on the device (include/usher_amd.h, "add mode"): all optional -- without them the driver re-derives batched answers on
    // the host and flattens the tree again when too many nodes have changed.  They act on the tree most recently given to `place`.
    int (*update)(void *ctx, const ugp_touched *recs, const uint32_t *retired, uint64_t n_retired, uint32_t *first_id) = nullptr;
    int (*touched_open)(void *ctx, const ugp_queries *batch) = nullptr;
    int (*touched_score)(void *ctx, uint32_t first_id, uint64_t first_sample) = nullptr;
    int (*touched_rescore)(void *ctx, uint64_t sample) = nullptr;
    int (*touched_fetch)(void *ctx, uint64_t first_sample, uint64_t n, uint32_t cap, int32_t *best, uint32_t *count, uint32_t *ids, uint8_t *has_unique) = nullptr;
    // End-to-end latency of the drop-in (round 5), both optional.  `warm`: start whatever the backend needs before its first call (the
    // device runtime) in the background -- called first thing, while the inputs are still being read.  `prepare`: make this tree
    // resident now (what the first `place` with this version would do): the front end calls it on a thread of its own while the VCF is
    // read, so that flattening and upload are over when the samples arrive.
    void (*warm)(void *ctx) = nullptr;
    int (*prepare)(void *ctx, const ugp_tree_desc *, uint64_t tree_version) = nullptr;
};

// The tree as loaded, turned into the backend's arrays (and handed to Backend::prepare) on a thread of its own; run_usher takes it
// over at its first flattening if the tree is still what it was.  Null when there is nothing to gain (no `prepare`).
struct Prebuilt;
Prebuilt *prebuild_start(const Options &opt, const Tree &T, const Backend &be);
void prebuild_drop(Prebuilt *p);   // waits for the thread; for callers that never reach run_usher

// Returns the process exit code (usher_common.cpp:6).
int run_usher(const Options &opt, Tree &T, std::vector<MissingSample> &missing, const Backend &be, Prebuilt *pre = nullptr);

// mapper2_body(inp, true, true) for one node (usher_mapper.cpp:167-504): the
// excess / imputed mutation vectors, score and has_unique the placement needs.
struct NodeVecs {
    std::vector<Mutation> excess, imputed;
    int set_difference = 0;
    bool has_unique = false;
    bool eligible = false;   // the placement predicate of usher_mapper.cpp:454-455
};
void node_vecs(const Node *n, const std::vector<Mutation> &sample, NodeVecs &out);

}  // namespace uh
