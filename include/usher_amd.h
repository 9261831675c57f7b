/*
 * usher_amd.h -- C ABI of the MI355X placement engine (libusher_amd.so).
 *
 * Drop-in boundary for UShER's parsimony-placement hot path.  The reference
 * has no FFI layer: the seam is the C++ call
 *     void mapper2_body(mapper2_input&, bool, bool)      src/usher_graph.hpp:103
 * made N times per sample from three tbb::parallel_for blocks in
 * usher_common()                                         src/usher_common.cpp:252-273, 389-414, 426-449
 * (4 more call sites in matUtils / ripples).  This library replaces the whole
 * per-sample block (usher_common.cpp:342-449) for a BATCH of samples against a
 * static tree: the caller hands over the tree once (ugp_mat_create), then any
 * number of query batches (ugp_place_batch & friends).  Plain pointers and
 * sizes only; no C++ or torch types cross the boundary; nothing is thrown and
 * the process is never exited (the reference's loaders call exit(1),
 * mutation_annotated_tree.cpp:473-475, 894-896, 2142-2145).
 *
 * Node numbering.  Every node index in this API is the node's position in the
 * reference's breadth-first expansion (Tree::breadth_first_expansion,
 * mutation_annotated_tree.cpp:1225-1251) -- the `j` of usher_common.cpp:391-403
 * and mapper2_input::j (usher_graph.hpp:82).  Tie-breaking among equally
 * parsimonious nodes (usher_mapper.cpp:483-486) is by larger number of
 * descendant leaves, then larger j, exactly as the reference.
 *
 * Alleles are the reference's one-hot codes A=1 C=2 G=4 T=8, ambiguity codes
 * are unions, N = 15 (mutation_annotated_tree.cpp:19-74).
 */
#ifndef USHER_AMD_H
#define USHER_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UGP_OK               0
#define UGP_ERR_INVALID     -1   /* bad argument / malformed tree or query arrays            */
#define UGP_ERR_UNSUPPORTED -2   /* input violates a precondition of the device algorithm    */
#define UGP_ERR_HIP         -3   /* a HIP runtime call failed (no device, out of memory ...)  */
#define UGP_ERR_NOMEM       -4   /* host allocation failed                                    */

/* Opaque handle: the flattened mutation-annotated tree resident in the HBM of
 * one device (replaces MAT::Tree* + the BFS vector of usher_common.cpp:342).
 * Threading: a handle owns its per-call workspaces, so calls on ONE handle must come from one thread at a time (the
 * reference serialises samples behind its global locks the same way, usher_mapper.cpp:3-4); different handles -- on the
 * same or on different devices -- are independent and may be driven by different host threads.  Overlap of batches on one
 * handle is what ugp_place_device / ugp_place_batch_async are for.  ugp_last_error() is per thread. */
typedef struct ugp_mat ugp_mat;
/* Opaque handle: a validated query batch resident in HBM. */
typedef struct ugp_qset ugp_qset;

/*
 * The tree, as flat arrays in BFS order (replaces MAT::Node / MAT::Mutation,
 * mutation_annotated_tree.hpp:45-111).  Children of a node are the nodes that
 * name it as parent, in increasing index (= the reference's child order).
 */
typedef struct ugp_tree_desc {
    uint64_t n_nodes;
    const uint32_t *parent;   /* [n_nodes]   parent[0] = UINT32_MAX, parent[j] < j            */
    const uint64_t *mut_off;  /* [n_nodes+1] CSR offsets into the mutation arrays             */
    const int32_t *mut_pos;   /* [n_muts]    1-based position; < 0 = masked (hpp:76-78)       */
    const uint8_t *mut_ref;   /* [n_muts]    one-hot reference base (ignored when masked)     */
    const uint8_t *mut_par;   /* [n_muts]    parent allele as stored in the MAT (informational:
                                             the library derives the true parent state itself,
                                             like usher_mapper.cpp:275-286 does)              */
    const uint8_t *mut_nuc;   /* [n_muts]    one-hot mutated base                             */
} ugp_tree_desc;

/*
 * A batch of query samples: per sample the rows read_vcf() would have put in
 * Missing_Sample::mutations (mutation_annotated_tree.cpp:2234-2270,
 * usher_graph.hpp:33-53).  Rows of one sample must be sorted by position with
 * no duplicates (UGP_ERR_UNSUPPORTED otherwise).
 */
typedef struct ugp_queries {
    uint64_t n_queries;
    const uint64_t *ent_off;     /* [n_queries+1] CSR offsets                                  */
    const int32_t *pos;          /* [n_ent] position                                           */
    const uint8_t *ref;          /* [n_ent] one-hot VCF REF base                               */
    const uint8_t *nuc;          /* [n_ent] allele mask of the sample at pos (15 when missing) */
    const uint8_t *is_missing;   /* [n_ent] 1 for N / '.' cells (mutation.is_missing)          */
} ugp_queries;

/* Per-sample placement summary: the values usher_common.cpp:451-453 prints and
 * uses -- *best_set_difference, *num_best, *best_j, *has_unique of
 * mapper2_input (usher_graph.hpp:79-92). */
typedef struct ugp_result {
    int32_t best_set_difference;   /* parsimony score of the best placement            */
    uint32_t num_best;             /* number of equally parsimonious placements        */
    uint32_t best_j;               /* BFS index of the chosen node                     */
    uint32_t best_has_unique;      /* 1: place as sibling even if the node is internal */
} ugp_result;

typedef struct ugp_info {
    uint64_t n_nodes;
    uint64_t n_muts;            /* non-masked mutations                                        */
    uint64_t n_sites;           /* distinct mutated positions (rows of an allele tile)         */
    uint64_t stream_bytes;      /* device bytes of the DFS record stream read per tree pass    */
    uint64_t algo_tree_bytes;   /* 4*M + 8*N                                      */
    uint64_t algo_tile_bytes;   /* per sample, L/2 + 16                           */
    uint32_t n_chunks;
    uint32_t max_slots;         /* depth of the per-lane D stack the tree needs                */
    uint32_t max_position;
    uint32_t device;
} ugp_info;

typedef struct ugp_timing {
    float table_ms;      /* allele-tile build kernels of the last call          */
    float place_ms;      /* the dominant kernel of the last call (k_best8, or k_place on the 32-bit path) */
    float merge_ms;      /* reduction after it (phase 2: k_gbest/k_select/k_ties/k_final, or k_merge)     */
    uint32_t place_launches;
    uint32_t n_tiles;    /* sample tiles (one tree pass each) in the last call: 512-sample tiles on
                            the packed path, 64-sample tiles on the 32-bit path */
    uint32_t n_groups;   /* waves per tile in the last call                     */
    uint32_t packed_path; /* 1: 8-samples-per-lane 16-bit kernel + phase 2; 0: 32-bit kernel */
    uint32_t reserved;
    uint64_t words_total;   /* packed path: stream words x tiles the dominant kernel had to cover      */
    uint64_t words_skipped; /* ... of which exact lower-bound pruning skipped (0 with UGP_NO_PRUNE)    */
    float coarse_ms;     /* locality pre-pass of the last call (placement on the coarse top-of-tree MAT), 0 when not run */
    uint32_t bound3;     /* 1: the last call's main walk used the third pruning bound (its per-batch tables were built); ugp_get_timing_sum: how many calls did */
} ugp_timing;

/* Optional: bring up the HIP runtime and the context of `device` now (what the first ugp_mat_create on that device would pay: a few
 * tenths of a second) -- e.g. on a thread of its own while the caller is still reading its inputs.  Thread-safe; calling it again, or
 * never, is harmless. */
int ugp_device_warmup(int device);

/* Flatten + upload.  device = HIP device ordinal.  Replaces the per-sample
 * BFS rebuild and 2N vector allocations of usher_common.cpp:342-365. */
int ugp_mat_create(const ugp_tree_desc *tree, int device, ugp_mat **out);
/* Same tree on several devices of one node (flattened once, uploaded n times): the replicated read-only MAT of
 * the multi-GPU path -- query samples shard across the handles, one host thread per handle. */
int ugp_mat_create_multi(const ugp_tree_desc *tree, const int *devices, int n_devices, ugp_mat **out /* [n_devices] */);
/* One process per GPU (the ranks of a torch.distributed / MPI launch): the flattening is host work with the same result on every
 * rank.  ugp_flat_save runs it once and writes the result to `path` (put it on /dev/shm: a memory copy of ~0.6 GB at 10M nodes);
 * ugp_mat_create_from_flat uploads that file to `device` without flattening again.  The file is only valid for the library build
 * and the flattening switches (UGP_CHUNK_NODES ...) it was written under -- checked, UGP_ERR_UNSUPPORTED otherwise. */
int ugp_flat_save(const ugp_tree_desc *tree, const char *path);
int ugp_mat_create_from_flat(const char *path, int device, ugp_mat **out);
void ugp_mat_destroy(ugp_mat *mat);
int ugp_mat_info(const ugp_mat *mat, ugp_info *out);

/*
 * Score every node of the tree for every query sample and reduce to the best
 * placement: passes 1 and 2 of usher_common.cpp:389-449 for each sample (static
 * tree: the -n / -p semantics).  Host buffers in, host buffer out, synchronous.
 */
int ugp_place_batch(ugp_mat *mat, const ugp_queries *q, ugp_result *out /* [n_queries] */);

/*
 * -p / --write-parsimony-scores-per-node (usher_common.cpp:406-412, 557-578):
 * out[q * n_nodes + j] = set_difference of placing sample q at BFS node j, +1
 * when the node is not an eligible placement (usher_mapper.cpp:498-502).
 */
int ugp_scores_per_node(ugp_mat *mat, const ugp_queries *q, int32_t *out /* [n_queries * n_nodes] */);

/*
 * best_j_vec / node_has_unique (usher_graph.hpp:89-90): all equally
 * parsimonious nodes of each sample, ascending BFS index (the order
 * usher_common.cpp:588 sorts them into), at most `cap` per sample;
 * tie_count[q] receives the true count.
 */
int ugp_tied_nodes(ugp_mat *mat, const ugp_queries *q, uint32_t cap,
                   uint32_t *tie_j /* [n_queries * cap] */,
                   uint8_t *tie_has_unique /* [n_queries * cap] */,
                   uint32_t *tie_count /* [n_queries] */);

/* ---- the other callers of mapper2_body ------------------------------------------------------------------
 * matUtils uncertainty (uncertainty.cpp:212-235: every node but the sample's own, depth-first indices), annotate
 * (annotate.cpp:615-638: depth-first indices), merge (merge.cpp:253-280: the breadth-first expansion of a subtree, cut
 * max_levels below its root) and ripples (ripples/main.cpp:343-377: nodes with enough descendants, a per-node
 * distance, per-node scores) run the same search over THEIR node vector: the index j they hand to mapper2_body --
 * which also breaks ties, usher_mapper.cpp:483-486 -- is a position in that vector.  These entry points take the
 * vector as an order (breadth-first or the reference's depth-first expansion, mutation_annotated_tree.cpp:1253-1273)
 * plus a mask; every node index going in or out is a position in the chosen order.  (A breadth-first expansion of a
 * subtree lists its nodes in the same relative order as the whole tree's, so merge's sub-BFS is order BFS + mask.)
 * Exact for any combination of options.  An order, a distance, a mask and an excluded node per sample run on the packed,
 * pruned path of ugp_place_batch (the order / distance is the tie rank of its second phase, the mask a temporary "no
 * candidate" bit in the tree on the device, the excluded node is taken out of the one chunk minimum it may have set);
 * the score matrix of a breadth-first search comes from the level-by-level kernel of ugp_scores_per_node; the score matrix in
 * depth-first indices and a mask that drops the root take the general one-sample-per-lane kernel (no pruning). */
#define UGP_ORDER_BFS 0u
#define UGP_ORDER_DFS 1u
typedef struct ugp_place_opts {
    uint32_t order;              /* UGP_ORDER_BFS / UGP_ORDER_DFS: meaning of every node index of this call             */
    const uint8_t *node_mask;    /* [n_nodes] or NULL: 0 = the node is not scored (shared by all samples of the call)   */
    const uint32_t *skip_node;   /* [n_queries] or NULL: one node left out for that sample, UINT32_MAX = none           */
    const uint32_t *distance;    /* [n_nodes] or NULL: mapper2_input::distance; among equal scores the smaller wins     */
    int32_t *scores;             /* [n_queries * n_nodes] or NULL: per-node scores as compute_parsimony_scores = true
                                    reports them (+1 when not eligible); nodes that were not scored read 0             */
} ugp_place_opts;
/* If no admitted node is eligible the result is {INT32_MAX, 0, UINT32_MAX, 0} (the reference's callers would be left
 * with their initial values). */
int ugp_place_batch_ex(ugp_mat *mat, const ugp_queries *q, const ugp_place_opts *opts, ugp_result *out /* [n_queries] */);
int ugp_tied_nodes_ex(ugp_mat *mat, const ugp_queries *q, const ugp_place_opts *opts, uint32_t cap, uint32_t *tie_j,
                      uint8_t *tie_has_unique, uint32_t *tie_count);
/* The node-level options prepared once.  ripples (ripples/main.cpp:303-377) runs thousands of searches with ONE node vector -- the
 * nodes with enough descendant leaves -- and ONE distance array; remapping a 10M-entry mask, ranking 10M (distance, leaves) keys and
 * uploading both is then the call (40 of 43 ms per 4,096 samples at 10M nodes).  ugp_ex_prepare does that part once: `opts`
 * contributes order, node_mask and distance (both copied: the arrays may be freed), its skip_node / scores are ignored.  The handle
 * belongs to `mat` and must be destroyed before it.
 * ugp_place_batch_prepared = ugp_place_batch_ex with those options plus, per call, skip_node ([n_queries] positions in the prepared
 * order, or NULL) and d_scores: a DEVICE buffer of n_queries * n_nodes int32 (or NULL) that receives the score matrix as
 * ugp_place_opts::scores describes it -- written by the level-by-level kernel of ugp_scores_per_node for a breadth-first order (no
 * host copy: a caller that reduces the rows on the device, as ripples' per-node filter could, never moves the 40 MB per sample),
 * valid when the call returns. */
typedef struct ugp_ex ugp_ex;
int ugp_ex_prepare(ugp_mat *mat, const ugp_place_opts *opts, ugp_ex **out);
void ugp_ex_destroy(ugp_ex *ex);
int ugp_place_batch_prepared(ugp_mat *mat, const ugp_queries *q, const ugp_ex *ex, const uint32_t *skip_node, ugp_result *out /* [n_queries], host */,
                             int32_t *d_scores /* device, or NULL */);
/* bfs_of[k] = breadth-first index of the node at position k of `order` (how a caller maps its own node vector). */
int ugp_node_order(ugp_mat *mat, uint32_t order, uint32_t *bfs_of /* [n_nodes] */);
/* mask_out[k] = 1 for the nodes of the subtree of root_j that lie at most max_levels below it (merge.cpp:253-256). */
int ugp_subtree_mask(ugp_mat *mat, uint32_t order, uint32_t root_j, uint32_t max_levels, uint8_t *mask_out /* [n_nodes] */);

/* Device-resident variant (no PCIe in the timed path): upload once, place many
 * times.  `stream` is a hipStream_t (NULL = the default stream); d_out is a
 * device pointer to n_queries ugp_result records.  ugp_place_device is asynchronous
 * and STREAM-ORDERED on `stream`, like a kernel launch: it runs behind everything
 * queued on `stream` before it and in front of everything queued after it. */
int ugp_qset_upload(ugp_mat *mat, const ugp_queries *q, ugp_qset **out);
/* (A query set belongs to the handle it was uploaded for.  Sets with many rows per sample -- from 128 on average: the runs of N
 * of low-coverage samples -- also get one bit per (sample, tree site) on the device, about 3 KB per sample at 25,000 sites for
 * the tree and again for its coarse copy, from which every placement call builds its allele tiles.) */
void ugp_qset_destroy(ugp_qset *qs);
uint64_t ugp_qset_size(const ugp_qset *qs);
int ugp_place_device(ugp_mat *mat, ugp_qset *qs, void *d_out, void *stream);
/* Opt-in: consecutive ugp_place_device_overlapped calls on one handle overlap on the device -- up to d = ugp_pipeline_depth(mat)
 * of them at a time (internal streams, workspace sets; d = 3 by default: the small latency-bound kernels around the tree walk of one
 * batch run in the gaps of the others', and each walk takes its share of the resident wave slots; +15 % placements/s at 16,384
 * samples per call over two at a time, +65 % over one).  The price is a weaker ordering than a kernel launch has:
 *   - `stream` receives every call's completion, in call order: work queued on `stream` after call k sees its results;
 *   - call k runs behind the work that was on `stream` when call k - (d - 1) was made -- d - 1 CALLS OF LAG; work queued since
 *     sits behind the completion of a call that is still running, and waiting for it would serialise the calls.  (When no
 *     overlapped call is running any more, call k waits for everything on `stream`.)
 * Hence: cycle through d output buffers.  Whatever is queued on `stream` to read buffer A between call k (which wrote A) and call
 * k + 1 is finished before call k + d overwrites A.  Do not prepare d_out or the query set on `stream` right before the call and
 * expect the call to wait for it -- use ugp_place_device for that. */
int ugp_place_device_overlapped(ugp_mat *mat, ugp_qset *qs, void *d_out, void *stream);
/* How many overlapped calls the handle keeps on the device at a time (its workspace sets: 3 by default, 2..4 with
 * UGP_PIPELINE_DEPTH in the environment when the handle is made) = the number of output buffers to cycle through. */
int ugp_pipeline_depth(const ugp_mat *mat);

/* ugp_place_batch with several batches in flight: host buffers in, host buffers out, asynchronous.  The rows are copied out of
 * `q` before the call returns (pinned staging), `out` is written by ugp_job_wait.  At most ugp_pipeline_depth(mat) jobs per
 * handle may be outstanding (3 by default; batches of more than 32,768 samples or of 128 rows per sample and more: 2 -- one
 * call more fails until the oldest has been waited for); wait for them in the order they were started.
 * ugp_job_wait returns the call's status -- the row checks of ugp_place_batch surface here -- and frees the job.
 *   ugp_job *a, *b; ugp_place_batch_async(mat, &q0, r0, &a);
 *   for (i = 1; i < n; i++) { ugp_place_batch_async(mat, &q[i], r[i], &b); ugp_job_wait(a); a = b; }   ugp_job_wait(a);
 * keeps the device busy with the kernels of one batch while the next one's rows are on their way (two in flight; a ring of
 * ugp_pipeline_depth jobs keeps the device as busy as ugp_place_device_overlapped does: 11.8 M placements/s from host buffers to
 * host buffers at 16,384 samples per job on a 10M-node tree, against 12.5 M/s for inputs resident on the device). */
typedef struct ugp_job ugp_job;
int ugp_place_batch_async(ugp_mat *mat, const ugp_queries *q, ugp_result *out /* [n_queries], valid until ugp_job_wait */, ugp_job **job);
int ugp_job_wait(ugp_job *job);

/* ---- add mode: the tree changes between samples ------------------------------------------------------------------------
 * Default `usher` inserts every sample before it searches for the next (usher_common.cpp:310, the tree edits :652-765), and
 * the reference therefore re-expands and re-searches the whole tree per sample (:342).  An insertion changes very little:
 * it creates a leaf (and, for a sibling placement, an internal node) and rewrites the branch of best_node; the root-path
 * mutation set of every OTHER node -- hence its cost, eligibility and has_unique for any sample (usher_mapper.cpp:167-504) --
 * stays what it was.  This library keeps the flattened tree of ugp_mat_create on the device and takes the edits beside it:
 *
 *   ugp_mat_update   the caller reports the nodes it created or rewrote ("touched") as records: own mutations + the state of
 *                    the node's parent wherever that differs from the reference base.  Records of nodes that exist in the
 *                    flattened tree (flat_j != UINT32_MAX) also take that node OUT of the candidate set of every later
 *                    ugp_place_* / ugp_tied_nodes call on the handle: those calls then return the exact best score, tie count
 *                    and winner over the flattened nodes that are still what they were (winner by the leaf counts and order of
 *                    the flattened tree -- a caller whose tree has grown re-ranks the tie list itself).
 *   ugp_touched_*    score the live records against a batch of pending samples on the device: per sample the minimum cost
 *                    over the eligible records and the records that attain it.  The answer on the tree as it is NOW is the
 *                    merge of the two -- usher_amd/csrc/host/driver.cpp does exactly that, INTEGRATION.md 2a.
 * Records get ids 0, 1, 2 ... in the order they are handed over.  A node that is rewritten again gets a new record; the old
 * one is retired.  Needs a tree of fewer than 2^30 nodes (one bit of the 32-bit record streams marks an excluded node). */
#define UGP_T_LEAF   1u   /* the node is a leaf */
#define UGP_T_MASKED 2u   /* the node carries a masked mutation: `own` lists the mutations in front of it (usher_mapper.cpp:197-200) */
typedef struct ugp_touched {
    uint64_t n;                  /* records in this call                                                                       */
    const uint32_t *flat_j;      /* [n] index of the node in the tree given to ugp_mat_create; UINT32_MAX: created since       */
    const uint8_t *flags;        /* [n] UGP_T_*                                                                                */
    const uint32_t *n_path;      /* [n] how many of the record's entries describe the parent's state (they come first)         */
    const uint64_t *ent_off;     /* [n + 1] CSR into the entry arrays: the parent-state entries, then the own mutations        */
    const int32_t *pos;          /* [n_ent] position                                                                           */
    const uint8_t *allele;       /* [n_ent] parent-state entry: the state (one-hot, != ref); own mutation: the mutated allele  */
    const uint8_t *prev;         /* [n_ent] own mutation: the true parent state at pos (one-hot); parent-state entry: unused   */
    const uint8_t *ref;          /* [n_ent] reference base at pos (one-hot)                                                    */
} ugp_touched;
/* Appends `recs` (may be NULL / n = 0), retires the records listed in `retired` (ids of earlier calls), excludes the flattened
 * nodes named by the new records.  *first_id = id of the first new record. */
int ugp_mat_update(ugp_mat *mat, const ugp_touched *recs, const uint32_t *retired, uint64_t n_retired, uint32_t *first_id);
/* Begin scoring for a batch of pending samples: uploads their rows (checked like ugp_place_batch's), scores every live record. */
int ugp_touched_open(ugp_mat *mat, const ugp_queries *batch);
/* Records with id >= first_id against the samples first_sample .. end of the open batch, merged into the running results
 * (after ugp_mat_update: the samples in front of first_sample have been consumed already). */
int ugp_touched_score(ugp_mat *mat, uint32_t first_id, uint64_t first_sample);
/* One sample again from every live record (its list held retired records only). */
int ugp_touched_rescore(ugp_mat *mat, uint64_t sample);
/* Results of samples [first_sample, first_sample + n): best[i] = minimum cost over the eligible live records (INT32_MAX: none),
 * count[i] = how many attain it (true count), ids / has_unique [i * cap ..] = the first min(count, cap) of them.  A record retired
 * after it entered a list is still listed: the caller knows which ids it retired. */
int ugp_touched_fetch(ugp_mat *mat, uint64_t first_sample, uint64_t n, uint32_t cap, int32_t *best, uint32_t *count, uint32_t *ids,
                      uint8_t *has_unique);

/* Per-kernel durations of the last ugp_place_* call on this handle, measured
 * with HIP events on the stream the kernels ran on (synchronises that stream). */
int ugp_get_timing(ugp_mat *mat, ugp_timing *out);
/* The same durations summed over every ugp_place_* call on this handle since the previous ugp_get_timing_sum (the other
 * fields are those of the last call); waits for the calls still in flight.  Lets a caller time a pipelined sequence of
 * ugp_place_device calls without synchronising after each of them. */
int ugp_get_timing_sum(ugp_mat *mat, ugp_timing *sum, uint32_t *n_calls);

/* Test hook: the third pruning bound's tables of 512-sample tile `tile` as the handle's most recent placement
 * call built them: cum_over / cum_under per block of 16 packed-stream words.  *n_blocks = entries per tile (0: that call did not
 * use the bound); with over == NULL only the count is returned.  Blocking; the caller checks `tile` against its own batch. */
int ugp_debug_bound3_tables(ugp_mat *mat, uint32_t tile, uint16_t *over, uint16_t *under, uint64_t cap, uint64_t *n_blocks);

/* Message for the last non-zero return on the calling thread. */
const char *ugp_last_error(void);

/* ---- MAT construction: Fitch-Sankoff over a batch of VCF sites ---------------
 * Replaces mapper_body::operator() (src/usher_mapper.cpp:6-161), which the VCF
 * reader runs once per site (src/mutation_annotated_tree.cpp:2108-2179) when a
 * MAT is built from a newick tree (`usher -t`).  All sites are assigned in one
 * call; the caller adds the returned mutations to its nodes (Node::add_mutation).
 * Cells of samples that are not in the tree never reach this call (the reader
 * keeps them as the samples' own mutation lists, usher_mapper.cpp:65-82). */
typedef struct ugp_sites {
    uint64_t n_sites;
    const uint8_t *ref;        /* [n_sites] one-hot reference allele (1,2,4,8) */
    const uint64_t *var_off;   /* [n_sites + 1] CSR into var_* */
    const uint32_t *var_node;  /* breadth-first index of the node the VCF column names (leaf or internal) */
    const uint8_t *var_nuc;    /* its allele mask 1..15 (15 = missing); cells equal to REF are simply absent */
} ugp_sites;
typedef struct ugp_fitch ugp_fitch;   /* result handle (host memory) */

/* `parent` is the breadth-first parent array of ugp_tree_desc (root first, parent[0] = UINT32_MAX).
 * The result lists every (site, node) whose assigned state differs from its parent's
 * (the root's parent state is REF), ordered by site, then node index. */
int ugp_fitch_sankoff(int device, uint64_t n_nodes, const uint32_t *parent, const ugp_sites *sites, ugp_fitch **out);
uint64_t ugp_fitch_count(const ugp_fitch *f);
int ugp_fitch_get(const ugp_fitch *f, uint32_t *site, uint32_t *node, uint8_t *par_nuc /* one-hot */, uint8_t *mut_nuc /* one-hot */);
void ugp_fitch_destroy(ugp_fitch *f);
/* ugp_fitch_sankoff keeps its device buffers (row storage of up to 16 GiB -- half of the free HBM if that is less --, sort and output buffers) in a per-device pool for the next
 * call; ugp_fitch_release(device) frees them -- e.g. when the MAT has been built and the same device goes on to place samples. */
void ugp_fitch_release(int device);

/* ---- test / tuning hooks (not part of the drop-in surface) ---------------- */

/* The tuning switches (UGP_* environment variables) are read ONCE, when a handle is created; no placement
 * call reads the environment.  A tool that sweeps switches on one handle re-reads them with this. */
int ugp_mat_reload_knobs(ugp_mat *mat);
/* 1: built with -DUGP_EXPERIMENTS (libusher_amd_exp.so: statistics build of the walk, UGP_STATS / UGP_TRACE /
 * UGP_SEED_PREV / UGP_SEED_CHECK / UGP_PHASE2_PACKED / UGP_KBEST_EXCLUSIVE); 0: the release library ignores those. */
int ugp_has_experiments(void);

/* ugp_mat_create with an explicit chunk size (nodes per DFS chunk), so small
 * fixtures exercise multi-chunk launches. */
int ugp_mat_create_chunked(const ugp_tree_desc *tree, int device, uint32_t chunk_nodes, ugp_mat **out);

/* Host-only view of the flattened tree (works without a GPU): the DFS record
 * stream and tables that ugp_mat_create uploads. */
typedef struct ugp_flat ugp_flat;
enum {
    UGP_FLAT_STREAM = 0,        /* uint32 */
    UGP_FLAT_PRE_STREAM = 1,    /* uint32 */
    UGP_FLAT_CHUNK_BODY_OFF = 2,/* uint32 [n_chunks+1] */
    UGP_FLAT_CHUNK_PRE_OFF = 3, /* uint32 [n_chunks+1] */
    UGP_FLAT_CHUNK_NODE_OFF = 4,/* uint32 [n_chunks+1] */
    UGP_FLAT_POS2SITE = 5,      /* int32  [max_pos+1] */
    UGP_FLAT_SITE_REF = 6,      /* uint8  [n_sites] */
    UGP_FLAT_RANK2BFS = 7,      /* uint32 [n_nodes] */
    UGP_FLAT_DFS2BFS = 8,       /* uint32 [n_nodes] */
    UGP_FLAT_MAX_SLOTS = 9,     /* count only */
    UGP_FLAT_STREAM8 = 10,      /* uint32: packed stream walked by the 8-samples-per-lane kernel */
    UGP_FLAT_PRE8_STREAM = 11,  /* uint32 */
    UGP_FLAT_CHUNK8_BODY_OFF = 12, /* uint32 [n_chunks+1] */
    UGP_FLAT_CHUNK8_PRE_OFF = 13,  /* uint32 [n_chunks+1] */
    UGP_FLAT_MAX_PATH_MUTS = 14, /* count only */
    UGP_FLAT_STREAM_T = 15,     /* uint32: tie stream walked by phase 2 (chunk bodies + pruning pseudo-records) */
    UGP_FLAT_CHUNK_T_OFF = 16,  /* uint32 [n_chunks+1] */
    UGP_FLAT_LDS_SLOTS = 17,    /* count only: saved-D slots the packed stream keeps on the fast path */
    UGP_FLAT_B3_GROUP_OFF = 18, /* uint32 [4][groups + 1]: third pruning bound, the four event lists of every group of 256 blocks of 16 packed-stream words */
    UGP_FLAT_B3_EVENTS = 19     /* uint32 [events]: 4 * site + allele (bits 23:0) | block within the group (31:24) */
};
int ugp_flat_create(const ugp_tree_desc *tree, uint32_t chunk_nodes, ugp_flat **out);
void ugp_flat_destroy(ugp_flat *flat);
int ugp_flat_get(const ugp_flat *flat, int which, const void **ptr, uint64_t *count);

#ifdef __cplusplus
}
#endif
#endif /* USHER_AMD_H */
