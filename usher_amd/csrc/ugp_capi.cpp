// ugp_capi.cpp -- the extern "C" boundary of libusher_amd.so (include/usher_amd.h).
// Owns device memory behind opaque handles; never throws across the ABI and
// never exits the process.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <climits>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <mutex>
#include <vector>

#include <unistd.h>

#include "ugp_bound3.hpp"
#include "ugp_tuner.hpp"
#include "ugp_flatten.hpp"
#include "ugp_kernels.hpp"
#include "ugp_knobs.hpp"
#include "ugp_update.hpp"
#include "usher_amd.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}

#define HIP_TRY(expr)                                                                         \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(UGP_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

}  // namespace

namespace ugp {
int set_error(int code, const std::string &msg) { return fail(code, msg); }   // for the other translation units
}

namespace {

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;   // elements
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
    }
    hipError_t reserve(size_t n) {
        if (n <= cap) return hipSuccess;
        release();
        hipError_t e = hipMalloc((void **)&p, std::max<size_t>(n, 1) * sizeof(T));
        if (e == hipSuccess) cap = n; else p = nullptr;
        return e;
    }
    // grow to at least n elements, keeping the first `used` (doubling: appended to many times)
    hipError_t grow_keep(size_t n, size_t used) {
        if (n <= cap) return hipSuccess;
        const size_t want = std::max<size_t>(n, cap * 2);
        T *np = nullptr;
        hipError_t e = hipMalloc((void **)&np, std::max<size_t>(want, 1) * sizeof(T));
        if (e != hipSuccess) return e;
        if (p && used) e = hipMemcpy(np, p, used * sizeof(T), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) { (void)hipFree(np); return e; }
        if (p) (void)hipFree(p);
        p = np; cap = want;
        return hipSuccess;
    }
    template <class V>
    hipError_t upload(const V &v) {
        hipError_t e = reserve(v.size());
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
};

// Pinned host memory (grown on demand): staging for copies that have to be asynchronous.
struct PinBuf {
    void *p = nullptr;
    size_t cap = 0;
    ~PinBuf() { if (p) (void)hipHostFree(p); }
    // device_writes: a kernel stores into the buffer (results of ugp_place_batch_async) -- mapped and coherent, so that what it wrote
    // is in host memory when its stream's event has been waited for, whatever the platform's default for pinned memory is
    int reserve(size_t n, bool device_writes = false) {
        if (n <= cap) return UGP_OK;
        if (p) (void)hipHostFree(p);
        p = nullptr; cap = 0;
        if (hipHostMalloc(&p, std::max<size_t>(n, 1), device_writes ? (hipHostMallocMapped | hipHostMallocCoherent) : hipHostMallocDefault) != hipSuccess) { p = nullptr; return fail(UGP_ERR_HIP, "hipHostMalloc (staging buffer)"); }
        cap = n;
        return UGP_OK;
    }
};

// A large device -> pageable host copy (the -p score matrix: 5 GB per 128 samples at 10M nodes) through two pinned staging buffers:
// the DMA of piece i runs while a few host threads move piece i - 1 to its place.  A plain hipMemcpy into pageable memory reaches
// ~21 GB/s on this platform, the link more than twice that.
int copy_d2h_staged(void *dst, const void *src, size_t bytes) {
    constexpr size_t kPiece = 32u << 20;
    if (bytes < 4 * kPiece) {
        if (bytes && hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost) != hipSuccess) return fail(UGP_ERR_HIP, "copy to the host");
        return UGP_OK;
    }
    PinBuf pin[2];
    hipStream_t st = nullptr;
    hipEvent_t ev[2] = {nullptr, nullptr};
    int rc = UGP_OK;
    if (pin[0].reserve(kPiece) != UGP_OK || pin[1].reserve(kPiece) != UGP_OK) return UGP_ERR_HIP;
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&ev[0], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&ev[1], hipEventDisableTiming) != hipSuccess) rc = fail(UGP_ERR_HIP, "staged copy: stream / events");
    const size_t n_pieces = (bytes + kPiece - 1) / kPiece;
    const unsigned n_thr = std::max(1u, std::min(8u, std::thread::hardware_concurrency() / 2));
    auto drain = [&](size_t i) {   // piece i: pinned -> dst, on n_thr threads
        const size_t off = i * kPiece, n = std::min(kPiece, bytes - off);
        const char *from = (const char *)pin[i & 1].p;
        char *to = (char *)dst + off;
        std::vector<std::thread> th;
        const size_t per = ((n + n_thr - 1) / n_thr + 4095) & ~(size_t)4095;
        for (unsigned t = 1; t < n_thr; t++)
            if ((size_t)t * per < n) th.emplace_back([=]() { memcpy(to + (size_t)t * per, from + (size_t)t * per, std::min(per, n - (size_t)t * per)); });
        memcpy(to, from, std::min(per, n));
        for (auto &x : th) x.join();
    };
    for (size_t i = 0; i <= n_pieces && rc == UGP_OK; i++) {
        if (i < n_pieces) {   // (buffer i & 1 was drained in iteration i - 1, as piece i - 2)
            const size_t off = i * kPiece, n = std::min(kPiece, bytes - off);
            if (hipMemcpyAsync(pin[i & 1].p, (const char *)src + off, n, hipMemcpyDeviceToHost, st) != hipSuccess || hipEventRecord(ev[i & 1], st) != hipSuccess)
                rc = fail(UGP_ERR_HIP, "staged copy: hipMemcpyAsync");
        }
        if (i > 0 && rc == UGP_OK) {
            if (hipEventSynchronize(ev[(i - 1) & 1]) != hipSuccess) rc = fail(UGP_ERR_HIP, "staged copy: wait");
            else drain(i - 1);
        }
    }
    if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    for (auto e : ev) if (e) (void)hipEventDestroy(e);
    return rc;
}

struct EventSet {
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool used = false;
    // third pruning bound: was it a choice for this sub-batch (-1: no), which way it went, its place in the tuner's sequence
    int b3_class = -1;
    bool b3_used = false;
    uint32_t b3_tiles = 0, b3_pos = 0;
    bool b3_first = false;   // (of the class's very first block: not counted)
    uint64_t b3_seq = 0;
};

using ugp::B3Tuner;   // (ugp_tuner.hpp: the third pruning bound, with or without, by measurement)

constexpr uint32_t kMaxTilesPerLaunch = 4096;   // 262,144 samples per sub-batch
constexpr int kMaxSets = 4;                     // workspace sets per handle = calls of ugp_place_device_overlapped that can be on the device at a time

}  // namespace

struct ugp_mat {
    int device = 0;
    ugp::Knobs knobs;    // tuning switches, read from the environment when the handle is made (ugp_knobs.hpp); no getenv in a call
    ugp::FlatMat flat;   // the scalars of the flattening only (counts, depths); the arrays live on the device
    uint64_t stream_dwords = 0, pre_dwords = 0;
    DevBuf<uint32_t> d_stream, d_pre, d_chunk_body, d_chunk_pre, d_chunk_node, d_rank2bfs, d_dfs2bfs;
    DevBuf<uint32_t> d_stream8, d_pre8, d_chunk8_body, d_chunk8_pre, d_stream_t, d_chunk_t;
    DevBuf<uint32_t> d_node_pos8, d_rank_dfs;   // packed-stream position of every node's words and its tie rank, by DFS index (k_best8 names nodes by position)
    DevBuf<uint32_t> d_b3_group_off, d_b3_events;   // third pruning bound: the mutation events in block order (FlatMat::b3_*)
    uint64_t stream8_dwords = 0;
    DevBuf<int32_t> d_pos2site, d_site_pos;   // position -> site, site -> position
    DevBuf<uint8_t> d_site_ref;
    // locality sort (speed only): a coarse MAT of the top of the tree and the map coarse BFS index -> DFS rank in the full tree
    ugp_mat *coarse = nullptr;
    DevBuf<uint32_t> d_coarse2bfs, d_node_pair, d_parent;   // seed descent (k_descend)
    bool wide_descent = false;
    DevBuf<uint32_t> d_coarse2dfs, d_coarse_bin;   // coarse node -> depth-first rank in the full tree / its position among the coarse nodes in that order
    uint32_t n_coarse_bins = 0;
    std::vector<uint32_t> h_level_off;   // breadth-first level boundaries (empty: the input is not a breadth-first expansion)
    uint32_t max_level_width = 0;
    // extended searches (ugp_place_batch_ex): host copy of the topology, the reference's depth-first order and its tie rank
    std::vector<uint32_t> h_parent, h_dfs2bfs, h_bfs2dfs, h_leaves;
    std::vector<uint32_t> h_flat_chunk_node;   // chunk cuts by the flattener's depth-first index (host copy, with h_flat_bfs2dfs)
    std::vector<uint32_t> h_flat_bfs2dfs;   // inverse of flat.dfs2bfs (the flattener's own depth-first order), built by the first search that leaves nodes out per sample
    DevBuf<uint32_t> d_dfs2bfs_caller;   // the reference's depth-first expansion on the device (position -> BFS index)
    DevBuf<uint32_t> d_dfs_rank, d_dfs_rank2out, d_bfs2dfs;
    bool dfs_rank_ready = false;
    ugp_qset *own_qs = nullptr;          // reusable query set / result buffer of the host-buffer entry points
    DevBuf<ugp_result> d_own_out;
    size_t occ_lds = ~(size_t)0;   // k_best8 occupancy cache: LDS bytes it was queried for
    int occ_per_cu = 0, n_cu = 0, occ_variant = -1;
    // Everything one call writes on the device, twice: consecutive ugp_place_device calls alternate between the two sets and
    // run on two streams of the handle's own, so the small latency-bound kernels around k_best8 (row checks, coarse pass,
    // sort, tile build, seed descent, phase 2) of one batch fill the idle issue slots of the other's.  The other entry
    // points use set 0 on the caller's stream.
    struct Work {
        DevBuf<uint32_t> d_table, d_zero, d_part_best, d_part_cnt, d_part_key;
        DevBuf<uint32_t> d_lpos;    // (coarse pass) which node set each chunk minimum
        DevBuf<uint32_t> d_tie_units, d_tie_info;   // phase 2 as a mode of the packed walk: its unit lists (room for every (tile, chunk)) and counters
        DevBuf<uint32_t> d_lbest, d_gbest, d_gbest_part, d_items, d_ub, d_gstart, d_hlen, d_cold, d_list, d_units, d_unit_info;
        DevBuf<uint64_t> d_dyn;
        uint32_t dyn_epoch = 0;
        // third pruning bound: per-batch block tables of every tile (ugp_bound3.hpp)
        DevBuf<uint32_t> d_b3_pairmask;
        DevBuf<uint8_t> d_b3_over, d_b3_under, d_b3_l1, d_b3_l2, d_b3_l3;
        DevBuf<ugp::B3Dev> d_b3_dev;
        ugp::B3Dev b3_host = {};
        DevBuf<uint64_t> d_stats, d_trace;
        uint64_t last_words_total = 0;
        const uint32_t *last_nitems = nullptr;   // (statistics) phase 2's item counter of the last launch sequence
        const uint32_t *last_list_n = nullptr;   // (UGP_STATS) record counts of the last packed launch, per 512-sample tile
        uint32_t last_list_tiles = 0;
        DevBuf<uint32_t> d_dnode, d_dres, d_luniq, d_gcnt, d_gcnt_part;   // phase 2 without a walk for single-node minima (Phase2Uniq)
        DevBuf<uint32_t> d_refined, d_keys, d_keys2, d_idx, d_order, d_slot, d_bins;
        DevBuf<ugp_result> d_coarse_res, d_prev_res;
        uint64_t prev_serial = 0;   // (UGP_SEED_PREV / UGP_SEED_CHECK diagnostics) content serial of the query set d_prev_res belongs to; 0 = none
        DevBuf<uint8_t> d_sort_tmp;
        bool last_used_best8 = false;
        // HIP events and durations of the last few calls that used this set, a small ring: a call takes the oldest entry, whose
        // events completed long ago, so recording never has to wait for a call still in flight
        struct Gen {
            std::vector<EventSet> events;
            hipEvent_t ev_coarse[2] = {nullptr, nullptr};
            bool coarse_timed = false;
            size_t events_used = 0;
            ugp_timing last = {};
            bool timing_pending = false;
        } gens[4];
        uint32_t cur = 0;   // ring entry of the set's latest call
        // ugp_place_batch_async: the set's reusable query set, result buffer and pinned staging (rows in, results out)
        ugp_qset *job_qs = nullptr;
        DevBuf<ugp_result> d_job_out;
        PinBuf job_in, job_out;
        bool job_busy = false;           // a job on this set has been started and not yet waited for
        // (round 6) a side stream of the set: work of one call that does not depend on each other runs beside its main chain -- the fill of
        // the main pass's allele tiles under the locality pre-pass, the third bound's tables beside the seed descent (fork / join events)
        hipStream_t aux = nullptr;
        hipEvent_t ev_fork = nullptr, ev_fill = nullptr, ev_join = nullptr;
        bool join_pending = false;       // work on `aux` that the call's stream has not waited for yet
        hipStream_t stream = nullptr;    // the handle's own stream for this set (ugp_place_device)
        hipEvent_t done = nullptr;       // recorded behind the last call that used this set
        hipStream_t done_on = nullptr;   // ... on this stream
    } work[kMaxSets];
    ugp_timing tsum = {};    // durations of all calls since the last ugp_get_timing_sum
    uint32_t tsum_calls = 0;
    int last_work = 0;       // set used by the most recent call (ugp_get_timing reports it)
    B3Tuner b3_tuner;        // third pruning bound: with or without, by measurement
    int next_work = 0;       // set the next ugp_place_device_overlapped call takes (cycles through knobs.depth sets)
    int next_job = 0;        // set the next ugp_place_batch_async job takes (sets 0 .. pipeline depth - 1)
    uint64_t n_overlapped = 0;                  // calls of ugp_place_device_overlapped so far
    hipEvent_t entry_ring[kMaxSets] = {};       // the caller's stream at the moment of the last kMaxSets such calls
    int share_sets = 2;      // (during a call) workspace sets the caller cycles through (ugp_place_device_overlapped: 2 for long calls, else the pipeline depth)
    int share_n = 1;         // (during a call) tree walks expected on the device at a time: this call's grid is its share of the resident wave slots
    uint32_t tie_lists_filled = 0, tie_sub_batches = 0;   // sub-batches of the current call whose tie lists phase 2 has filled / all of them (ugp_tied_nodes)
    bool sharing = false;    // (during a call) another set's call was still running when this one, or the one before it, was queued
    bool was_busy = false;   // ... when the previous call was queued
    int was_n = 1;           // share_n of the previous call
    // ---- add mode (ugp_mat_update / ugp_touched_*): where each node's words sit in the record streams (by BFS index; the coarse
    // tree's by coarse index), the records of the nodes created or rewritten since, and the open batch's scoring state
    struct Upd {
        std::vector<uint32_t> hdr8, rec, post, coarse2bfs, excluded_j;
        DevBuf<uint32_t> d_hdr8, d_rec_pos, d_post;   // the same maps on the device (node masks of the extended searches), uploaded on first use
        DevBuf<ugp::TouchedRec> d_rec;
        DevBuf<ugp::TouchedEnt> d_ent;
        DevBuf<uint8_t> d_alive;
        DevBuf<uint32_t> d_tmp;
        uint64_t n_rec = 0, n_ent = 0, n_excluded = 0;
        ugp_qset *qs = nullptr;
        DevBuf<uint8_t> d_dense, d_hu;
        DevBuf<int32_t> d_dbot, d_best, d_list_best;
        DevBuf<uint32_t> d_cnt, d_ids;
        uint32_t n_pos = 0, qpad = 0;
        uint64_t Q = 0;
        bool open = false;
        PinBuf stage_up;   // ugp_mat_update: records, entries and word positions in one staging buffer
        PinBuf stage;   // ugp_touched_fetch: the four result arrays cross in one go (asynchronous copies into pinned memory, one wait)
    } upd;
    hipEvent_t kb_done = nullptr;    // behind the latest k_best8 launch of this handle ...
    hipStream_t kb_done_on = nullptr;   // ... on this stream
};

struct ugp_qset {
    int device = 0;
    uint64_t n_queries = 0, n_ent = 0;
    uint64_t max_rows = 0;           // largest number of rows of one sample
    uint64_t serial = 0;             // changes whenever the set is (re)filled: results remembered for one content never seed another
    DevBuf<int32_t> d_pos;
    DevBuf<uint8_t> d_ref, d_nuc, d_missing;
    DevBuf<uint32_t> d_ent_q;
    DevBuf<uint64_t> d_ent_off;
    DevBuf<unsigned long long> d_err;
    // Many missing rows per sample (N runs): one bit per (sample, site) per tree the set is placed on -- the MAT and its coarse
    // MAT number their sites differently -- and the list of the rows that are NOT missing (k_nmask_build, k_ntiles)
    DevBuf<uint32_t> d_nmask[2], d_plain_rows, d_n_plain;
    const ugp_mat *nmask_for[2] = {nullptr, nullptr};
    uint32_t nmask_words[2] = {0, 0};
    std::vector<uint64_t> ent_off;   // host copy, for sub-batching
};

namespace ugp { void fitch_drop_streams(int device); }   // ugp_fitch.hip
namespace {

uint32_t pick_groups(const ugp_mat *m, uint32_t n_tiles, uint32_t target_waves = 4096) {
    if (m->knobs.target_waves) target_waves = m->knobs.target_waves;
    uint32_t g = (target_waves + n_tiles - 1) / n_tiles;
    if (m->knobs.groups) g = m->knobs.groups;
    g = std::min<uint32_t>(g, m->flat.n_chunks);
    g = std::max<uint32_t>(g, 1);
    if (g >= 8) g &= ~7u;   // XCD-aware block mapping wants a multiple of 8
    return g;
}

// Host part of the query checks: the CSR offsets.  The per-row checks run on the device (k_rows_prepare).
int validate_offsets(const ugp_queries *q, uint64_t &n_ent, uint64_t &max_rows) {
    if (!q || (q->n_queries && !q->ent_off)) return fail(UGP_ERR_INVALID, "null query arrays");
    if (q->n_queries >= (1ull << 31)) return fail(UGP_ERR_UNSUPPORTED, "more than 2^31 queries in one batch");
    n_ent = q->n_queries ? q->ent_off[q->n_queries] : 0;
    if (n_ent && (!q->pos || !q->ref || !q->nuc || !q->is_missing)) return fail(UGP_ERR_INVALID, "null query entry arrays");
    max_rows = 0;
    for (uint64_t s = 0; s < q->n_queries; s++) {
        const uint64_t b = q->ent_off[s], e = q->ent_off[s + 1];
        if (e < b || e > n_ent) return fail(UGP_ERR_INVALID, "ent_off is not monotone");
        max_rows = std::max(max_rows, e - b);
    }
    if (q->n_queries && q->ent_off[0] != 0) return fail(UGP_ERR_INVALID, "ent_off[0] must be 0");
    return UGP_OK;
}

}  // namespace
static int harvest_timing(ugp_mat *m, ugp_mat::Work &W, ugp_mat::Work::Gen &G);
static void tuner_poll(ugp_mat *m);
namespace {
int ensure_events(ugp_mat::Work::Gen &G, size_t n) {
    while (G.events.size() < n) {
        EventSet es;
        for (int i = 0; i < 4; i++) HIP_TRY(hipEventCreate(&es.ev[i]));
        G.events.push_back(es);
    }
    return UGP_OK;
}

// Device-side options of an extended search (the other callers of mapper2_body); null members = not used.
struct ExDev {
    const uint8_t *mask = nullptr; const uint32_t *skip = nullptr, *alt_rank = nullptr, *out_index = nullptr, *rank2out = nullptr;
    const uint32_t *skip_chunk = nullptr;   // [n_queries] with skip: the chunk of the flattened tree that holds the sample's excluded node
    int32_t *scores = nullptr;
    bool packed = false;   // the caller has arranged for the packed path (mask turned into exclusions; no scores)
};

// mode 0: results to d_out (device ugp_result[n_queries]);
// mode 1: per-node scores to d_scores (device int32 [n_queries][n_nodes]);
// mode 2: tied nodes (needs d_best_in) -- see ugp_kernels.hip.
int run_place(ugp_mat *m, ugp_qset *qs, int mode, ugp_result *d_out, int32_t *d_scores, const int32_t *d_best_in,
              uint32_t *d_tie_count, uint32_t *d_tie_j, uint8_t *d_tie_hu, uint32_t tie_cap, hipStream_t s, bool coarse_only = false,
              const ExDev *ex = nullptr, int wi = 0) {
    HIP_TRY(hipSetDevice(m->device));
    const auto &f = m->flat;
    const ugp::Knobs &K = m->knobs;
    const uint64_t Q = qs->n_queries;
    ugp_mat::Work &W = m->work[wi];
    m->last_work = wi;
    // the previous call that used this workspace set may have run on another stream
    if (W.done && W.done_on != s) HIP_TRY(hipStreamWaitEvent(s, W.done, 0));
    struct Finish {   // whatever way this call ends: mark the set's last use
        ugp_mat::Work &W; hipStream_t s;
        ~Finish() {
            if (W.join_pending) { (void)hipStreamWaitEvent(s, W.ev_join, 0); W.join_pending = false; }   // (a call that failed between fork and join)
            if (!W.done && hipEventCreateWithFlags(&W.done, hipEventDisableTiming) != hipSuccess) { W.done = nullptr; return; }
            (void)hipEventRecord(W.done, s);
            W.done_on = s;
        }
    } finish{W, s};
    W.cur = (W.cur + 1u) & 3u;
    ugp_mat::Work::Gen &TG = W.gens[W.cur];
    if (int rc = harvest_timing(m, W, TG)) return rc;   // (this ring entry's previous call, four uses of the set ago, before its events are recorded again)
    TG.events_used = 0;
    TG.last = {};
    TG.timing_pending = true;
    if (Q == 0) return UGP_OK;
    // at least one table row, so that words without a row of their own (headers,
    // reference-everywhere sites) always have the valid row 0 to fetch
    const uint32_t n_sites = std::max<uint32_t>((uint32_t)f.n_sites, 1u);
    const uint32_t active_words = (n_sites + 31) / 32;
    // Locality sort: place every sample on the coarse top-of-the-tree MAT first; samples are then
    // assigned to 512-sample tiles in the DFS order of that coarse placement (k_sort_keys).
    // (16-bit phase 1: every D / cost must stay below 0x7F7F, the value the shared upper bounds start from;
    // a tree with a masked mutation behind an ordinary one on the same node -- never produced by the reference's
    // sorted Node::add_mutation, mutation_annotated_tree.cpp:720-752 -- needs the order-aware 32-bit walk)
    // (the extended searches take the packed path too when their options are the kind it can express: a node order / distance is a
    // tie rank of phase 2, a node mask has been turned into exclusions by the caller, a per-sample excluded node is taken out of
    // the one chunk minimum it can have set, behind phase 1 (k_fix_skip); per-node scores stay on the one-sample-per-lane kernel)
    const bool ex_packable = ex && ex->packed && !ex->mask && (!ex->skip || ex->skip_chunk) && !ex->scores;
    const uint32_t *const ex_skip = (ex_packable && ex->skip) ? ex->skip : nullptr;   // (caller order; + q0 per sub-batch)
    const bool packed_ok = (mode == 0) && (!ex || ex_packable) && !K.force_v1 && !f.mask_not_first && (qs->max_rows + f.max_path_muts + 2 < 0x7F7Full);
    const bool sorted = packed_ok && m->coarse && Q > 512 && !K.no_sort && !K.no_prune;
    TG.coarse_timed = false;
    // The side stream (see Work::aux): only where there is something to put on it -- the sorted main pass.
    // And only for a call that has the device to itself: measured with three calls in flight (six queues instead of three), the
    // cross-queue waits cost far more than the overlap gives -- 12.9 -> 10.4 M placements/s; a lone call gains 45 us of its 1.96 ms.
    // OPT-IN (UGP_FORK=1) since the end of round 6: a process has four hardware queues by default, a handle's three streams plus
    // their side streams oversubscribe them, and which streams then share a queue -- and serialise -- depends on what the process
    // created before: the SECOND handle of a process ran its overlapped calls one after the other (tools/probe_context.py: config 3's
    // size 7.7 -> 4.7 M/s, the headline workload 13.9 -> 9.6) for a gain of 45 us on the first.
    const bool can_fork = sorted && !coarse_only && !K.no_fork && !m->sharing;
    bool fill_ahead = false;   // the first sub-batch's table has been filled with the reference bases on the side stream, under the pre-pass
    if (can_fork) {
        if (!W.aux) {
            HIP_TRY(hipStreamCreateWithFlags(&W.aux, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&W.ev_fork, hipEventDisableTiming)); HIP_TRY(hipEventCreateWithFlags(&W.ev_fill, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&W.ev_join, hipEventDisableTiming));
        }
        // plain batches (fill + scatter; the other tile builders write every row themselves): the fill depends on nothing but the set
        // being free, which `s` has just waited for
        int nmi0 = -1;
        for (int i = 0; i < 2; i++) if (qs->nmask_for[i] == m && n_sites) nmi0 = i;
        const uint64_t nq0 = std::min<uint64_t>(Q, (uint64_t)kMaxTilesPerLaunch * 64);
        if (nmi0 < 0 && K.tile_build <= 0 && f.n_sites && nq0 == Q) {
            const uint64_t dw = (uint64_t)((nq0 + 511) / 512) * (n_sites + ugp::TABLE_CONST_ROWS) * 64;
            HIP_TRY(W.d_table.reserve(dw));
            HIP_TRY(hipEventRecord(W.ev_fork, s));
            HIP_TRY(hipStreamWaitEvent(W.aux, W.ev_fork, 0));
            HIP_TRY(ugp::launch_fill_table(W.d_table.p, m->d_site_ref.p, n_sites, dw, W.aux));
            HIP_TRY(hipEventRecord(W.ev_fill, W.aux));
            fill_ahead = true;
        }
    }
    if (sorted) {
        HIP_TRY(W.d_coarse_res.reserve(Q));
        if (!TG.ev_coarse[0]) { HIP_TRY(hipEventCreate(&TG.ev_coarse[0])); HIP_TRY(hipEventCreate(&TG.ev_coarse[1])); }
        HIP_TRY(hipEventRecord(TG.ev_coarse[0], s));
        // The pre-pass has no phase 2: its walk records which node set every chunk minimum (k_best8<ARG>, k_coarse_result) -- any
        // node of minimal cost serves the sort and the descent.  (UGP_COARSE_PHASE2=1: the full phase 2 instead, i.e. the
        // reference's tie-break winner: 0.2 ms more per 16,384 samples, the same answers.)
        m->coarse->sharing = m->sharing; m->coarse->share_n = m->share_n; m->coarse->share_sets = m->share_sets;
        const bool coarse_arg = m->coarse->d_node_pos8.p && m->coarse->flat.max_chunk8_words < 65536u && !K.coarse_phase2;
        if (int rc = run_place(m->coarse, qs, 0, W.d_coarse_res.p, nullptr, nullptr, nullptr, nullptr, nullptr, 0, s, coarse_arg, nullptr, wi)) return rc;
        HIP_TRY(hipSetDevice(m->device));
        HIP_TRY(hipEventRecord(TG.ev_coarse[1], s));
        TG.coarse_timed = true;
    }
    // samples per sub-batch: at most 262,144, and few enough that the per-(chunk, sample) minima of phase 1
    // (2 bytes each) stay below 24 GiB (8 until round 4: 131,072 samples per launch sequence at 10M nodes; the walk does
    // 16,384 samples' worth of work in 0.55 ms at 262,144 per launch against 0.68 at 65,536 -- 1M queries per call 20.0 -> 23.1 M/s)
    uint64_t sub_tiles = kMaxTilesPerLaunch;
    if (m->flat.n_chunks) sub_tiles = std::min<uint64_t>(sub_tiles, std::max<uint64_t>(8, ((K.lbest_gib ? (uint64_t)K.lbest_gib : 24ull) << 30) / ((uint64_t)m->flat.n_chunks * 128) & ~7ull));
    for (uint64_t q0 = 0; q0 < Q; q0 += sub_tiles * 64) {
        const uint64_t nq = std::min<uint64_t>(Q - q0, sub_tiles * 64);
        if (mode == 0 && d_tie_count) m->tie_sub_batches++;
        const uint32_t n_tiles = (uint32_t)((nq + 63) / 64);
        const uint32_t n_tiles512 = (uint32_t)((nq + 511) / 512);
        // 16-bit packed phase 1 is exact while every D / cost stays below 0x8000 (bit 15 is the ineligible flag)
        const bool use8 = packed_ok;
        uint32_t G = pick_groups(m, n_tiles);
        if (use8) {   // work units of a few chunks each, pulled from per-XCD queues by persistent waves (see k_best8)
            uint32_t unit_chunks = 16;   // (most far units end in their preamble: the replay is the cost to amortise)
            if (K.unit_chunks) unit_chunks = K.unit_chunks;
            G = std::max<uint32_t>(1, (f.n_chunks + unit_chunks - 1) / unit_chunks);
            // a small batch still has to fill the chip: at least ~4096 units in total
            while (G < f.n_chunks && (uint64_t)G * n_tiles512 < 4096) G = std::min<uint32_t>(f.n_chunks, G * 2);
            if (K.groups) G = std::min<uint32_t>(f.n_chunks, K.groups);
        }
        const uint64_t table_dwords = (uint64_t)n_tiles512 * (n_sites + ugp::TABLE_CONST_ROWS) * 64;
        HIP_TRY(W.d_table.reserve(table_dwords));
        const uint64_t pairs = (uint64_t)f.n_chunks * n_tiles512 * 8;
        // every small buffer that has to start from zero lives in ONE allocation cleared by one memset:
        // D(bottom) counters, active-row bitmap, work-queue heads, record lists' lengths, phase-2 item count, tie counts / keys
        // Third pruning bound (round 5): for the sorted main walk of batches of up to 256 tiles, when the tree carries its posting lists
        // and the tiles are built by the scatter kernels (they mark the tile's useful (site, allele) pairs)
        const uint32_t useful_words = (n_sites + 7) / 8;
        // (what the walk will be launched with is decided here, in front of the tile build: a batch whose walk keeps its active-row
        // bitmap in LDS -- lds_bits == 1 -- has no third-bound variant, and building the tables for it, or letting the tuner book
        // the batch as one "with", would be cost without effect: ADVICE r5)
        int nmi = -1;   // the query set carries N masks for this tree
        for (int i = 0; i < 2; i++) if (qs->nmask_for[i] == m && use8 && n_sites) nmi = i;
        const uint64_t e0 = qs->ent_off[q0], e1 = qs->ent_off[q0 + nq];
        // many rows per sample (high-ambiguity queries): the (tile, site block)-in-LDS builder; otherwise fill + one atomic per row
        // (only for batches in arrival order, i.e. the coarse pass: consecutive threads then read neighbouring row lists; behind
        // the locality sort the builder's uncoalesced row reads cost more than the scatter's atomics -- measured on config 5:
        // coarse pass 2.72 -> 2.33 ms, sorted build 1.77 -> 2.03 ms)
        bool lds_build = use8 && n_sites && !sorted && (e1 - e0) >= (uint64_t)nq * 128;
        if (K.tile_build >= 0) lds_build = use8 && n_sites && K.tile_build != 0;
        uint32_t lds_bits_plan = 0;
        if (use8) {
            const bool stats_on = K.stats;
            lds_bits_plan = (!stats_on && (size_t)active_words * 4 <= 4096 && n_tiles512 >= 64 && !m->sharing) ? 1u : 0u;
            if (nmi >= 0 && !stats_on) lds_bits_plan = 2u;
            if (K.lds_bits >= 0) lds_bits_plan = stats_on ? 0u : (K.lds_bits == 2 ? 2u : (((size_t)active_words * 4 <= 4096 && K.lds_bits != 0) ? 1u : 0u));
        }
        const bool b3_can = use8 && !coarse_only && sorted && m->d_b3_events.p && !K.no_bound3 && K.bound3 != 0 && !K.no_prune && n_tiles512 <= 256 && K.tile_build <= 0 &&
                            lds_bits_plan != 1u && !(lds_build && nmi < 0);
        bool b3_want = b3_can;
        const int b3_class = B3Tuner::class_of(e1 - e0, nq);
        uint32_t b3_pos = 0;
        uint64_t b3_seq = 0;
        // UGP_BOUND3 unset: decided from what is known of the tree and of the batch (ugp_tuner.hpp b3_static_choice) -- every call of a
        // kind runs the same way from the first one on; UGP_BOUND3=auto: the handle's run-time A/B (B3Tuner); 1 / 0: pinned
        const bool b3_tuned = b3_can && K.bound3 == -1;
        if (b3_tuned) { tuner_poll(m); b3_want = m->b3_tuner.next(b3_class, &b3_pos, &b3_seq); }
        else if (b3_can && K.bound3 < -1) b3_want = ugp::b3_static_choice(m->wide_descent, b3_class, f.n_nodes);
        const size_t z_dbottom = 0, z_active = z_dbottom + (size_t)n_tiles512 * 512, z_queue = z_active + (size_t)n_tiles512 * active_words,
                     z_list_n = z_queue + 8, z_nitems = z_list_n + n_tiles512, z_cnt = z_nitems + 8, z_key = z_cnt + (size_t)n_tiles512 * 512,
                     z_useful = z_key + (size_t)n_tiles512 * 512, z_end = z_useful + (b3_want ? (size_t)n_tiles512 * useful_words : 0);
        HIP_TRY(W.d_zero.reserve(z_end));
        uint32_t *const d_dbottom = W.d_zero.p + z_dbottom, *const d_active = W.d_zero.p + z_active, *const d_queue = W.d_zero.p + z_queue,
                 *const d_list_n = W.d_zero.p + z_list_n, *const d_nitems = W.d_zero.p + z_nitems, *const d_cnt = W.d_zero.p + z_cnt,
                 *const d_key = W.d_zero.p + z_key, *const d_useful = b3_want ? W.d_zero.p + z_useful : nullptr;
        if (use8) {
            HIP_TRY(W.d_lbest.reserve((size_t)f.n_chunks * n_tiles512 * 256));
            HIP_TRY(W.d_list.reserve((size_t)f.n_chunks * n_tiles512));
            HIP_TRY(W.d_ub.reserve((size_t)n_tiles512 * 256));
            if (!coarse_only) {
                HIP_TRY(W.d_gbest.reserve((size_t)n_tiles512 * 256));
                HIP_TRY(W.d_gbest_part.reserve((size_t)ugp::GBEST_SLICES * n_tiles512 * 256));
                HIP_TRY(W.d_items.reserve(pairs));
            }
        } else if (mode == 0) {
            const size_t np = (size_t)n_tiles * G * 64;
            HIP_TRY(W.d_part_best.reserve(np));
            HIP_TRY(W.d_part_cnt.reserve(np));
            HIP_TRY(W.d_part_key.reserve(np));
        }
        if (int rc = ensure_events(TG, TG.events_used + 1)) return rc;
        EventSet &es = TG.events[TG.events_used++];
        es.b3_class = -1;   // (set behind this sub-batch's last event record: until then the events still hold their previous use)
        es.b3_used = b3_want; es.b3_tiles = n_tiles512; es.b3_pos = b3_pos; es.b3_seq = b3_seq; es.b3_first = m->b3_tuner.first;

        HIP_TRY(hipEventRecord(es.ev[0], s));
        const uint32_t *slot_of = nullptr, *order = nullptr;
        if (sorted) {
            HIP_TRY(W.d_keys.reserve(nq)); HIP_TRY(W.d_keys2.reserve(nq)); HIP_TRY(W.d_idx.reserve(nq));
            HIP_TRY(W.d_order.reserve(nq)); HIP_TRY(W.d_slot.reserve(nq)); HIP_TRY(W.d_bins.reserve(std::max<uint32_t>(m->n_coarse_bins, 1)));
            size_t tmp_bytes = 0;
            HIP_TRY(ugp::launch_locality_sort(nullptr, nullptr, (uint32_t)nq, W.d_keys.p, W.d_keys2.p, W.d_idx.p, W.d_order.p,
                                              W.d_slot.p, nullptr, &tmp_bytes, nullptr, 0, nullptr, s));
            HIP_TRY(W.d_sort_tmp.reserve(tmp_bytes));
            HIP_TRY(ugp::launch_locality_sort(W.d_coarse_res.p + q0, m->d_coarse2dfs.p, (uint32_t)nq, W.d_keys.p, W.d_keys2.p,
                                              W.d_idx.p, W.d_order.p, W.d_slot.p, W.d_sort_tmp.p, &tmp_bytes, K.radix_sort ? nullptr : m->d_coarse_bin.p, m->n_coarse_bins, W.d_bins.p, s));
            slot_of = W.d_slot.p; order = W.d_order.p;
        }
        HIP_TRY(hipMemsetAsync(W.d_zero.p, 0, z_end * sizeof(uint32_t), s));
        if (nmi >= 0) {
            HIP_TRY(ugp::launch_ntiles(W.d_table.p, d_active, active_words, n_tiles512, qs->d_nmask[nmi].p, qs->nmask_words[nmi], order, (uint32_t)q0, (uint32_t)nq,
                                       m->d_site_ref.p, n_sites, s));
            HIP_TRY(ugp::launch_scatter_list(W.d_table.p, d_dbottom, qs->d_pos.p, qs->d_ref.p, qs->d_nuc.p, qs->d_missing.p, qs->d_ent_q.p, m->d_pos2site.p,
                                             f.max_pos, n_sites, (uint32_t)q0, (uint32_t)nq, d_active, active_words, slot_of, qs->d_plain_rows.p, qs->d_n_plain.p, qs->d_err.p,
                                             d_useful, useful_words, s));
        } else if (lds_build)
            HIP_TRY(ugp::launch_build_tiles(W.d_table.p, d_active, active_words, n_tiles512, qs->d_ent_off.p, (uint32_t)q0, order, (uint32_t)nq, qs->d_pos.p,
                                            qs->d_ref.p, qs->d_nuc.p, qs->d_missing.p, m->d_pos2site.p, m->d_site_pos.p, m->d_site_ref.p, n_sites, f.max_pos,
                                            d_dbottom, qs->d_err.p, s));
        else {
        if (fill_ahead && q0 == 0) HIP_TRY(hipStreamWaitEvent(s, W.ev_fill, 0));
        else HIP_TRY(ugp::launch_fill_table(W.d_table.p, m->d_site_ref.p, n_sites, table_dwords, s));
        HIP_TRY(ugp::launch_scatter(W.d_table.p, d_dbottom, qs->d_pos.p + e0, qs->d_ref.p + e0,
                                    qs->d_nuc.p + e0, qs->d_missing.p + e0, qs->d_ent_q.p + e0, m->d_pos2site.p,
                                    f.max_pos, n_sites, e1 - e0, (uint32_t)q0, d_active, active_words, slot_of, qs->d_err.p, d_useful, useful_words, s));
        }
        // third pruning bound: the events of every tile's useful pairs -> block tables (ugp_bound3.hip)
        const bool b3_on = b3_want;
        if (b3_on) {
            const uint32_t nb = ugp::b3_blocks(m->stream8_dwords), n_l1 = ugp::b3_div64(nb), n_l2 = ugp::b3_div64(n_l1), n_l3 = ugp::b3_div64(n_l2);
            HIP_TRY(W.d_b3_pairmask.reserve((size_t)((n_tiles512 + 31) / 32) * n_sites * 4));
            HIP_TRY(W.d_b3_over.reserve((size_t)n_tiles512 * nb)); HIP_TRY(W.d_b3_under.reserve((size_t)n_tiles512 * nb));
            HIP_TRY(W.d_b3_l1.reserve((size_t)n_tiles512 * n_l1)); HIP_TRY(W.d_b3_l2.reserve((size_t)n_tiles512 * n_l2)); HIP_TRY(W.d_b3_l3.reserve((size_t)n_tiles512 * n_l3));
            HIP_TRY(W.d_b3_dev.reserve(1));
            const ugp::B3Dev hd{W.d_b3_over.p, W.d_b3_under.p, W.d_b3_l1.p, W.d_b3_l2.p, W.d_b3_l3.p, nb, n_l1, n_l2, n_l3};
            if (memcmp(&hd, &W.b3_host, sizeof hd) != 0) {   // (pointers and sizes: they change only when a buffer grows)
                HIP_TRY(hipStreamSynchronize(s));
                HIP_TRY(hipMemcpy(W.d_b3_dev.p, &hd, sizeof hd, hipMemcpyHostToDevice));
                W.b3_host = hd;
            }
            // (beside the seed descent, which needs the tiles but not the tables: the walk's launch joins the two)
            hipStream_t sb = s;
            if (can_fork) {
                HIP_TRY(hipEventRecord(W.ev_fork, s));
                HIP_TRY(hipStreamWaitEvent(W.aux, W.ev_fork, 0));
                sb = W.aux;
            }
            HIP_TRY(ugp::launch_b3_tables(d_useful, useful_words, n_sites, n_tiles512, m->d_b3_group_off.p, m->d_b3_events.p, nb, W.d_b3_pairmask.p,
                                          W.d_b3_over.p, W.d_b3_under.p, W.d_b3_l1.p, W.d_b3_l2.p, W.d_b3_l3.p, sb));
            if (can_fork) { HIP_TRY(hipEventRecord(W.ev_join, W.aux)); W.join_pending = true; }
        }
        // phase 2 without a walk for the samples whose minimum is attained by one node (ugp_kernels.hpp Phase2Uniq): the plain search only
        ugp::Phase2Uniq uq{};
        const bool uniq_ok = use8 && sorted && !coarse_only && mode == 0 && !ex && !d_tie_count && f.n_nodes < (1ull << 31) && !K.no_uniq;
        if (use8) {   // upper bounds of best(s) the pruning starts from
#ifdef UGP_EXPERIMENTS
            if (sorted && K.seed_prev && W.d_prev_res.cap >= Q && W.prev_serial == qs->serial)   // (experiment: bounds = the previous call's exact answers)
                HIP_TRY(ugp::launch_seed_ub(W.d_prev_res.p + q0, order, (uint32_t)nq, n_tiles512, W.d_ub.p, nullptr, nullptr, 0, nullptr, nullptr, nullptr, nullptr, s));
            else
#endif
            if (sorted && !K.no_seed) {
                // the coarse pass's best costs (real costs of real nodes), tightened by a greedy descent from the coarse best node
                // through the full tree (k_descend; it reads the sample's alleles from the tiles just built)
                const uint32_t *refined = nullptr;
                uint32_t *dnode = nullptr;
                // (the descent derives D of its start node from "cost(best_j) == best", which both forms of the pre-pass's result
                // guarantee)
                if (m->d_node_pair.p && m->d_coarse2bfs.p && !K.no_descent) {
                    HIP_TRY(W.d_refined.reserve(nq));
                    // (round 6) the descent also says WHICH node has the cost it reports: phase 2 answers the samples whose minimum is
                    // attained by one node from it, without a walk (Phase2Uniq; plain searches of trees below 2^31 nodes)
                    if (uniq_ok) { HIP_TRY(W.d_dnode.reserve(nq)); HIP_TRY(W.d_dres.reserve((size_t)n_tiles512 * 256)); dnode = W.d_dnode.p; }
                    HIP_TRY(ugp::launch_descend(W.d_coarse_res.p + q0, order, (uint32_t)nq, m->d_coarse2bfs.p, m->d_node_pair.p,
                                                m->d_parent.p, m->d_stream.p, W.d_table.p, n_sites, W.d_refined.p, m->wide_descent, K.descent_max, K.descent_slack,
                                                ex_skip ? ex_skip + q0 : nullptr, dnode, s));
                    refined = W.d_refined.p;
#ifdef UGP_EXPERIMENTS
                    if (K.stats && K.seed_check && W.prev_serial == qs->serial && W.d_prev_res.cap >= Q) {   // debug: seeds against the previous call's answers
                        std::vector<uint32_t> ref(nq), ord(nq);
                        std::vector<ugp_result> prev(nq), coarse(nq);
                        HIP_TRY(hipStreamSynchronize(s));
                        HIP_TRY(hipMemcpy(ref.data(), W.d_refined.p, nq * 4, hipMemcpyDeviceToHost));
                        HIP_TRY(hipMemcpy(ord.data(), order, nq * 4, hipMemcpyDeviceToHost));
                        HIP_TRY(hipMemcpy(prev.data(), W.d_prev_res.p + q0, nq * sizeof(ugp_result), hipMemcpyDeviceToHost));
                        HIP_TRY(hipMemcpy(coarse.data(), W.d_coarse_res.p + q0, nq * sizeof(ugp_result), hipMemcpyDeviceToHost));
                        uint64_t hist_r[18] = {0}, hist_c[18] = {0}, miss_anc = 0, miss_other = 0, miss_depth = 0;
                        std::vector<uint32_t> c2b(m->coarse ? m->coarse->flat.n_nodes : 0);
                        if (!c2b.empty()) HIP_TRY(hipMemcpy(c2b.data(), m->d_coarse2bfs.p, c2b.size() * 4, hipMemcpyDeviceToHost));
                        for (uint64_t k = 0; k < nq; k++) {
                            const int exact = prev[ord[k]].best_set_difference;
                            hist_r[std::min(17, std::max(0, (int)ref[k] - exact))]++;
                            hist_c[std::min(17, std::max(0, coarse[ord[k]].best_set_difference - exact))]++;
                            if ((int)ref[k] > exact && !c2b.empty() && coarse[ord[k]].best_j < c2b.size()) {   // is the coarse best node an ancestor of the true one?
                                const uint32_t j0 = c2b[coarse[ord[k]].best_j];
                                uint32_t x = prev[ord[k]].best_j, depth = 0;
                                while (x != UINT32_MAX && x != j0 && x != 0) { x = m->h_parent[x]; depth++; }
                                if (x == j0) { miss_anc++; miss_depth += depth; } else {
                                    miss_other++;
                                    if (miss_other <= 12) {   // a few examples: depths of j0, the true best and their lowest common ancestor
                                        auto depth_of = [&](uint32_t v) { uint32_t d = 0; while (v != 0 && v != UINT32_MAX) { v = m->h_parent[v]; d++; } return d; };
                                        uint32_t a = j0, b = prev[ord[k]].best_j;
                                        uint32_t da = depth_of(a), db = depth_of(b);
                                        const uint32_t da0 = da, db0 = db;
                                        while (da > db) { a = m->h_parent[a]; da--; }
                                        while (db > da) { b = m->h_parent[b]; db--; }
                                        while (a != b) { a = m->h_parent[a]; b = m->h_parent[b]; da--; }
                                        fprintf(stderr, "[ugp stats]   sample %llu: coarse cost %d at depth %u, exact %d (x%u) at depth %u, common ancestor at depth %u, descent %u\n",
                                                (unsigned long long)ord[k], coarse[ord[k]].best_set_difference, da0, exact, prev[ord[k]].num_best, db0, da, ref[k]);
                                    }
                                }
                            }
                        }
                        fprintf(stderr, "[ugp stats] loose seeds: coarse best is an ancestor of the true best for %llu (mean distance %.1f), is not for %llu\n",
                                (unsigned long long)miss_anc, miss_anc ? (double)miss_depth / miss_anc : 0.0, (unsigned long long)miss_other);
                        fprintf(stderr, "[ugp stats] seed - exact best, descent:");
                        for (int i = 0; i < 18; i++) fprintf(stderr, " %llu", (unsigned long long)hist_r[i]);
                        fprintf(stderr, "\n[ugp stats] seed - exact best, coarse: ");
                        for (int i = 0; i < 18; i++) fprintf(stderr, " %llu", (unsigned long long)hist_c[i]);
                        fprintf(stderr, "\n");
                    }
#endif
                }
                // (the unused slots of the last tile: far from everything, see k_seed_ub; 16-bit safe by the guard of the packed path)
                const uint32_t pad_d = K.no_pad_fix ? 0u : (uint32_t)std::min<uint64_t>(4096, 0x7F7Eu - 2u - std::min<uint64_t>(f.max_path_muts, 0x7F00u));
                HIP_TRY(ugp::launch_seed_ub(W.d_coarse_res.p + q0, order, (uint32_t)nq, n_tiles512, W.d_ub.p, refined,
                                            K.no_pad_fix ? nullptr : d_dbottom, pad_d, ex_skip ? ex_skip + q0 : nullptr, m->d_coarse2bfs.p, dnode, dnode ? W.d_dres.p : nullptr, s));
                if (dnode && refined) { uq.dnode = dnode; uq.refined = refined; uq.dres = W.d_dres.p; }
            } else
                HIP_TRY(hipMemsetAsync(W.d_ub.p, 0x7F, (size_t)n_tiles512 * 256 * sizeof(uint32_t), s));   // 0x7F7F: above every valid cost
        }
        HIP_TRY(hipEventRecord(es.ev[1], s));

        ugp::PlaceArgs a;
        memset(&a, 0, sizeof(a));
        a.stream = m->d_stream.p; a.pre_stream = m->d_pre.p;
        a.chunk_body_off = m->d_chunk_body.p; a.chunk_pre_off = m->d_chunk_pre.p; a.chunk_node_off = m->d_chunk_node.p;
        a.stream_t = m->d_stream_t.p; a.chunk_t_off = m->d_chunk_t.p;
        a.table = W.d_table.p; a.dbottom = d_dbottom;
        a.n_sites = n_sites; a.n_chunks = f.n_chunks; a.n_groups = G; a.n_tiles = n_tiles; a.n_queries = (uint32_t)nq;
        a.part_best = W.d_part_best.p; a.part_cnt = W.d_part_cnt.p; a.part_key = W.d_part_key.p;
        a.dfs2bfs = m->d_dfs2bfs.p; a.n_nodes = f.n_nodes;
        a.scores = d_scores ? d_scores + q0 * f.n_nodes : nullptr;
        a.best_in = d_best_in ? d_best_in + q0 : nullptr;
        a.tie_count = d_tie_count ? d_tie_count + q0 : nullptr;
        a.tie_j = d_tie_j ? d_tie_j + q0 * tie_cap : nullptr;
        a.tie_hu = d_tie_hu ? d_tie_hu + q0 * tie_cap : nullptr;
        a.tie_cap = tie_cap;
        if (ex) {
            a.node_mask = ex->mask; a.skip = ex->skip ? ex->skip + q0 : nullptr; a.alt_rank = ex->alt_rank; a.out_index = ex->out_index;
            if (mode == 0) a.scores = ex->scores ? ex->scores + q0 * f.n_nodes : nullptr;
        }
        if (use8) {
            ugp::Best8Args b;
            memset(&b, 0, sizeof(b));
            b.stream8 = m->d_stream8.p; b.pre8 = m->d_pre8.p;
            b.chunk8_body_off = m->d_chunk8_body.p; b.chunk8_pre_off = m->d_chunk8_pre.p;
            b.table = W.d_table.p; b.dbottom = d_dbottom;
            b.n_sites = n_sites; b.n_chunks = f.n_chunks; b.n_groups = G; b.n_tiles = n_tiles512;
            b.lbest = W.d_lbest.p;
            if (coarse_only) { HIP_TRY(W.d_lpos.reserve((size_t)f.n_chunks * n_tiles512 * 256)); b.lpos = W.d_lpos.p; }
            b.list = W.d_list.p; b.list_n = d_list_n;
            if (uq.dnode) {
                HIP_TRY(W.d_luniq.reserve((size_t)f.n_chunks * n_tiles512 * 16));   // (64 bytes per record)
                HIP_TRY(W.d_gcnt.reserve((size_t)n_tiles512 * 256)); HIP_TRY(W.d_gcnt_part.reserve((size_t)ugp::GBEST_SLICES * n_tiles512 * 256));
                b.luniq = W.d_luniq.p; uq.luniq = W.d_luniq.p; uq.gcnt = W.d_gcnt.p; uq.gcnt_part = W.d_gcnt_part.p;
            }
            W.last_list_n = d_list_n; W.last_list_tiles = n_tiles512; W.last_nitems = coarse_only ? nullptr : d_nitems;
            b.queue = d_queue;
            b.ub = K.no_prune ? nullptr : W.d_ub.p;
            const uint32_t *hstart = nullptr, *hlen = nullptr;
            if (sorted && !K.no_lpt) {   // hand out every tile's own region first (scheduling only)
                HIP_TRY(W.d_gstart.reserve(n_tiles512)); HIP_TRY(W.d_hlen.reserve(n_tiles512));
                HIP_TRY(ugp::launch_tile_ranges(W.d_keys2.p, (uint32_t)nq, n_tiles512, m->d_chunk_node.p, f.n_chunks,
                                                std::max<uint32_t>(1, (f.n_chunks + G - 1) / G), W.d_gstart.p, W.d_hlen.p, s));
                hstart = W.d_gstart.p; hlen = W.d_hlen.p;
            }
            const uint32_t unit_chunks = std::max<uint32_t>(1, (f.n_chunks + G - 1) / G);
            b.ub_every = 128;
            if (K.ub_every) b.ub_every = K.ub_every;
            b.freeze_ub = ex_skip ? 1u : 0u;   // (the chunk minima include the samples' excluded nodes: no bound may be taken from them)
            b.refill_all_rows = K.refill_all ? 1u : 0u;
            b.heavy_prio = K.heavy_prio;
            // (trees with large polytomies keep the tile-after-tile order: measured, 7 % apart in either direction)
            const uint32_t light_order = K.light_order >= 0 ? (uint32_t)K.light_order : (m->wide_descent ? 1u : 0u);
            uint32_t heavy_chunks = 16;
            if (K.heavy_chunks) heavy_chunks = K.heavy_chunks;
            // Units outside the tiles' own regions grow with the distance from the region (they end in their preamble or after a
            // few jumps: what they cost is the replay, not their length) -- only when there is a region to measure from and bounds
            // to prune with.
            uint32_t grow_every = (hstart && b.ub) ? 8u : 0u, unit_max = unit_chunks * 16u;
            if (K.unit_grow >= 0) grow_every = (hstart && b.ub) ? (uint32_t)K.unit_grow : 0u;
            if (K.unit_max) unit_max = K.unit_max;
            // A preamble record says where the body goes on behind a path node's subtree in INFO_JUMP_MASK's 18 bits, the largest
            // value meaning "beyond the unit": no unit may be longer than that many words.  (Should even the basic units be --
            // chunks of thousands of words: nodes with thousands of mutations -- the replay runs without those records.)
            {
                const uint64_t reach = ugp::INFO_JUMP_MASK - 1u, cw = std::max<uint32_t>(1, f.max_chunk8_words);
                unit_max = (uint32_t)std::max<uint64_t>(unit_chunks, std::min<uint64_t>(unit_max, reach / cw));
                b.no_pre_records = ((uint64_t)std::max(unit_chunks, heavy_chunks) * cw > reach) ? 1u : 0u;
            }
            {
                auto len_of = [&](uint32_t i) -> uint64_t {
                    if (!grow_every) return unit_chunks;
                    return std::min<uint64_t>((uint64_t)unit_chunks << std::min<uint32_t>(i / grow_every, 16u), std::max(unit_max, unit_chunks));
                };
                uint32_t n_side = 0;
                for (uint64_t done = 0; done < f.n_chunks; n_side++) done += len_of(n_side);
                const uint32_t per_tile_cap = (f.n_chunks + heavy_chunks - 1) / heavy_chunks + 2u * n_side + 2u;
                HIP_TRY(W.d_units.reserve((size_t)n_tiles512 * per_tile_cap * 4));
                HIP_TRY(W.d_unit_info.reserve(32 + 96));
                // units that run long are cut while they run: the shared list of split-off halves (k_best8)
                uint32_t split_cycles = 400000, split_heavy = 400000;
                if (K.split_cycles >= 0) split_cycles = (uint32_t)K.split_cycles;
                if (K.split_heavy >= 0) split_heavy = (uint32_t)K.split_heavy;
                if (!split_heavy) split_heavy = 0xFFFFFFFFu;
                if (!split_cycles || f.n_chunks >= (1u << 20) || n_tiles512 > 4096) split_cycles = split_heavy = 0xFFFFFFFFu;   // (never; the entry's fields)
                constexpr uint32_t kDynCap = 1u << 17;
                if (!W.d_dyn.p) {
                    HIP_TRY(W.d_dyn.reserve(kDynCap));
                    W.dyn_epoch = 2047;
                }
                if (++W.dyn_epoch >= 2048u) { HIP_TRY(hipMemsetAsync(W.d_dyn.p, 0, (size_t)kDynCap * 8, s)); W.dyn_epoch = 1; }   // (11 bits: stale entries never alias)
                uint32_t *dyn_ctl = W.d_unit_info.p + 32;
                b.dyn_ctl = dyn_ctl; b.dyn_units = (unsigned long long *)W.d_dyn.p; b.dyn_cap = kDynCap; b.dyn_epoch = W.dyn_epoch; b.split_cycles = split_cycles; b.split_heavy = split_heavy; b.split_dense = K.split_dense >= 0 ? (uint32_t)K.split_dense : 0xFFFFFFFFu; b.split_many = K.split_many ? ((K.split_many & 0xFFFFu) | (std::max(1u, K.split_many_heavy ? K.split_many_heavy : K.split_many) << 16)) : 0u;
                HIP_TRY(ugp::launch_build_units(hstart, hlen, n_tiles512, f.n_chunks, unit_chunks, heavy_chunks, grow_every, unit_max, light_order, per_tile_cap,
                                                W.d_units.p, W.d_unit_info.p, W.d_unit_info.p + 8, dyn_ctl, s));
                b.units = (const uint4 *)W.d_units.p; b.unit_base = W.d_unit_info.p; b.unit_count = W.d_unit_info.p + 8;
            }
            HIP_TRY(W.d_stats.reserve(96));
            if (q0 == 0) { if (K.stats) HIP_TRY(hipMemsetAsync(W.d_stats.p, 0, 96 * sizeof(uint64_t), s)); W.last_words_total = 0; }   // (the counters exist only in the statistics build: no launch for them otherwise)
            b.stats = K.stats ? W.d_stats.p : nullptr;   // the counters are two contended atomics per skip: debug only
            if (b.stats && !K.trace.empty() && !coarse_only) {   // per-unit records of this launch, dumped by ugp_get_timing
                constexpr size_t kTraceCap = 1u << 20;
                HIP_TRY(W.d_trace.reserve(8 + kTraceCap * 6));
                HIP_TRY(hipMemsetAsync(W.d_trace.p, 0, 64, s));
                b.trace = W.d_trace.p; b.trace_cap = kTraceCap;
            }
            W.last_words_total += (uint64_t)n_tiles512 * m->stream8_dwords;
            b.max_slots = f.max_slots;
            // LDS holds the hot slots only (the kernel's registers allow 6 waves per SIMD, 13 KB of LDS per wave
            // would stop at 3); the colder ones, touched once per ~1,300 words, go to a small global scratch
            b.lds_slots = f.lds_slots;   // fixed when the tree was flattened (headers touching colder slots are flagged there)
            // persistent grid: as many one-wave blocks as the device keeps resident (cached per handle and LDS
            // size), never more than there are units; the cold-slot scratch is sized for exactly that grid
            // The variant with the tile's active-row bitmap in LDS (up to 4 KB of it, i.e. 32,768 sites): a restart of the walk is two
            // memory round trips instead of three, for 3 KB more LDS per wave (13 resident waves per CU instead of 17).  Measured:
            // k_best8 4.52 -> 4.09 ms at 65,536 samples, where the launch is long enough to be bound by its throughput; 1.87 -> 1.97 ms
            // at 16,384, where the tail of the launch and the number of resident waves matter more.  Hence: from 64 tiles on.
            // (not when two calls share the device: the grids are halved then, and the LDS is better spent on resident waves --
            // 65,536 samples per call, pipelined: 10.5 M/s with, 11.0 M/s without)
            b.lds_bits = lds_bits_plan;   // (decided in front of the tile build, with the third bound)
            // Batches with thousands of N cells per sample (their tiles come from the N masks: nmi >= 0): every site row of every tile is
            // live, the bitmap says "fetch the row" for every word -- the variant without a bitmap saves each restart a dependent round
            // trip and each group a load (round 5; exact for any batch: a site's own row is always right, the constant row is the shortcut)
            // (nmi >= 0: lds_bits_plan == 2)
            const size_t lds_bytes = (size_t)b.lds_slots * 64 * 16 + (b.lds_bits == 1 ? (((size_t)active_words * 4 + 15) & ~(size_t)15) : 0);
            b.b3 = (b3_on && b.ub && b.lds_bits != 1) ? W.d_b3_dev.p : nullptr;
            if (b.b3) TG.last.bound3 = 1;
            const int variant = coarse_only ? (b.lds_bits == 2 ? 4 : 2) : (b.b3 ? (b.lds_bits == 2 ? 6 : 5) : (b.lds_bits == 2 ? 3 : (b.lds_bits ? 1 : 0)));   // (the kernel launch_best8 will pick)
            if (m->occ_lds != lds_bytes || m->occ_variant != variant) {
                HIP_TRY(hipDeviceGetAttribute(&m->n_cu, hipDeviceAttributeMultiprocessorCount, m->device));
                HIP_TRY(ugp::best8_occupancy(lds_bytes, variant, &m->occ_per_cu));
                m->occ_lds = lds_bytes; m->occ_variant = variant;
            }
            int waves_cu = std::max(m->occ_per_cu, 1);
            // Two calls on the device (ugp_place_device, the other set's call still running when this one is queued): each walk
            // takes half of what the device keeps resident, so that both grids ARE resident instead of one waiting for the other's
            // waves to exit -- measured at 16,384 samples per call: 2.71 -> 2.29 ms per call (6.05 -> 7.2 M placements/s), best at
            // 8 of the 17 waves per CU of that time (7: 2.32, 9: 2.39, 12: 2.51); with 16 resident since the B halves moved into registers, 8 again (6: 1.85, 8: 1.75, 10: 1.86 ms).  A call that finds the device to itself keeps the full grid.
            // (end of round 6, measured again with the smaller helpers, interleaved A/Bs in profiles/r06_ab_shared*.txt: a caller that keeps
            // THREE short calls in flight does best with 5 of the 16 slots per walk whatever the number of walks found running at the
            // moment -- 14.1-14.5 M/s against 13.6 with the share recomputed per call (5 or 8), 13.9 with 6, 13.8 with 4 or 7; the
            // SARS-CoV-2 shape 12.6 against 12.3.  Callers of long calls -- two sets: 65 536 samples, hundreds of rows per sample -- keep
            // half of the slots: 6.8 M/s against 5.8 with 5 on config 5.)
            if (m->sharing) waves_cu = std::max(1, K.shared_waves ? (int)K.shared_waves : (m->share_sets >= 3 ? waves_cu * 5 / 16 : waves_cu / 2));
            // (round 6) A lone call of a short, plain batch fills three quarters of the slots: with every slot taken more waves wait for
            // work at the end of the launch, every waiting wave makes a running one cut its unit, and every piece replays a preamble --
            // measured alone (profiles/r06_sweep_lone_waves.txt), 16 -> 12 waves per CU: 0.98 -> 0.89 ms at 16 384 x 10 M, 0.92 -> 0.77 at
            // 4 096, 0.76 -> 0.67 at 1 M nodes; batches of hundreds of rows per sample, 65 536 samples and the polytomy shape gain nothing or lose.
            else if (!coarse_only && sorted && b3_class == 0 && n_tiles512 <= 32 && !m->wide_descent && waves_cu >= 16) waves_cu = waves_cu * 3 / 4;
            if (K.waves_per_cu) waves_cu = std::max(1, std::min(std::max(m->occ_per_cu, 1), (int)K.waves_per_cu));   // tuning
            uint64_t blocks = (uint64_t)waves_cu * std::max(m->n_cu, 1);
            blocks = std::min<uint64_t>(blocks, (uint64_t)n_tiles512 * G);
            blocks = ((blocks + 7) / 8) * 8;
            HIP_TRY(W.d_cold.reserve((size_t)blocks * std::max<uint32_t>(f.max_slots - b.lds_slots, 1) * 512));   // 32 B per lane and cold slot
            b.cold = W.d_cold.p;
            b.active = d_active; b.active_words = active_words;
            // (UGP_KBEST_EXCLUSIVE: of the two calls that may be on the device at a time, ugp_place_device, only one runs this
            // kernel at any moment -- measured: 2.77 against 2.71 ms per step when the two persistent grids simply share the
            // chip; the small kernels in front of the second walk are slowed by the first and become the critical path)
            const bool exclusive = K.kbest_exclusive;
            if (exclusive && !coarse_only && m->kb_done && m->kb_done_on != s) HIP_TRY(hipStreamWaitEvent(s, m->kb_done, 0));
            if (W.join_pending) { HIP_TRY(hipStreamWaitEvent(s, W.ev_join, 0)); W.join_pending = false; }
            HIP_TRY(ugp::launch_best8(b, (uint32_t)blocks, s));
            if (exclusive && !coarse_only) {
                if (!m->kb_done) HIP_TRY(hipEventCreateWithFlags(&m->kb_done, hipEventDisableTiming));
                HIP_TRY(hipEventRecord(m->kb_done, s));
                m->kb_done_on = s;
            }
            HIP_TRY(hipEventRecord(es.ev[2], s));
            if (ex_skip && !coarse_only)   // the one chunk minimum per sample that its excluded node may have set, again without it
                HIP_TRY(ugp::launch_fix_skip(a, W.d_lbest.p, ex->skip_chunk + q0, n_tiles512, m->d_rank2bfs.p, order, f.max_slots, s));
            if (coarse_only)
                HIP_TRY(ugp::launch_coarse_result(W.d_lbest.p, W.d_lpos.p, W.d_list.p, d_list_n, f.n_chunks, n_tiles512, (uint32_t)nq, m->d_chunk_node.p,
                                                  m->d_chunk8_body.p, m->d_node_pos8.p, m->d_dfs2bfs.p, d_out + q0, s));
#ifdef UGP_EXPERIMENTS
            else if (!d_tie_count && m->d_node_pos8.p && m->d_rank_dfs.p && b.ub && !b.stats && K.phase2_packed) {
                // (experiment, UGP_PHASE2_PACKED=1) phase 2 as a mode of the packed walk: one unit per (tile, chunk) record that holds some
                // sample's global minimum.  Exact, but 6x slower than k_ties as it stands (1.8 against 0.3 ms per 16,384 samples): its
                // units re-walk half of their chunks; DESIGN.md 7.2
                HIP_TRY(W.d_tie_units.reserve((size_t)((n_tiles512 + 7) / 8) * 8 * f.n_chunks * 4));
                HIP_TRY(W.d_tie_info.reserve(128));
                HIP_TRY(ugp::launch_phase2_packed(b, W.d_list.p, d_list_n, W.d_gbest_part.p, W.d_gbest.p, n_tiles512, W.d_tie_units.p, W.d_tie_info.p, d_cnt, d_key,
                                                  m->d_node_pos8.p, m->d_rank_dfs.p, m->d_chunk_node.p, m->d_rank2bfs.p, (uint32_t)nq, d_out + q0, order,
                                                  (uint32_t)blocks, s));
            }
#endif
            else
                HIP_TRY(ugp::launch_phase2(a, W.d_lbest.p, W.d_list.p, d_list_n, W.d_gbest_part.p, W.d_gbest.p, n_tiles512, W.d_items.p, d_nitems,
                                           (uint32_t)std::min<uint64_t>(pairs, 0xFFFFFFFFull), d_cnt, d_key,
                                           m->d_rank2bfs.p, ex ? ex->rank2out : nullptr, d_out + q0, order, f.max_slots, d_tie_count != nullptr,
                                           uq.luniq ? &uq : nullptr, s));
                if (d_tie_count) m->tie_lists_filled++;
        } else if (mode == 1 && !ex && !m->h_level_off.empty() && !K.scores_dfs) {
            // -p in the output's own order: level by level of the breadth-first expansion, 64 consecutive scores of one sample per
            // wave store (k_scores_level); the depth-first walk below writes 4 bytes per 32-byte sector
            const uint32_t qpad = (uint32_t)((nq + 7) / 8 * 8);
            const bool d16 = qs->max_rows + f.max_path_muts + 2 < 0xFFFFull;
            const size_t d_words = ((size_t)m->max_level_width * ((qpad + ugp::SCORES_SB - 1) / ugp::SCORES_SB * ugp::SCORES_SB) * (d16 ? 2 : 4) + 3) / 4;
            HIP_TRY(W.d_part_best.reserve(d_words)); HIP_TRY(W.d_part_cnt.reserve(d_words));   // (the two D arrays; this mode has no partial results)
            HIP_TRY(ugp::launch_scores_levels(m->d_node_pair.p, m->d_parent.p, m->d_stream.p, W.d_table.p, n_sites, d_dbottom, m->h_level_off.data(),
                                              (uint32_t)m->h_level_off.size() - 1, W.d_part_best.p, W.d_part_cnt.p, d16, m->max_level_width, qpad, (uint32_t)nq, f.n_nodes,
                                              a.scores, K.scores_block, s));
            HIP_TRY(hipEventRecord(es.ev[2], s));
        } else {
            HIP_TRY(ugp::launch_place(a, ex ? mode + 4 : mode, f.max_slots, s));
            HIP_TRY(hipEventRecord(es.ev[2], s));
            if (mode == 0)
                HIP_TRY(ugp::launch_merge(W.d_part_best.p, W.d_part_cnt.p, W.d_part_key.p, (ex && ex->rank2out) ? ex->rank2out : m->d_rank2bfs.p, G,
                                          (uint32_t)nq, d_out + q0, s));
        }
        if ((K.seed_prev || K.seed_check) && use8 && !coarse_only && mode == 0) {
            HIP_TRY(W.d_prev_res.reserve(Q));
            HIP_TRY(hipMemcpyAsync(W.d_prev_res.p + q0, d_out + q0, nq * sizeof(ugp_result), hipMemcpyDeviceToDevice, s));
            W.prev_serial = (q0 + nq >= Q) ? qs->serial : 0;   // valid once every sub-batch of THIS query set has been stored
        }
        HIP_TRY(hipEventRecord(es.ev[3], s));
        if (b3_tuned) es.b3_class = b3_class;   // (from here on the tuner may read this sub-batch's events)
        W.last_used_best8 = use8;
        TG.last.packed_path = use8 ? 1u : 0u;
        es.used = true;
        TG.last.place_launches++;
        TG.last.n_tiles += use8 ? n_tiles512 : n_tiles;
        TG.last.n_groups = G;
    }
    return UGP_OK;
}

}  // namespace

extern "C" {

const char *ugp_last_error(void) { return g_err.c_str(); }

int ugp_device_warmup(int device) {
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr));   // (creates the context and loads this library's code objects)
    return UGP_OK;
}

// Host side of a handle: the flattened tree plus, for trees large enough to profit from the locality sort, the
// flattened coarse MAT (the top of the tree) and the map coarse BFS index -> DFS rank in the full tree.  Built
// once per tree whatever the number of devices it is uploaded to.
struct HostFlat {
    ugp::FlatMat f;
    std::vector<uint32_t> parent;   // copy of the caller's BFS parent array (extended searches need the topology)
    std::vector<uint32_t> coarse2dfs, coarse2bfs;
    ugp::UVec<uint32_t> node_pair;   // full tree, for the seed descent (empty: input not in breadth-first order)
    bool wide_descent = false;       // more than 5 % of the nodes hang off a node with more than 16 children
    HostFlat *coarse = nullptr;
    ~HostFlat() { delete coarse; }
};

static int host_flatten(const ugp_tree_desc *tree, const ugp::Options &opt, bool with_coarse, HostFlat &hf);

// The top of the tree (the nodes with the largest subtrees: N/4096 of them, at least 4096) as a MAT of its own.
// `ex` = subtree sizes and DFS positions from the flattening of the full tree.
static int build_coarse(const ugp_tree_desc *t, const ugp::Options &opt, ugp::FlatExtras &ex, HostFlat &hf) {
    const uint64_t N = t->n_nodes;
    uint64_t min_nodes = 1u << 18;   // below this a tree pass is too short for the sort to pay off
    if (const char *e = getenv("UGP_COARSE_MIN_NODES")) min_nodes = (uint64_t)atoll(e);   // tests lower it
    if (N < min_nodes || N < 64 || getenv("UGP_NO_SORT")) return UGP_OK;
    const uint32_t *sub = ex.sub.data();
    uint64_t div = 4096;   // (1024 until round 5: with the third bound and the descent behind it the walk does not notice a coarser start -- 0.93 ms either
                           // way -- and the pre-pass is a quarter shorter: 12.4 -> 12.9 M placements/s; below the floor of 4096 nodes the sort gets too coarse)
    if (const char *e = getenv("UGP_COARSE_DIV")) div = (uint64_t)std::max(2, atoi(e));
    uint64_t floor_n = 4096;
    if (const char *e = getenv("UGP_COARSE_FLOOR")) floor_n = (uint64_t)std::max(16, atoi(e));
    const uint64_t target = std::min<uint64_t>(N / 2, std::max<uint64_t>(N / div, std::min<uint64_t>(floor_n, N / 4)));
    // S = the (N - target)-th smallest subtree size (0-based): histogram of the sizes below 2^16 on the host threads,
    // a selection over a copy only when the threshold lies beyond it
    uint32_t S = 0;
    {
        constexpr uint32_t CAP = 1u << 16;
        const ugp::Par par = ugp::flatten_par(opt);
        std::vector<uint64_t> hist((size_t)par.T * (CAP + 1), 0);
        par.run(N, [&](uint64_t b, uint64_t e, unsigned tid) {
            uint64_t *h = &hist[(size_t)tid * (CAP + 1)];
            for (uint64_t j = b; j < e; j++) h[std::min<uint32_t>(sub[j], CAP)]++;
        }, 1u << 16);
        uint64_t below = 0;   // nodes with a size < v
        uint32_t v = 0;
        for (; v < CAP; v++) {
            uint64_t c = 0;
            for (unsigned th = 0; th < par.T; th++) c += hist[(size_t)th * (CAP + 1) + v];
            if (below + c > N - target) break;
            below += c;
        }
        if (v < CAP) S = v;
        else {
            std::vector<uint32_t> sorted_sub(sub, sub + N);
            std::nth_element(sorted_sub.begin(), sorted_sub.begin() + (N - target), sorted_sub.end());
            S = sorted_sub[N - target];
        }
        S = std::max<uint32_t>(2, S);
    }
    std::vector<uint32_t> keep;   // ascending BFS index; the parent of a kept node is kept (its subtree is larger)
    for (uint64_t j = 0; j < N; j++) if (sub[j] >= S || j == 0) keep.push_back((uint32_t)j);
    std::vector<uint32_t> parent(keep.size());
    std::vector<uint64_t> mut_off(keep.size() + 1, 0);
    std::vector<int32_t> pos; std::vector<uint8_t> ref, par, nuc;
    for (size_t k = 0; k < keep.size(); k++) {
        const uint32_t j = keep[k];
        parent[k] = j ? (uint32_t)(std::lower_bound(keep.begin(), keep.begin() + k, t->parent[j]) - keep.begin()) : UINT32_MAX;
        for (uint64_t i = t->mut_off[j]; i < t->mut_off[j + 1]; i++) {
            pos.push_back(t->mut_pos[i]); ref.push_back(t->mut_ref[i]);
            par.push_back(t->mut_par ? t->mut_par[i] : 0); nuc.push_back(t->mut_nuc[i]);
        }
        mut_off[k + 1] = pos.size();
    }
    ugp_tree_desc d{keep.size(), parent.data(), mut_off.data(), pos.data(), ref.data(), par.data(), nuc.data()};
    ugp::Options copt;
    copt.chunk_nodes = 256;   // (the coarse pass is bound by row fetches and by the replay in front of every chunk: long chunks)
    copt.keep_node_pos8 = true;
    copt.keep_update_maps = opt.keep_update_maps;
    copt.keep_b3_events = false;   // (the third bound serves the main walk only)
    if (const char *e = getenv("UGP_LDS_SLOTS")) copt.lds_slots = (uint32_t)std::max(1, std::min(60, atoi(e)));
    if (const char *e = getenv("UGP_COARSE_CHUNK_NODES")) copt.chunk_nodes = (uint32_t)std::max(1, atoi(e));
    hf.coarse = new HostFlat();
    if (int rc = host_flatten(&d, copt, false, *hf.coarse)) return rc;
    hf.coarse2dfs.resize(keep.size());
    for (size_t k = 0; k < keep.size(); k++) hf.coarse2dfs[k] = ex.dfsidx[keep[k]];
    hf.coarse2bfs = keep;
    return UGP_OK;
}

static int host_flatten(const ugp_tree_desc *tree, const ugp::Options &opt, bool with_coarse, HostFlat &hf) {
    std::string err;
    int rc;
    try {
        ugp::FlatExtras ex;
        rc = ugp::flatten(*tree, opt, hf.f, err, with_coarse ? &ex : nullptr);
        if (rc == UGP_OK && with_coarse) hf.parent.assign(tree->parent, tree->parent + tree->n_nodes);   // (the full tree only)
        if (rc == UGP_OK && with_coarse) {
            rc = build_coarse(tree, opt, ex, hf);
            if (rc != UGP_OK) return rc;   // (message already set)
            hf.node_pair.swap(ex.node_pair);   // (empty when the input is not a breadth-first expansion)
            if (hf.coarse) hf.wide_descent = ex.children_of_wide_nodes * 20 > tree->n_nodes;
        }
    } catch (const std::bad_alloc &) {
        return fail(UGP_ERR_NOMEM, "out of host memory while flattening the tree");
    }
    if (rc != UGP_OK) return fail(rc, err);
    return UGP_OK;
}

// Upload a flattened tree to one device.  The handle keeps the scalars of the flattening only.
static int upload_flat(const HostFlat &hf, int device, ugp_mat **out) {
    *out = nullptr;
    ugp_mat *m = new (std::nothrow) ugp_mat();
    if (!m) return fail(UGP_ERR_NOMEM, "out of host memory");
    m->device = device;
    m->knobs = ugp::Knobs::from_env();   // (the only place a handle reads its tuning switches: no getenv in a placement call)
    auto bail = [&](hipError_t e, const char *what) {
        std::string msg = std::string(what) + ": " + hipGetErrorString(e);
        ugp_mat_destroy(m);
        return fail(UGP_ERR_HIP, msg);
    };
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    const auto &f = hf.f;
    {   // scalars
        auto &g = m->flat;
        g.n_nodes = f.n_nodes; g.n_muts = f.n_muts; g.n_sites = f.n_sites; g.max_pos = f.max_pos; g.max_slots = f.max_slots;
        g.n_chunks = f.n_chunks; g.max_path_muts = f.max_path_muts; g.max_chunk8_words = f.max_chunk8_words; g.mask_not_first = f.mask_not_first; g.lds_slots = f.lds_slots;
    }
    m->upd.hdr8.assign(f.hdr8_of_bfs.begin(), f.hdr8_of_bfs.end());   // (empty unless the flattening kept them: Options::keep_update_maps)
    m->upd.rec.assign(f.rec_of_bfs.begin(), f.rec_of_bfs.end());
    m->upd.post.assign(f.post_of_bfs.begin(), f.post_of_bfs.end());
    m->upd.coarse2bfs = hf.coarse2bfs;
    if ((e = m->d_stream.upload(f.stream)) != hipSuccess) return bail(e, "upload stream");
    if ((e = m->d_pre.upload(f.pre_stream)) != hipSuccess) return bail(e, "upload preambles");
    if ((e = m->d_chunk_body.upload(f.chunk_body_off)) != hipSuccess) return bail(e, "upload chunk table");
    if ((e = m->d_chunk_pre.upload(f.chunk_pre_off)) != hipSuccess) return bail(e, "upload chunk table");
    if ((e = m->d_chunk_node.upload(f.chunk_node_off)) != hipSuccess) return bail(e, "upload chunk table");
    if ((e = m->d_rank2bfs.upload(f.rank2bfs)) != hipSuccess) return bail(e, "upload rank table");
    if ((e = m->d_dfs2bfs.upload(f.dfs2bfs)) != hipSuccess) return bail(e, "upload dfs table");
    if ((e = m->d_pos2site.upload(f.pos2site)) != hipSuccess) return bail(e, "upload site table");
    {
        std::vector<int32_t> sp(std::max<size_t>(f.n_sites, 1), 0);
        for (size_t p = 0; p < f.pos2site.size(); p++) if (f.pos2site[p] >= 0) sp[(size_t)f.pos2site[p]] = (int32_t)p;
        if ((e = m->d_site_pos.upload(sp)) != hipSuccess) return bail(e, "upload site table");
    }
    {
        std::vector<uint8_t> sr(f.site_ref);
        if (sr.empty()) sr.push_back(1);   // a tree without mutations still gets one (unused) table row
        if ((e = m->d_site_ref.upload(sr)) != hipSuccess) return bail(e, "upload site table");
    }
    if ((e = m->d_stream8.upload(f.stream8)) != hipSuccess) return bail(e, "upload packed stream");
    if ((e = m->d_pre8.upload(f.pre8_stream)) != hipSuccess) return bail(e, "upload packed preambles");
    if ((e = m->d_chunk8_body.upload(f.chunk8_body_off)) != hipSuccess) return bail(e, "upload chunk table");
    if ((e = m->d_chunk8_pre.upload(f.chunk8_pre_off)) != hipSuccess) return bail(e, "upload chunk table");
    if ((e = m->d_stream_t.upload(f.stream_t)) != hipSuccess) return bail(e, "upload tie stream");
    if ((e = m->d_chunk_t.upload(f.chunk_t_off)) != hipSuccess) return bail(e, "upload chunk table");
    if (!f.node_pos8.empty() && (e = m->d_node_pos8.upload(f.node_pos8)) != hipSuccess) return bail(e, "upload node positions");
    if (!f.rank_dfs.empty() && (e = m->d_rank_dfs.upload(f.rank_dfs)) != hipSuccess) return bail(e, "upload node ranks");
    if (!f.b3_events.empty()) {
        if ((e = m->d_b3_group_off.upload(f.b3_group_off)) != hipSuccess) return bail(e, "upload event lists");
        if ((e = m->d_b3_events.upload(f.b3_events)) != hipSuccess) return bail(e, "upload event lists");
    }
    m->stream8_dwords = f.stream8.size();
    m->stream_dwords = f.stream.size();
    m->pre_dwords = f.pre_stream.size();
    m->h_parent = hf.parent;
    if (!hf.node_pair.empty()) {
        // A breadth-first expansion: {first child - 1, record offset} per node and the parent array on the device (seed descent,
        // per-node scores by levels), and the level boundaries (a level is an index range).
        if ((e = m->d_node_pair.upload(hf.node_pair)) != hipSuccess) return bail(e, "upload node table");
        if ((e = m->d_parent.upload(hf.parent)) != hipSuccess) return bail(e, "upload parent table");
        const uint64_t N = f.n_nodes;
        std::vector<uint32_t> level(N, 0);
        m->h_level_off.assign(1, 0);
        for (uint64_t j = 1; j < N; j++) {
            level[j] = level[hf.parent[j]] + 1;
            if (level[j] != level[j - 1]) m->h_level_off.push_back((uint32_t)j);   // (levels are non-decreasing in a breadth-first numbering)
        }
        m->h_level_off.push_back((uint32_t)N);
        m->max_level_width = 0;
        for (size_t l = 0; l + 1 < m->h_level_off.size(); l++) m->max_level_width = std::max(m->max_level_width, m->h_level_off[l + 1] - m->h_level_off[l]);
    }
    if (hf.coarse) {
        if (int rc = upload_flat(*hf.coarse, device, &m->coarse)) { ugp_mat_destroy(m); return rc; }
        if ((e = m->d_coarse2dfs.upload(hf.coarse2dfs)) != hipSuccess) return bail(e, "upload coarse table");
        {   // bins of the counting sort of the samples (k_locality_sort1): the coarse nodes in depth-first order
            std::vector<uint32_t> by_rank(hf.coarse2dfs.size()), bin(hf.coarse2dfs.size());
            for (uint32_t i = 0; i < by_rank.size(); i++) by_rank[i] = i;
            std::sort(by_rank.begin(), by_rank.end(), [&](uint32_t a, uint32_t b) { return hf.coarse2dfs[a] < hf.coarse2dfs[b]; });
            for (uint32_t i = 0; i < by_rank.size(); i++) bin[by_rank[i]] = i;
            if ((e = m->d_coarse_bin.upload(bin)) != hipSuccess) return bail(e, "upload coarse table");
            m->n_coarse_bins = (uint32_t)bin.size();
        }
        // (always: the seeds of a search that leaves one node out per sample compare the coarse winner with that node, k_seed_ub --
        // also on a tree that is not numbered breadth-first, where the descent below does not run)
        if ((e = m->d_coarse2bfs.upload(hf.coarse2bfs)) != hipSuccess) return bail(e, "upload coarse table");
        if (!hf.node_pair.empty()) {
            m->wide_descent = getenv("UGP_DESCENT_LANES") ? atoi(getenv("UGP_DESCENT_LANES")) > 16 : hf.wide_descent;
        }
    }
    *out = m;
    return UGP_OK;
}

static ugp::Options default_options() {
    ugp::Options opt;
    opt.keep_update_maps = !getenv("UGP_NO_UPDATE_MAPS");   // 12 bytes per node on the host: what ugp_mat_update needs to exclude a rewritten node
    if (const char *e = getenv("UGP_CHUNK_NODES")) opt.chunk_nodes = (uint32_t)std::max(1, atoi(e));
    if (const char *e = getenv("UGP_PRUNE_MIN_WORDS")) opt.prune_min_words = (uint32_t)std::max(1, atoi(e));
    if (getenv("UGP_NO_SIB")) opt.sibling_records = false;
#ifdef UGP_EXPERIMENTS
    opt.keep_node_pos8 = getenv("UGP_PHASE2_PACKED") != nullptr;   // (the experiment's node tables: 8 bytes per node on the device)
#endif
    if (getenv("UGP_NO_BOUND2")) opt.second_bound = false;   // first lower bound only (tests, tuning)
    opt.keep_b3_events = !getenv("UGP_NO_BOUND3");           // posting lists of the third bound (8 bytes per mutation on the device)
    if (const char *e = getenv("UGP_LDS_SLOTS")) opt.lds_slots = (uint32_t)std::max(1, std::min(60, atoi(e)));
    if (const char *e = getenv("UGP_PRE_WEIGHT")) opt.pre_weight = (uint32_t)std::max(0, atoi(e));
    return opt;
}

static int mat_create_impl(const ugp_tree_desc *tree, int device, const ugp::Options &opt, ugp_mat **out) {
    if (!tree || !out) return fail(UGP_ERR_INVALID, "null argument");
    *out = nullptr;
    HostFlat hf;
    const bool verbose = getenv("UGP_FLATTEN_VERBOSE") != nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    if (int rc = host_flatten(tree, opt, true, hf)) return rc;
    const auto t1 = std::chrono::steady_clock::now();
    const int rc = upload_flat(hf, device, out);
    if (verbose)
        fprintf(stderr, "[ugp mat_create] flatten + coarse MAT %.1f ms, upload %.1f ms\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count());
    return rc;
}

// One flattening, n devices: the replicated read-only MAT of the multi-GPU path.  out[i] lives on
// devices[i]; on failure nothing is left allocated.
int ugp_mat_create_multi(const ugp_tree_desc *tree, const int *devices, int n_devices, ugp_mat **out) {
    if (!tree || !devices || !out || n_devices < 1) return fail(UGP_ERR_INVALID, "null argument");
    for (int i = 0; i < n_devices; i++) out[i] = nullptr;
    HostFlat hf;
    if (int rc = host_flatten(tree, default_options(), true, hf)) return rc;
    for (int i = 0; i < n_devices; i++) {
        if (int rc = upload_flat(hf, devices[i], &out[i])) {
            const std::string msg = g_err;
            for (int k = 0; k < i; k++) { ugp_mat_destroy(out[k]); out[k] = nullptr; }
            return fail(rc, msg);
        }
    }
    return UGP_OK;
}

int ugp_mat_create(const ugp_tree_desc *tree, int device, ugp_mat **out) {
    return mat_create_impl(tree, device, default_options(), out);
}

// ---- one flattening for several processes (one process per GPU: the ranks of a torch.distributed launch) ----------------------
// The flattening is host work whose result is the same for every rank.  ugp_flat_save writes it -- streams, tables, the coarse tree
// of the locality pre-pass -- to a file (on /dev/shm: a memory copy), ugp_mat_create_from_flat uploads it: an 8-rank launch
// flattens once instead of eight times over.  The file is valid for this library build and this environment's flattening switches
// only (both are hashed into its header; a mismatch is refused).
}   // extern "C" (templates below)
namespace {
constexpr uint64_t kFlatMagic = 0x3154414C46504755ull;   // "UGPFLAT1"
uint64_t flat_signature() {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const char *p) { for (; p && *p; p++) { h ^= (uint8_t)*p; h *= 1099511628211ull; } h ^= 0xFF; h *= 1099511628211ull; };
    mix(__DATE__ " " __TIME__);   // (this translation unit's build ...
    // ... and the format itself: an explicit version of the flattening's layouts plus the constants they are built from)
    const uint64_t fmt[] = {ugp::FLAT_FORMAT_VERSION, ugp::B3_BLOCK_SHIFT, ugp::B3_GROUP_SHIFT, ugp::T_PRUNE_MIN_DWORDS, ugp::PRUNE_MIN_WORDS, ugp::LDS_SLOTS, ugp::MAX_HOT_SLOTS,
                            ugp::INFO_JUMP_MASK, sizeof(ugp::FlatMat), sizeof(HostFlat), sizeof(ugp_result)};
    for (uint64_t v : fmt) { h ^= v; h *= 1099511628211ull; }
    for (const char *k : {"UGP_CHUNK_NODES", "UGP_PRUNE_MIN_WORDS", "UGP_NO_SIB", "UGP_NO_BOUND2", "UGP_LDS_SLOTS", "UGP_PRE_WEIGHT", "UGP_NO_UPDATE_MAPS",
                          "UGP_COARSE_MIN_NODES", "UGP_NO_SORT", "UGP_COARSE_DIV", "UGP_COARSE_FLOOR", "UGP_COARSE_CHUNK_NODES", "UGP_NO_BOUND3"}) { mix(k); mix(getenv(k)); }
    return h;
}
struct FlatWriter {
    FILE *f; bool ok = true;
    void raw(const void *p, size_t n) { if (ok && n && fwrite(p, 1, n, f) != n) ok = false; }
    template <class T> void pod(const T &v) { raw(&v, sizeof v); }
    template <class V> void vec(const V &v) { const uint64_t n = v.size(); pod(n); raw(v.data(), n * sizeof(typename V::value_type)); }
};
struct FlatReader {
    const uint8_t *p, *e; bool ok = true;
    void raw(void *d, size_t n) { if (!ok || (size_t)(e - p) < n) { ok = false; return; } memcpy(d, p, n); p += n; }
    template <class T> void pod(T &v) { raw(&v, sizeof v); }
    template <class V> void vec(V &v) {
        uint64_t n = 0; pod(n);
        if (!ok || n > (uint64_t)(e - p) / sizeof(typename V::value_type)) { ok = false; return; }
        v.resize(n); raw(v.data(), n * sizeof(typename V::value_type));
    }
};
template <class IO> void flat_io(IO &io, HostFlat &hf) {
    auto &f = hf.f;
    io.pod(f.n_nodes); io.pod(f.n_muts); io.pod(f.n_sites); io.pod(f.max_pos); io.pod(f.max_slots); io.pod(f.n_chunks);
    io.pod(f.max_path_muts); io.pod(f.max_chunk8_words); io.pod(f.lds_slots); io.pod(f.mask_not_first);
    io.vec(f.stream); io.vec(f.pre_stream); io.vec(f.chunk_body_off); io.vec(f.chunk_pre_off); io.vec(f.chunk_node_off); io.vec(f.pos2site);
    io.vec(f.site_ref); io.vec(f.rank2bfs); io.vec(f.dfs2bfs); io.vec(f.stream8); io.vec(f.pre8_stream); io.vec(f.chunk8_body_off); io.vec(f.chunk8_pre_off);
    io.vec(f.stream_t); io.vec(f.chunk_t_off); io.vec(f.rank_dfs); io.vec(f.node_pos8); io.vec(f.hdr8_of_bfs); io.vec(f.rec_of_bfs); io.vec(f.post_of_bfs);
    io.vec(f.b3_group_off); io.vec(f.b3_events);
    io.vec(hf.parent); io.vec(hf.coarse2dfs); io.vec(hf.coarse2bfs); io.vec(hf.node_pair); io.pod(hf.wide_descent);
}
}  // namespace
extern "C" {

int ugp_flat_save(const ugp_tree_desc *tree, const char *path) {
    if (!tree || !path) return fail(UGP_ERR_INVALID, "null argument");
    HostFlat hf;
    if (int rc = host_flatten(tree, default_options(), true, hf)) return rc;
    // (written under a temporary name and renamed: a reader never sees half a file)
    const std::string tmp = std::string(path) + ".tmp." + std::to_string((unsigned long long)getpid());
    FILE *fp = fopen(tmp.c_str(), "wb");
    if (!fp) return fail(UGP_ERR_INVALID, std::string("cannot write ") + tmp);
    FlatWriter w{fp};
    w.pod(kFlatMagic); const uint64_t sig = flat_signature(); w.pod(sig);
    const uint8_t has_coarse = hf.coarse ? 1 : 0;
    w.pod(has_coarse);
    flat_io(w, hf);
    if (hf.coarse) flat_io(w, *hf.coarse);
    const bool ok = (fclose(fp) == 0) && w.ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { (void)remove(tmp.c_str()); return fail(UGP_ERR_INVALID, std::string("short write to ") + path); }
    return UGP_OK;
}

int ugp_mat_create_from_flat(const char *path, int device, ugp_mat **out) {
    if (!path || !out) return fail(UGP_ERR_INVALID, "null argument");
    *out = nullptr;
    FILE *fp = fopen(path, "rb");
    if (!fp) return fail(UGP_ERR_INVALID, std::string("cannot read ") + path);
    std::vector<uint8_t> buf;
    try {
        if (fseek(fp, 0, SEEK_END) != 0) { fclose(fp); return fail(UGP_ERR_INVALID, "cannot size the flattening file"); }
        const long sz = ftell(fp);
        rewind(fp);
        buf.resize(sz > 0 ? (size_t)sz : 0);
        const bool ok = buf.empty() || fread(buf.data(), 1, buf.size(), fp) == buf.size();
        fclose(fp);
        if (!ok) return fail(UGP_ERR_INVALID, std::string("short read from ") + path);
        FlatReader r{buf.data(), buf.data() + buf.size()};
        uint64_t magic = 0, sig = 0; uint8_t has_coarse = 0;
        r.pod(magic); r.pod(sig); r.pod(has_coarse);
        if (!r.ok || magic != kFlatMagic) return fail(UGP_ERR_INVALID, "not a flattening written by ugp_flat_save");
        if (sig != flat_signature()) return fail(UGP_ERR_UNSUPPORTED, "the flattening was written by another library build or under other flattening switches");
        HostFlat hf;
        flat_io(r, hf);
        if (has_coarse) { hf.coarse = new HostFlat(); flat_io(r, *hf.coarse); }
        if (!r.ok || r.p != r.e) return fail(UGP_ERR_INVALID, "truncated or oversized flattening file");
        // (the file's contents index device tables: what the kernels trust is checked here)
        for (const HostFlat *x : {(const HostFlat *)&hf, (const HostFlat *)hf.coarse}) {
            if (!x) continue;
            const auto &g = x->f;
            if (g.site_ref.size() != g.n_sites || g.pos2site.size() != (size_t)g.max_pos + 1 || g.chunk8_body_off.size() != (size_t)g.n_chunks + 1 ||
                g.chunk_body_off.size() != (size_t)g.n_chunks + 1 || g.rank2bfs.size() != g.n_nodes || g.dfs2bfs.size() != g.n_nodes)
                return fail(UGP_ERR_INVALID, "inconsistent flattening file (array sizes)");
            for (int32_t v : g.pos2site) if (v >= (int64_t)g.n_sites) return fail(UGP_ERR_INVALID, "inconsistent flattening file (site index out of range)");
        }
        return upload_flat(hf, device, out);
    } catch (const std::bad_alloc &) { return fail(UGP_ERR_NOMEM, "out of host memory"); }
}

void ugp_mat_destroy(ugp_mat *m) {
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->knobs.stats || getenv("UGP_BOUND3_VERBOSE"))
        for (int c = 0; c < B3Tuner::kClasses; c++)
            if (m->b3_tuner.blocks[c])
                fprintf(stderr, "[ugp stats] third bound, batches of class %d: %u blocks; ms per tile with %.5f (%u blocks measured), without %.5f (%u)\n", c, m->b3_tuner.blocks[c],
                        m->b3_tuner.ema[c][1], m->b3_tuner.n[c][1], m->b3_tuner.ema[c][0], m->b3_tuner.n[c][0]);
    for (auto &W : m->work) {
        if (W.aux) { (void)hipStreamSynchronize(W.aux); (void)hipStreamDestroy(W.aux); }
        for (hipEvent_t e : {W.ev_fork, W.ev_fill, W.ev_join}) if (e) (void)hipEventDestroy(e);
        if (W.stream) { (void)hipStreamSynchronize(W.stream); (void)hipStreamDestroy(W.stream); }
        if (W.done) { (void)hipEventSynchronize(W.done); (void)hipEventDestroy(W.done); }
        for (auto &G : W.gens) {
            for (auto &es : G.events)
                for (int i = 0; i < 4; i++)
                    if (es.ev[i]) (void)hipEventDestroy(es.ev[i]);
            for (int i = 0; i < 2; i++) if (G.ev_coarse[i]) (void)hipEventDestroy(G.ev_coarse[i]);
        }
    }
    for (auto &W : m->work) delete W.job_qs;
    delete m->upd.qs;
    for (hipEvent_t e : m->entry_ring) if (e) (void)hipEventDestroy(e);
    if (m->kb_done) (void)hipEventDestroy(m->kb_done);
    if (m->coarse) ugp_mat_destroy(m->coarse);
    delete m->own_qs;
    delete m;
}

int ugp_mat_info(const ugp_mat *m, ugp_info *out) {
    if (!m || !out) return fail(UGP_ERR_INVALID, "null argument");
    const auto &f = m->flat;
    out->n_nodes = f.n_nodes;
    out->n_muts = f.n_muts;
    out->n_sites = f.n_sites;
    out->stream_bytes = m->stream8_dwords * 4;
    out->algo_tree_bytes = 4 * f.n_muts + 8 * f.n_nodes;
    out->algo_tile_bytes = (uint64_t)f.max_pos / 2 + 16;
    out->n_chunks = f.n_chunks;
    out->max_slots = f.max_slots;
    out->max_position = f.max_pos;
    out->device = (uint32_t)m->device;
    return UGP_OK;
}

// Upload the rows of a batch into `qs` (buffers grow on demand and are reused) and check them on the device.
// Translate the first offending row recorded by k_rows_prepare (~0: none) into an error.
static int rows_error(const ugp_qset *qs, const ugp_queries *q, unsigned long long err) {
    if (err == ~0ull) return UGP_OK;
    const uint64_t row = err >> 3;
    const uint64_t smp = (uint64_t)(std::upper_bound(qs->ent_off.begin(), qs->ent_off.end(), row) - qs->ent_off.begin()) - 1;
    switch ((int)(err & 7)) {
        case ugp::ROWS_UNSORTED:
            return fail(UGP_ERR_UNSUPPORTED, "rows of sample " + std::to_string(smp) + " are not sorted by position / contain a duplicate position");
        case ugp::ROWS_BAD_REF: return fail(UGP_ERR_UNSUPPORTED, "VCF REF base of a row is not one of A,C,G,T");
        case ugp::ROWS_BAD_MASK: return fail(UGP_ERR_INVALID, "allele mask out of range");
        default:
            return fail(UGP_ERR_UNSUPPORTED, "VCF REF differs from the tree's reference base at position " + (q ? std::to_string(q->pos[row]) : std::string("of row ") + std::to_string(row)));
    }
}

// `stage` (pinned host memory, or null): the rows go through it, so that the copies to the device are asynchronous on `s` and
// the caller's arrays are free on return; the row check is then left to the caller (qs->d_err, after the stream has run).
static int qset_fill(ugp_mat *m, const ugp_queries *q, ugp_qset *qs, hipStream_t s = nullptr, PinBuf *stage = nullptr) {
    uint64_t n_ent = 0, max_rows = 0;
    if (int rc = validate_offsets(q, n_ent, max_rows)) return rc;
    HIP_TRY(hipSetDevice(m->device));
    qs->device = m->device;
    qs->n_queries = q->n_queries;
    qs->n_ent = n_ent;
    qs->max_rows = max_rows;
    {
        static std::atomic<uint64_t> next_serial{0};
        qs->serial = ++next_serial;
    }
    try {
        if (q->n_queries) qs->ent_off.assign(q->ent_off, q->ent_off + q->n_queries + 1);
        else qs->ent_off.assign(1, 0);
    } catch (const std::bad_alloc &) {
        return fail(UGP_ERR_NOMEM, "out of host memory");
    }
    if (n_ent == 0) return UGP_OK;
    const auto &f = m->flat;
    HIP_TRY(qs->d_pos.reserve(n_ent)); HIP_TRY(qs->d_ref.reserve(n_ent)); HIP_TRY(qs->d_nuc.reserve(n_ent));
    HIP_TRY(qs->d_missing.reserve(n_ent)); HIP_TRY(qs->d_ent_q.reserve(n_ent));
    HIP_TRY(qs->d_ent_off.reserve(q->n_queries + 1)); HIP_TRY(qs->d_err.reserve(1));
    const void *h_pos = q->pos, *h_ref = q->ref, *h_nuc = q->nuc, *h_mis = q->is_missing, *h_off = q->ent_off;
    if (stage) {
        const size_t o_ref = n_ent * 4, o_nuc = o_ref + n_ent, o_mis = o_nuc + n_ent, o_off = (o_mis + n_ent + 7) & ~(size_t)7,
                     total = o_off + (q->n_queries + 1) * 8;
        if (int rc = stage->reserve(total)) return rc;
        char *b = (char *)stage->p;
        memcpy(b, q->pos, n_ent * 4); memcpy(b + o_ref, q->ref, n_ent); memcpy(b + o_nuc, q->nuc, n_ent); memcpy(b + o_mis, q->is_missing, n_ent);
        memcpy(b + o_off, q->ent_off, (q->n_queries + 1) * 8);
        h_pos = b; h_ref = b + o_ref; h_nuc = b + o_nuc; h_mis = b + o_mis; h_off = b + o_off;
    }
    HIP_TRY(hipMemcpyAsync(qs->d_pos.p, h_pos, n_ent * 4, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(qs->d_ref.p, h_ref, n_ent, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(qs->d_nuc.p, h_nuc, n_ent, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(qs->d_missing.p, h_mis, n_ent, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(qs->d_ent_off.p, h_off, (q->n_queries + 1) * 8, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemsetAsync(qs->d_err.p, 0xFF, sizeof(unsigned long long), s));
    HIP_TRY(ugp::launch_rows_prepare(qs->d_ent_off.p, (uint32_t)q->n_queries, n_ent, qs->d_pos.p, qs->d_ref.p, qs->d_nuc.p, qs->d_missing.p,
                                     m->d_pos2site.p, m->d_site_ref.p, f.max_pos, (uint32_t)f.n_sites, qs->d_ent_q.p, qs->d_err.p, s));
    // N masks: from 128 rows per sample on, for trees whose mask fits a workgroup's LDS (up to 524,288 sites)
    qs->nmask_for[0] = qs->nmask_for[1] = nullptr;
    {
        bool want = n_ent >= (uint64_t)q->n_queries * 128 && n_ent < (1ull << 32) && q->n_queries < (1ull << 32);
        if (m->knobs.nmask >= 0) want = m->knobs.nmask != 0 && n_ent < (1ull << 32);
        const ugp_mat *trees[2] = {m, m->coarse};
        for (int i = 0; want && i < 2; i++) {
            const ugp_mat *t = trees[i];
            if (!t || !t->flat.n_sites) continue;
            const uint32_t words = (uint32_t)(((t->flat.n_sites + 31) / 32 + 15) & ~(uint64_t)15);
            if ((size_t)words * 4 > 65536) continue;
            HIP_TRY(qs->d_nmask[i].reserve((size_t)q->n_queries * words));
            if (i == 0) { HIP_TRY(qs->d_plain_rows.reserve(n_ent)); HIP_TRY(qs->d_n_plain.reserve(1)); HIP_TRY(hipMemsetAsync(qs->d_n_plain.p, 0, 4, s)); }
            HIP_TRY(ugp::launch_nmask_build(qs->d_ent_off.p, (uint32_t)q->n_queries, qs->d_pos.p, qs->d_missing.p, t->d_pos2site.p, t->flat.max_pos, words,
                                            qs->d_nmask[i].p, i == 0 ? qs->d_plain_rows.p : nullptr, qs->d_n_plain.p, s));
            qs->nmask_for[i] = t; qs->nmask_words[i] = words;
        }
        if (!qs->nmask_for[0]) qs->nmask_for[1] = nullptr;   // (the row list comes with the first)
    }
    if (stage) return UGP_OK;   // (checked by the caller once the stream has run)
    unsigned long long err = ~0ull;
    HIP_TRY(hipMemcpy(&err, qs->d_err.p, sizeof err, hipMemcpyDeviceToHost));
    if (int rc = rows_error(qs, q, err)) return rc;
    return UGP_OK;
}

int ugp_qset_upload(ugp_mat *m, const ugp_queries *q, ugp_qset **out) {
    if (!m || !q || !out) return fail(UGP_ERR_INVALID, "null argument");
    *out = nullptr;
    ugp_qset *qs = new (std::nothrow) ugp_qset();
    if (!qs) return fail(UGP_ERR_NOMEM, "out of host memory");
    if (int rc = qset_fill(m, q, qs)) { (void)hipSetDevice(m->device); delete qs; return rc; }
    *out = qs;
    return UGP_OK;
}

void ugp_qset_destroy(ugp_qset *qs) {
    if (!qs) return;
    (void)hipSetDevice(qs->device);
    delete qs;
}

uint64_t ugp_qset_size(const ugp_qset *qs) { return qs ? qs->n_queries : 0; }

// Stream-ordered: behaves like a kernel launched on `stream` (waits for the work queued there before it, its results are
// visible to the work queued behind it).  One call at a time per handle; workspace set 0.
int ugp_place_device(ugp_mat *m, ugp_qset *qs, void *d_out, void *stream) {
    if (!m || !qs || (!d_out && qs->n_queries)) return fail(UGP_ERR_INVALID, "null argument");
    if (qs->device != m->device) return fail(UGP_ERR_INVALID, "query set lives on another device");
    return run_place(m, qs, 0, (ugp_result *)d_out, nullptr, nullptr, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
}

// Is the other workspace set's call still on the device?  (Then this call's tree walks leave it half of the chip.)  Or was it
// when the previous call was queued?  The first call of a burst finds the device idle; a caller that has just been issuing
// calls back to back is about to do so again.
static void note_sharing(ugp_mat *m, int wi, int depth) {
    int n_busy = 0;
    for (int o = 0; o < depth; o++) {
        if (o == wi) continue;
        const ugp_mat::Work &O = m->work[o];
        if (O.done && hipEventQuery(O.done) == hipErrorNotReady) n_busy++;
    }
    (void)hipGetLastError();   // (hipErrorNotReady is not an error)
    const bool busy = n_busy > 0;
    m->sharing = busy || m->was_busy;
    // a caller that keeps `depth` calls in flight has depth - 1 others on the device when the pipeline is full; while it fills
    // (or drains) the calls found running, or found by the previous call, say how many walks share the chip
    m->share_n = std::max(n_busy + 1, m->was_busy ? m->was_n : 1);
    m->share_sets = depth;   // the sets this caller cycles through: 3 for short calls, 2 for long ones
    m->was_busy = busy;
    m->was_n = n_busy + 1;
    if (m->knobs.debug_sharing) fprintf(stderr, "[ugp] call on set %d: sharing=%d among %d\n", wi, (int)m->sharing, m->share_n);
}

// The opt-in form: consecutive calls alternate between the handle's two workspace sets and run on two streams of its own,
// so that two batches are on the device at a time and the small latency-bound kernels around k_best8 of one fill the idle
// issue slots of the other's (measured +18 % placements/s at 16,384 samples per call, +50 % with the half-size grids).
// Ordering (include/usher_amd.h): `stream` receives every call's completion, in call order; a call is ordered behind the
// work that was queued on `stream` before the PREVIOUS overlapped call on this handle (one call of lag: work queued since
// then sits behind that call's completion and would serialise the two) -- and behind all of it when the handle is idle.
int ugp_pipeline_depth(const ugp_mat *m) { return m ? std::max(2, std::min(kMaxSets, (int)m->knobs.depth)) : 0; }

int ugp_place_device_overlapped(ugp_mat *m, ugp_qset *qs, void *d_out, void *stream) {
    if (!m || !qs || (!d_out && qs->n_queries)) return fail(UGP_ERR_INVALID, "null argument");
    if (qs->device != m->device) return fail(UGP_ERR_INVALID, "query set lives on another device");
    HIP_TRY(hipSetDevice(m->device));
    const int depth = std::max(2, std::min(kMaxSets, (int)m->knobs.depth));
    // Long calls gain nothing from a third batch on the device (measured: 65,536 samples per call 17.4 M/s with two, 16.9 with three;
    // thousands of N rows per sample 3.1 against 2.9), short ones do (16,384: 10.2 -> 11.3; 1,024 on a 100k-node tree 5.7 -> 6.8):
    // they cycle through two of the handle's sets.  The ordering promise is the handle's (depth - 1 calls of lag) either way.
    const int use = (qs->n_queries > 32768 || qs->n_ent > qs->n_queries * 128) ? 2 : depth;
    const int wi = m->next_work % use;
    m->next_work = (wi + 1) % use;
    ugp_mat::Work &W = m->work[wi];
    if (!W.stream) { ugp::fitch_drop_streams(m->device); HIP_TRY(hipStreamCreateWithFlags(&W.stream, hipStreamNonBlocking)); }   // (the Fitch-Sankoff pool's idle upload stream would shift this one's hardware queue)
    note_sharing(m, wi, use);
    struct Unshare { ugp_mat *m; ~Unshare() { m->sharing = false; m->share_n = 1; } } unshare{m};
    // What was on the caller's stream when this call was made (a ring of events by call number) is waited for by the call
    // depth - 1 calls later: that call has to wait for the call `depth` calls before it anyway (its workspaces), and whatever was
    // queued on `stream` in front of the call after THAT one sits behind its completion -- waiting for it serialises nothing.
    // When no other call is running this call waits for everything queued so far (nothing to overlap with).
    const uint64_t kc = m->n_overlapped++;
    hipEvent_t &mine = m->entry_ring[kc % kMaxSets];
    if (!mine) HIP_TRY(hipEventCreateWithFlags(&mine, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(mine, (hipStream_t)stream));
    if (!m->was_busy) HIP_TRY(hipStreamWaitEvent(W.stream, mine, 0));
    // (one more call of lag -- and one more output buffer for the caller to cycle through -- was measured: 12.3 -> 12.5 M/s, the
    // cross-queue signalling it hides is 2 % of a chain; not worth a wider contract)
    else if (kc + 1 >= (uint64_t)depth) HIP_TRY(hipStreamWaitEvent(W.stream, m->entry_ring[(kc - (uint64_t)(depth - 1)) % kMaxSets], 0));
    if (int rc = run_place(m, qs, 0, (ugp_result *)d_out, nullptr, nullptr, nullptr, nullptr, nullptr, 0, W.stream, false, nullptr, wi)) return rc;
    if (W.done) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, W.done, 0));   // (recorded at the end of run_place)
    return UGP_OK;
}

// ---- host buffers in, host buffers out, two batches in flight -------------------------------------------------------------
struct ugp_job {
    ugp_mat *m = nullptr;
    int wi = 0;
    ugp_result *out = nullptr;
    uint64_t n = 0;
};

int ugp_place_batch_async(ugp_mat *m, const ugp_queries *q, ugp_result *out, ugp_job **job) {
    if (!m || !q || !job || (!out && q->n_queries)) return fail(UGP_ERR_INVALID, "null argument");
    *job = nullptr;
    HIP_TRY(hipSetDevice(m->device));
    // as many jobs in flight as the handle keeps overlapped calls on the device (ugp_pipeline_depth; long batches: two, as there)
    const int depth = std::max(2, std::min(kMaxSets, (int)m->knobs.depth));
    const uint64_t n_rows = q->n_queries ? q->ent_off[q->n_queries] - q->ent_off[0] : 0;
    const int use = (q->n_queries > 32768 || n_rows > q->n_queries * 128) ? 2 : depth;
    // the next free set from the cursor on (jobs of different lengths cycle through different numbers of sets: a long job resets
    // the cursor of the short ones, and the set behind it may still hold a job although fewer than `use` are outstanding)
    int wi = -1;
    for (int i = 0; i < use && wi < 0; i++) if (!m->work[(m->next_job + i) % use].job_busy) wi = (m->next_job + i) % use;
    if (wi < 0) return fail(UGP_ERR_INVALID, "as many jobs as the handle keeps in flight (ugp_pipeline_depth; two for long batches) are outstanding: ugp_job_wait the oldest first");
    ugp_mat::Work &W = m->work[wi];
    if (W.job_busy) return fail(UGP_ERR_INVALID, "as many jobs as the handle keeps in flight (ugp_pipeline_depth) are outstanding: ugp_job_wait the oldest first");
    if (!W.stream) { ugp::fitch_drop_streams(m->device); HIP_TRY(hipStreamCreateWithFlags(&W.stream, hipStreamNonBlocking)); }   // (the Fitch-Sankoff pool's idle upload stream would shift this one's hardware queue)
    if (!W.job_qs) { W.job_qs = new (std::nothrow) ugp_qset(); if (!W.job_qs) return fail(UGP_ERR_NOMEM, "out of host memory"); }
    // the previous use of this set (a job two calls ago, or any other entry point) has to be over before its staging is overwritten
    if (W.done) HIP_TRY(hipEventSynchronize(W.done));
    if (int rc = qset_fill(m, q, W.job_qs, W.stream, &W.job_in)) return rc;
    HIP_TRY(W.d_job_out.reserve(q->n_queries));
    if (int rc = W.job_out.reserve(std::max<size_t>(q->n_queries, 1) * sizeof(ugp_result) + 8, true)) return rc;
    note_sharing(m, wi, use);   // (is the other set's job still on the device?)
    int rc = run_place(m, W.job_qs, 0, W.d_job_out.p, nullptr, nullptr, nullptr, nullptr, nullptr, 0, W.stream, false, nullptr, wi);
    m->sharing = false; m->share_n = 1;
    if (rc != UGP_OK) return rc;
    // results and the row check's verdict into pinned memory, behind the kernels; W.done is recorded again behind them.
    // By a KERNEL that stores into the pinned buffer, not by copy commands: the copy engine takes the copies of all streams in the
    // order they were queued, so a device -> host copy that waits for its batch's kernels held up the host -> device copies of
    // the next jobs -- and with them their whole pipelines: jobs "in flight" ran strictly one after the other (rocprofv3 trace;
    // 2.9 ms per 16,384-sample job with the copies, 1.39 without).
    if (q->n_queries && ugp::launch_copy_words((uint32_t *)W.job_out.p, (const uint32_t *)W.d_job_out.p, q->n_queries * (sizeof(ugp_result) / 4), W.stream) != hipSuccess)
        return fail(UGP_ERR_HIP, "copying the results out");
    if (W.job_qs->n_ent) {
        if (ugp::launch_copy_words((uint32_t *)((char *)W.job_out.p + q->n_queries * sizeof(ugp_result)), (const uint32_t *)W.job_qs->d_err.p, 2, W.stream) != hipSuccess)
            return fail(UGP_ERR_HIP, "copying the row check's verdict out");
    } else memset((char *)W.job_out.p + q->n_queries * sizeof(ugp_result), 0xFF, 8);
    HIP_TRY(hipEventRecord(W.done, W.stream));
    W.done_on = W.stream;
    ugp_job *j = new (std::nothrow) ugp_job();
    if (!j) return fail(UGP_ERR_NOMEM, "out of host memory");
    j->m = m; j->wi = wi; j->out = out; j->n = q->n_queries;
    W.job_busy = true;
    m->next_job = (wi + 1) % use;
    *job = j;
    return UGP_OK;
}

int ugp_job_wait(ugp_job *j) {
    if (!j) return fail(UGP_ERR_INVALID, "null argument");
    ugp_mat *m = j->m;
    ugp_mat::Work &W = m->work[j->wi];
    int rc = UGP_OK;
    if (hipSetDevice(m->device) != hipSuccess || hipEventSynchronize(W.done) != hipSuccess) rc = fail(UGP_ERR_HIP, "waiting for the job");
    if (rc == UGP_OK) {
        unsigned long long err;
        memcpy(&err, (char *)W.job_out.p + j->n * sizeof(ugp_result), 8);
        rc = rows_error(W.job_qs, nullptr, err);
        if (rc == UGP_OK && j->n) memcpy(j->out, W.job_out.p, j->n * sizeof(ugp_result));
    }
    W.job_busy = false;
    delete j;
    return rc;
}

int ugp_place_batch(ugp_mat *m, const ugp_queries *q, ugp_result *out) {
    if (!m || !q || (!out && q->n_queries)) return fail(UGP_ERR_INVALID, "null argument");
    // the handle's own query set and result buffer are reused from call to call (no hipMalloc in the steady state)
    if (!m->own_qs) { m->own_qs = new (std::nothrow) ugp_qset(); if (!m->own_qs) return fail(UGP_ERR_NOMEM, "out of host memory"); }
    ugp_qset *qs = m->own_qs;
    if (int rc = qset_fill(m, q, qs)) return rc;
    HIP_TRY(m->d_own_out.reserve(q->n_queries));
    if (int rc = run_place(m, qs, 0, m->d_own_out.p, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr)) return rc;
    if (q->n_queries) HIP_TRY(hipMemcpy(out, m->d_own_out.p, q->n_queries * sizeof(ugp_result), hipMemcpyDeviceToHost));
    return UGP_OK;
}

int ugp_scores_per_node(ugp_mat *m, const ugp_queries *q, int32_t *out) {
    if (!m || !q || (!out && q->n_queries)) return fail(UGP_ERR_INVALID, "null argument");
    ugp_qset *qs = nullptr;
    if (int rc = ugp_qset_upload(m, q, &qs)) return rc;
    const size_t total = (size_t)q->n_queries * m->flat.n_nodes;
    DevBuf<int32_t> d_scores;
    int rc = UGP_OK;
    hipError_t e = d_scores.reserve(total);
    if (e != hipSuccess) rc = fail(UGP_ERR_HIP, std::string("hipMalloc scores: ") + hipGetErrorString(e));
    if (rc == UGP_OK) rc = run_place(m, qs, 1, nullptr, d_scores.p, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
    if (rc == UGP_OK && total) {
        rc = copy_d2h_staged(out, d_scores.p, total * sizeof(int32_t));
    }
    ugp_qset_destroy(qs);
    return rc;
}

int ugp_tied_nodes(ugp_mat *m, const ugp_queries *q, uint32_t cap, uint32_t *tie_j, uint8_t *tie_has_unique,
                   uint32_t *tie_count) {
    if (!m || !q || !tie_count || (cap && (!tie_j || !tie_has_unique))) return fail(UGP_ERR_INVALID, "null argument");
    const uint64_t Q = q->n_queries;
    if (Q == 0) return UGP_OK;
    ugp_qset *qs = nullptr;
    if (int rc = ugp_qset_upload(m, q, &qs)) return rc;
    DevBuf<ugp_result> d_res;
    DevBuf<int32_t> d_best;
    DevBuf<uint32_t> d_cnt, d_j;
    DevBuf<uint8_t> d_hu;
    int rc = UGP_OK;
    const uint64_t padded = ((Q + 63) / 64) * 64;
    hipError_t e = hipSuccess;
    auto chk = [&](hipError_t x, const char *what) {
        if (rc == UGP_OK && x != hipSuccess) rc = fail(UGP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(x));
    };
    chk(d_res.reserve(Q), "hipMalloc");
    chk(d_best.reserve(Q), "hipMalloc");
    chk(d_cnt.reserve(padded), "hipMalloc");
    chk(d_j.reserve((size_t)Q * std::max<uint32_t>(cap, 1)), "hipMalloc");
    chk(d_hu.reserve((size_t)Q * std::max<uint32_t>(cap, 1)), "hipMalloc");
    // On the packed path the lists come out of phase 2 itself: the tied nodes lie in the chunks that attain the sample's minimum,
    // which phase 2 re-walks anyway (k_ties<LIST>) -- no second pass over the tree.  Otherwise (32-bit fallback): the best scores
    // first, then the one-sample-per-lane walk of the whole tree that appends every node attaining them.
    if (rc == UGP_OK) chk(hipMemsetAsync(d_cnt.p, 0, padded * sizeof(uint32_t), nullptr), "memset");
    m->tie_lists_filled = m->tie_sub_batches = 0;
    if (rc == UGP_OK) rc = run_place(m, qs, 0, d_res.p, nullptr, nullptr, m->knobs.ties_dfs ? nullptr : d_cnt.p, d_j.p, d_hu.p, cap, nullptr);
    const bool filled = rc == UGP_OK && m->tie_sub_batches > 0 && m->tie_lists_filled == m->tie_sub_batches;
    if (rc == UGP_OK && !filled) {   // the wanted scores stay on the device
        chk(ugp::launch_extract_best(d_res.p, (uint32_t)Q, d_best.p, nullptr), "extract best");
        chk(hipMemsetAsync(d_cnt.p, 0, padded * sizeof(uint32_t), nullptr), "memset");
        if (rc == UGP_OK) rc = run_place(m, qs, 2, nullptr, nullptr, d_best.p, d_cnt.p, d_j.p, d_hu.p, cap, nullptr);
    }
    if (rc == UGP_OK) {
        chk(hipMemcpy(tie_count, d_cnt.p, Q * sizeof(uint32_t), hipMemcpyDeviceToHost), "copy tie counts");
        if (cap) {
            chk(hipMemcpy(tie_j, d_j.p, (size_t)Q * cap * sizeof(uint32_t), hipMemcpyDeviceToHost), "copy ties");
            chk(hipMemcpy(tie_has_unique, d_hu.p, (size_t)Q * cap, hipMemcpyDeviceToHost), "copy ties");
        }
    }
    if (rc == UGP_OK && cap) {
        // ascending BFS index, the order usher_common.cpp:588 establishes
        std::vector<std::pair<uint32_t, uint8_t>> tmp;
        for (uint64_t i = 0; i < Q; i++) {
            const uint32_t k = std::min<uint32_t>(tie_count[i], cap);
            tmp.resize(k);
            for (uint32_t t = 0; t < k; t++) tmp[t] = {tie_j[i * cap + t], tie_has_unique[i * cap + t]};
            std::sort(tmp.begin(), tmp.end());
            for (uint32_t t = 0; t < k; t++) { tie_j[i * cap + t] = tmp[t].first; tie_has_unique[i * cap + t] = tmp[t].second; }
        }
    }
    (void)e;
    ugp_qset_destroy(qs);
    return rc;
}

// ---- extended searches: the other callers of mapper2_body --------------------------------------------------

// The reference's depth-first expansion (mutation_annotated_tree.cpp:1253-1273: preorder, children in stored order).
static void ensure_dfs_order(ugp_mat *m) {
    if (!m->h_dfs2bfs.empty()) return;
    const uint64_t N = m->flat.n_nodes;
    const auto &par = m->h_parent;
    // Child lists by count, prefix sum and fill (children of a node in increasing index = the reference's stored order).
    // A true breadth-first expansion has them contiguous (1 + first[j] ..), but the boundary only promises
    // parent[j] < j, and ugp_flatten.cpp accepts such topological-but-not-level-ordered arrays too.
    std::vector<uint32_t> first(N + 1, 0), kids(N ? N - 1 : 0);
    for (uint64_t j = 1; j < N; j++) first[par[j] + 1]++;
    for (uint64_t j = 0; j < N; j++) first[j + 1] += first[j];   // the children of j are kids[first[j] .. first[j+1])
    {
        std::vector<uint32_t> fill(first.begin(), first.end() - 1);
        for (uint64_t j = 1; j < N; j++) kids[fill[par[j]]++] = (uint32_t)j;
    }
    m->h_dfs2bfs.resize(N); m->h_bfs2dfs.resize(N);
    std::vector<std::pair<uint32_t, uint32_t>> st;   // (node, next child offset)
    st.push_back({0u, 0u});
    uint64_t d = 0;
    m->h_dfs2bfs[d] = 0; m->h_bfs2dfs[0] = 0; d++;
    while (!st.empty()) {
        auto &fr = st.back();
        const uint32_t b = first[fr.first], e = first[fr.first + 1];
        if (b + fr.second < e) {
            const uint32_t c = kids[b + fr.second++];
            m->h_dfs2bfs[d] = c; m->h_bfs2dfs[c] = (uint32_t)d; d++;
            st.push_back({c, 0u});
        } else st.pop_back();
    }
}

int ugp_node_order(ugp_mat *m, uint32_t order, uint32_t *bfs_of) {
    if (!m || !bfs_of) return fail(UGP_ERR_INVALID, "null argument");
    const uint64_t N = m->flat.n_nodes;
    if (order == UGP_ORDER_BFS) { for (uint64_t j = 0; j < N; j++) bfs_of[j] = (uint32_t)j; return UGP_OK; }
    if (order != UGP_ORDER_DFS) return fail(UGP_ERR_INVALID, "unknown node order");
    try { ensure_dfs_order(m); } catch (const std::bad_alloc &) { return fail(UGP_ERR_NOMEM, "out of host memory"); }
    memcpy(bfs_of, m->h_dfs2bfs.data(), N * sizeof(uint32_t));
    return UGP_OK;
}

int ugp_subtree_mask(ugp_mat *m, uint32_t order, uint32_t root_j, uint32_t max_levels, uint8_t *mask_out) {
    if (!m || !mask_out) return fail(UGP_ERR_INVALID, "null argument");
    const uint64_t N = m->flat.n_nodes;
    if (root_j >= N || order > UGP_ORDER_DFS) return fail(UGP_ERR_INVALID, "node index / order out of range");
    try {
        if (order == UGP_ORDER_DFS) ensure_dfs_order(m);
        const uint32_t root = order == UGP_ORDER_DFS ? m->h_dfs2bfs[root_j] : root_j;
        // depth below `root` in BFS order (parents first); UINT32_MAX = outside the subtree
        std::vector<uint32_t> depth(N, UINT32_MAX);
        depth[root] = 0;
        for (uint64_t j = (uint64_t)root + 1; j < N; j++) {
            const uint32_t dp = depth[m->h_parent[j]];
            if (dp != UINT32_MAX) depth[j] = dp + 1;
        }
        for (uint64_t j = 0; j < N; j++) {
            const uint64_t o = order == UGP_ORDER_DFS ? m->h_bfs2dfs[j] : j;
            mask_out[o] = depth[j] != UINT32_MAX && depth[j] <= max_levels;   // merge.cpp:254: level - level(root) > max_levels is skipped
        }
    } catch (const std::bad_alloc &) { return fail(UGP_ERR_NOMEM, "out of host memory"); }
    return UGP_OK;
}

namespace {
// Everything an extended call needs on the device, converted from the caller's node order to BFS indexing.
struct ExHost {
    DevBuf<uint8_t> d_mask;
    DevBuf<uint32_t> d_skip, d_skip_chunk, d_rank, d_rank2out, d_out_index;
    DevBuf<int32_t> d_scores;
    ExDev dev;
};

// The per-sample part of the options: the excluded nodes (and the chunks that hold them).  `x` receives dev.skip / dev.skip_chunk.
int prepare_samples(ugp_mat *m, uint64_t Q, uint32_t order, const uint32_t *skip_node, ExHost &x) {
    const uint64_t N = m->flat.n_nodes;
    const bool dfs = order == UGP_ORDER_DFS;
    HIP_TRY(hipSetDevice(m->device));
    try {
        if (dfs) ensure_dfs_order(m);
        auto to_bfs = [&](uint64_t k) -> uint32_t { return dfs ? m->h_dfs2bfs[k] : (uint32_t)k; };
        const ugp_place_opts oo{order, nullptr, skip_node, nullptr, nullptr};
        const ugp_place_opts *o = &oo;
        if (o->skip_node) {
            std::vector<uint32_t> sk(Q);
            for (uint64_t i = 0; i < Q; i++) {
                if (o->skip_node[i] != UINT32_MAX && o->skip_node[i] >= N) return fail(UGP_ERR_INVALID, "skip_node out of range");
                sk[i] = o->skip_node[i] == UINT32_MAX ? UINT32_MAX : to_bfs(o->skip_node[i]);
            }
            HIP_TRY(x.d_skip.upload(sk));
            x.dev.skip = x.d_skip.p;
            // the packed path needs to know which chunk of the flattened tree holds each excluded node (k_fix_skip)
            // (the flattener's own depth-first order and chunk cuts: the handle keeps them on the device only; fetched once)
            if (m->d_dfs2bfs.p && m->d_chunk_node.p) {
                if (m->h_flat_bfs2dfs.size() != N) {
                    std::vector<uint32_t> d2b(N);
                    m->h_flat_chunk_node.resize((size_t)m->flat.n_chunks + 1);
                    HIP_TRY(hipMemcpy(d2b.data(), m->d_dfs2bfs.p, N * sizeof(uint32_t), hipMemcpyDeviceToHost));
                    HIP_TRY(hipMemcpy(m->h_flat_chunk_node.data(), m->d_chunk_node.p, m->h_flat_chunk_node.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                    m->h_flat_bfs2dfs.resize(N);
                    for (uint64_t d = 0; d < N; d++) m->h_flat_bfs2dfs[d2b[d]] = (uint32_t)d;
                }
                std::vector<uint32_t> sc(Q, UINT32_MAX);
                const auto &cno = m->h_flat_chunk_node;
                for (uint64_t i = 0; i < Q; i++)
                    if (sk[i] != UINT32_MAX) sc[i] = (uint32_t)(std::upper_bound(cno.begin(), cno.end(), m->h_flat_bfs2dfs[sk[i]]) - cno.begin()) - 1u;
                HIP_TRY(x.d_skip_chunk.upload(sc));
                x.dev.skip_chunk = x.d_skip_chunk.p;
            }
        }
    } catch (const std::bad_alloc &) { return fail(UGP_ERR_NOMEM, "out of host memory"); }
    return UGP_OK;
}

// The node-level part of the options -- the caller's node order, its mask, its distances: what ripples keeps for a whole run
// (ripples/main.cpp:303-377) and ugp_ex_prepare therefore does once.  `x` receives dev.mask / alt_rank / rank2out / out_index.
int prepare_nodes(ugp_mat *m, const ugp_place_opts *o, ExHost &x) {
    const uint64_t N = m->flat.n_nodes;
    if (o->order > UGP_ORDER_DFS) return fail(UGP_ERR_INVALID, "unknown node order");
    if (m->h_parent.size() != N) return fail(UGP_ERR_INVALID, "this handle has no host topology (created from a coarse tree?)");
    const bool dfs = o->order == UGP_ORDER_DFS;
    HIP_TRY(hipSetDevice(m->device));
    try {
        if (dfs) ensure_dfs_order(m);
        auto to_bfs = [&](uint64_t k) -> uint32_t { return dfs ? m->h_dfs2bfs[k] : (uint32_t)k; };
        if (o->node_mask) {
            std::vector<uint8_t> mk(N);
            for (uint64_t k = 0; k < N; k++) mk[to_bfs(k)] = o->node_mask[k] ? 1 : 0;
            HIP_TRY(x.d_mask.upload(mk));
            x.dev.mask = x.d_mask.p;
        }
        if (dfs || o->distance) {
            // tie rank of usher_mapper.cpp:483-486 in the caller's terms: smaller distance, then more descendant leaves,
            // then the larger index j of the caller's node vector; rank = position in ascending order of "how good"
            const bool cached = dfs && !o->distance && m->dfs_rank_ready;
            if (!cached) {
                if (m->h_leaves.size() != N) {   // leaves below every node (once per handle)
                    std::vector<uint32_t> nch(N, 0);
                    m->h_leaves.assign(N, 0);
                    for (uint64_t j = 1; j < N; j++) nch[m->h_parent[j]]++;
                    for (uint64_t j = N; j-- > 0;) { if (!nch[j]) m->h_leaves[j] = 1; if (j) m->h_leaves[m->h_parent[j]] += m->h_leaves[j]; }
                }
                const std::vector<uint32_t> &leaves = m->h_leaves;
                const uint32_t *dist = o->distance;
                // worst -> best: larger distance first, then fewer leaves, then the smaller caller index -- one stable radix sort
                // of 64-bit keys {~distance, leaves} over the caller indices in ascending order, on the device (a comparison sort
                // of 10M nodes on the host was 1.5 s of a 1.6 s call)
                std::vector<uint64_t> keys(N);
                for (uint64_t k = 0; k < N; k++) keys[k] = ((uint64_t)(dist ? 0xFFFFFFFFu - dist[k] : 0u) << 32) | leaves[to_bfs(k)];
                DevBuf<uint32_t> &dr = (dfs && !dist) ? m->d_dfs_rank : x.d_rank, &d2o = (dfs && !dist) ? m->d_dfs_rank2out : x.d_rank2out;
                DevBuf<uint64_t> d_keys, d_keys2;
                DevBuf<uint32_t> d_iota;
                DevBuf<uint8_t> d_tmp;
                HIP_TRY(d_keys.upload(keys)); HIP_TRY(d_keys2.reserve(N)); HIP_TRY(d_iota.reserve(N)); HIP_TRY(dr.reserve(N)); HIP_TRY(d2o.reserve(N));
                if (dfs && !m->d_dfs2bfs_caller.p) HIP_TRY(m->d_dfs2bfs_caller.upload(m->h_dfs2bfs));
                size_t tmp_bytes = 0;
                HIP_TRY(ugp::launch_rank_sort(nullptr, &tmp_bytes, d_keys.p, d_keys2.p, d_iota.p, d2o.p, (uint32_t)N, nullptr, nullptr, nullptr));
                HIP_TRY(d_tmp.reserve(tmp_bytes));
                HIP_TRY(ugp::launch_rank_sort(d_tmp.p, &tmp_bytes, d_keys.p, d_keys2.p, d_iota.p, d2o.p, (uint32_t)N, dfs ? m->d_dfs2bfs_caller.p : nullptr, dr.p, nullptr));
                HIP_TRY(hipStreamSynchronize(nullptr));
                if (dfs && !dist) m->dfs_rank_ready = true;
            }
            const bool use_cache = dfs && !o->distance;
            x.dev.alt_rank = use_cache ? m->d_dfs_rank.p : x.d_rank.p;
            x.dev.rank2out = use_cache ? m->d_dfs_rank2out.p : x.d_rank2out.p;
        }
        if (dfs) {
            if (!m->d_bfs2dfs.p) HIP_TRY(m->d_bfs2dfs.upload(m->h_bfs2dfs));
            x.dev.out_index = m->d_bfs2dfs.p;
        }
    } catch (const std::bad_alloc &) { return fail(UGP_ERR_NOMEM, "out of host memory"); }
    return UGP_OK;
}

int prepare_ex(ugp_mat *m, const ugp_queries *q, const ugp_place_opts *o, ExHost &x, bool want_scores) {
    const uint64_t N = m->flat.n_nodes, Q = q->n_queries;
    if (int rc = prepare_nodes(m, o, x)) return rc;
    if (int rc = prepare_samples(m, Q, o->order, o->skip_node, x)) return rc;
    if (want_scores && o->scores && Q) {
        HIP_TRY(x.d_scores.reserve((size_t)Q * N));
        HIP_TRY(hipMemset(x.d_scores.p, 0, (size_t)Q * N * sizeof(int32_t)));
        x.dev.scores = x.d_scores.p;
    }
    return UGP_OK;
}
}  // namespace

namespace {
int drain(ugp_mat *m);
// Can this extended search run on the packed, pruned path?  Its node order / distance becomes the tie rank of phase 2; its node
// mask becomes a temporary exclusion (the "no candidate" bit of ugp_mat_update, set for the call and cleared behind it -- the
// pruning bounds stay valid with fewer candidates, and the coarse tree of the locality pre-pass is masked alike, so that its
// seeds are costs of admitted nodes); a per-sample excluded node is handled behind phase 1 (k_fix_skip: the chunk minimum it may
// have set is recomputed without it; seeds and bounds never rely on it).  Not with per-node scores, not when the mask drops the
// root (it always scores), not on a handle that carries exclusions of its own.
bool ex_packs(const ugp_mat *m, const ugp_place_opts *o) {
    if (o->scores || m->knobs.ex_slow) return false;
    if (o->node_mask && (!o->node_mask[0] || m->upd.rec.size() != m->flat.n_nodes || m->upd.n_excluded || m->flat.n_nodes >= (1ull << 30))) return false;
    return true;
}
int mask_words(ugp_mat *m, const uint8_t *d_mask, bool set) {
    // (the unmask pass clears the "no candidate" bit of every node outside the mask, also one that ugp_mat_update set for good:
    // a handle with exclusions of its own never comes here -- ex_packs -- and this keeps it so)
    if (m->upd.n_excluded) return fail(UGP_ERR_UNSUPPORTED, "node masks on the packed path need a handle without ugp_mat_update exclusions");
    ugp_mat *trees[2] = {m, m->coarse};
    for (ugp_mat *t : trees) {
        if (!t) continue;
        auto &U = t->upd;
        if (U.rec.size() != t->flat.n_nodes) { if (t == m) return fail(UGP_ERR_UNSUPPORTED, "no update maps"); continue; }
        if (!U.d_rec_pos.p) { HIP_TRY(U.d_hdr8.upload(U.hdr8)); HIP_TRY(U.d_rec_pos.upload(U.rec)); HIP_TRY(U.d_post.upload(U.post)); }
        HIP_TRY(ugp::launch_mask_words(d_mask, t == m ? nullptr : m->d_coarse2bfs.p, (uint32_t)t->flat.n_nodes, U.d_hdr8.p, U.d_rec_pos.p, U.d_post.p,
                                       t->d_stream8.p, t->d_stream.p, t->d_stream_t.p, ugp::H_NOSCORE, set, nullptr));
    }
    return UGP_OK;
}
}  // namespace

int ugp_place_batch_ex(ugp_mat *m, const ugp_queries *q, const ugp_place_opts *opts, ugp_result *out) {
    if (!m || !q || !opts || (!out && q->n_queries)) return fail(UGP_ERR_INVALID, "null argument");
    if (q->n_queries == 0) return UGP_OK;
    ugp_qset *qs = nullptr;
    if (int rc = ugp_qset_upload(m, q, &qs)) return rc;
    ExHost x;
    DevBuf<ugp_result> d_out;
    int rc = prepare_ex(m, q, opts, x, true);
    if (rc == UGP_OK && d_out.reserve(q->n_queries) != hipSuccess) rc = fail(UGP_ERR_HIP, "hipMalloc results");
    // The score matrix of a search in breadth-first indices (ripples: ripples/main.cpp:343-377) does not need the one-sample-per-lane
    // kernel either: the results come from the packed path as without it, the scores from the level-by-level kernel of -p (a node's
    // score does not depend on who else is a candidate), and the nodes that were not scored -- outside the mask, a sample's excluded
    // node -- are set to 0 behind it.
    ugp_place_opts no_scores = *opts;
    no_scores.scores = nullptr;
    const bool scores_by_levels = rc == UGP_OK && opts->scores && opts->order == UGP_ORDER_BFS && !m->h_level_off.empty() && !m->knobs.scores_dfs &&
                                  !m->knobs.ex_slow && !m->knobs.force_v1 && !m->upd.n_excluded;
    const ugp_place_opts *popts = scores_by_levels ? &no_scores : opts;
    const bool pack = rc == UGP_OK && ex_packs(m, popts) && (!m->coarse || m->d_coarse2bfs.p || !opts->node_mask);
    if (pack) {
        if (rc == UGP_OK) rc = drain(m);
        ExDev xd = x.dev;
        if (rc == UGP_OK && xd.mask) rc = mask_words(m, xd.mask, true);
        const uint8_t *masked = xd.mask;
        xd.mask = nullptr; xd.packed = true; xd.scores = nullptr;
        if (rc == UGP_OK) rc = run_place(m, qs, 0, d_out.p, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, false, &xd);
        if (masked) { const int rc2 = mask_words(m, masked, false); if (rc == UGP_OK) rc = rc2; }
        if (rc == UGP_OK && scores_by_levels) {
            rc = run_place(m, qs, 1, nullptr, x.d_scores.p, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
            if (rc == UGP_OK && ugp::launch_scores_mask(x.d_scores.p, q->n_queries, m->flat.n_nodes, x.dev.mask, x.dev.skip, nullptr) != hipSuccess)
                rc = fail(UGP_ERR_HIP, "masking the scores");
        }
    } else
    if (rc == UGP_OK) rc = run_place(m, qs, 0, d_out.p, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, false, &x.dev);
    if (rc == UGP_OK && hipMemcpy(out, d_out.p, q->n_queries * sizeof(ugp_result), hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(UGP_ERR_HIP, "copy results");
    if (rc == UGP_OK && opts->scores) rc = copy_d2h_staged(opts->scores, x.d_scores.p, (size_t)q->n_queries * m->flat.n_nodes * sizeof(int32_t));
    if (rc == UGP_OK) {   // no candidate was eligible: the reference's callers would be left with their initial values
        for (uint64_t i = 0; i < q->n_queries; i++)
            if (out[i].num_best == 0) { out[i].best_set_difference = INT32_MAX; out[i].best_j = UINT32_MAX; out[i].best_has_unique = 0; }
    }
    ugp_qset_destroy(qs);
    return rc;
}

// ---- the node-level options of an extended search, prepared once ---------------------------------------------------------------
struct ugp_ex {
    ugp_mat *m = nullptr;
    uint32_t order = UGP_ORDER_BFS;
    bool has_mask = false, mask_keeps_root = true;
    ExHost x;   // mask, tie rank, index maps on the device
};

int ugp_ex_prepare(ugp_mat *m, const ugp_place_opts *opts, ugp_ex **out) {
    if (!m || !opts || !out) return fail(UGP_ERR_INVALID, "null argument");
    *out = nullptr;
    ugp_ex *e = new (std::nothrow) ugp_ex();
    if (!e) return fail(UGP_ERR_NOMEM, "out of host memory");
    e->m = m; e->order = opts->order;
    e->has_mask = opts->node_mask != nullptr;
    e->mask_keeps_root = !opts->node_mask || opts->node_mask[0] != 0;   // (position 0 is the root in either order)
    if (int rc = prepare_nodes(m, opts, e->x)) { delete e; return rc; }
    *out = e;
    return UGP_OK;
}
void ugp_ex_destroy(ugp_ex *e) { delete e; }

int ugp_place_batch_prepared(ugp_mat *m, const ugp_queries *q, const ugp_ex *e, const uint32_t *skip_node, ugp_result *out, int32_t *d_scores) {
    if (!m || !q || !e || (!out && q->n_queries)) return fail(UGP_ERR_INVALID, "null argument");
    if (e->m != m) return fail(UGP_ERR_INVALID, "the prepared options belong to another handle");
    if (q->n_queries == 0) return UGP_OK;
    const uint64_t N = m->flat.n_nodes, Q = q->n_queries;
    // the handle's own query set and result buffer (no allocation in the steady state)
    if (!m->own_qs) { m->own_qs = new (std::nothrow) ugp_qset(); if (!m->own_qs) return fail(UGP_ERR_NOMEM, "out of host memory"); }
    ugp_qset *qs = m->own_qs;
    if (int rc = qset_fill(m, q, qs)) return rc;
    HIP_TRY(m->d_own_out.reserve(Q));
    ExHost y;   // the per-sample part: excluded nodes
    if (int rc = prepare_samples(m, Q, e->order, skip_node, y)) return rc;
    ExDev xd = e->x.dev;
    xd.skip = y.dev.skip; xd.skip_chunk = y.dev.skip_chunk; xd.scores = nullptr;
    const bool by_levels = d_scores && e->order == UGP_ORDER_BFS && !m->h_level_off.empty() && !m->knobs.scores_dfs && !m->knobs.ex_slow && !m->knobs.force_v1 &&
                           !m->upd.n_excluded;
    const bool packs = !m->knobs.ex_slow && (!d_scores || by_levels) &&
                       (!e->has_mask || (e->mask_keeps_root && m->upd.rec.size() == N && !m->upd.n_excluded && N < (1ull << 30) && (!m->coarse || m->d_coarse2bfs.p)));
    int rc = UGP_OK;
    if (packs) {
        rc = drain(m);
        const uint8_t *masked = xd.mask;
        if (rc == UGP_OK && masked) rc = mask_words(m, masked, true);
        ExDev xp = xd;
        xp.mask = nullptr; xp.packed = true;
        if (rc == UGP_OK) rc = run_place(m, qs, 0, m->d_own_out.p, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, false, &xp);
        if (masked) { const int rc2 = mask_words(m, masked, false); if (rc == UGP_OK) rc = rc2; }
        if (rc == UGP_OK && d_scores) {   // the score matrix straight into the caller's device buffer, level by level; what was not scored reads 0
            rc = run_place(m, qs, 1, nullptr, d_scores, nullptr, nullptr, nullptr, nullptr, 0, nullptr);
            if (rc == UGP_OK && ugp::launch_scores_mask(d_scores, Q, N, xd.mask, xd.skip, nullptr) != hipSuccess) rc = fail(UGP_ERR_HIP, "masking the scores");
        }
    } else {
        if (d_scores) { HIP_TRY(hipMemsetAsync(d_scores, 0, (size_t)Q * N * sizeof(int32_t), nullptr)); xd.scores = d_scores; }
        rc = run_place(m, qs, 0, m->d_own_out.p, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, false, &xd);
    }
    if (rc == UGP_OK && hipMemcpy(out, m->d_own_out.p, Q * sizeof(ugp_result), hipMemcpyDeviceToHost) != hipSuccess) rc = fail(UGP_ERR_HIP, "copy results");
    if (rc == UGP_OK)
        for (uint64_t i = 0; i < Q; i++)
            if (out[i].num_best == 0) { out[i].best_set_difference = INT32_MAX; out[i].best_j = UINT32_MAX; out[i].best_has_unique = 0; }
    return rc;
}

int ugp_tied_nodes_ex(ugp_mat *m, const ugp_queries *q, const ugp_place_opts *opts, uint32_t cap, uint32_t *tie_j, uint8_t *tie_has_unique,
                      uint32_t *tie_count) {
    if (!m || !q || !opts || !tie_count || (cap && (!tie_j || !tie_has_unique))) return fail(UGP_ERR_INVALID, "null argument");
    const uint64_t Q = q->n_queries;
    if (Q == 0) return UGP_OK;
    ugp_qset *qs = nullptr;
    if (int rc = ugp_qset_upload(m, q, &qs)) return rc;
    ExHost x;
    DevBuf<ugp_result> d_res;
    DevBuf<int32_t> d_best;
    DevBuf<uint32_t> d_cnt, d_j;
    DevBuf<uint8_t> d_hu;
    const uint64_t padded = ((Q + 63) / 64) * 64;
    int rc = prepare_ex(m, q, opts, x, false);
    auto chk = [&](hipError_t e, const char *what) {
        if (rc == UGP_OK && e != hipSuccess) rc = fail(UGP_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    chk(d_res.reserve(Q), "hipMalloc"); chk(d_best.reserve(Q), "hipMalloc"); chk(d_cnt.reserve(padded), "hipMalloc");
    chk(d_j.reserve((size_t)Q * std::max<uint32_t>(cap, 1)), "hipMalloc"); chk(d_hu.reserve((size_t)Q * std::max<uint32_t>(cap, 1)), "hipMalloc");
    bool listed = false;
    if (rc == UGP_OK && ex_packs(m, opts) && (!m->coarse || m->d_coarse2bfs.p || !opts->node_mask)) {
        // the packed path: phase 2 writes the lists itself, from the chunks that attain the minimum
        rc = drain(m);
        ExDev xd = x.dev;
        if (rc == UGP_OK && xd.mask) rc = mask_words(m, xd.mask, true);
        const uint8_t *masked = xd.mask;
        xd.mask = nullptr; xd.packed = true;
        if (rc == UGP_OK) chk(hipMemsetAsync(d_cnt.p, 0, padded * sizeof(uint32_t), nullptr), "memset");
        m->tie_lists_filled = m->tie_sub_batches = 0;
        if (rc == UGP_OK) rc = run_place(m, qs, 0, d_res.p, nullptr, nullptr, d_cnt.p, d_j.p, d_hu.p, cap, nullptr, false, &xd);
        listed = rc == UGP_OK && m->tie_sub_batches > 0 && m->tie_lists_filled == m->tie_sub_batches;
        if (masked) { const int rc2 = mask_words(m, masked, false); if (rc == UGP_OK) rc = rc2; }
    }
    if (rc == UGP_OK && !listed) rc = run_place(m, qs, 0, d_res.p, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, false, &x.dev);
    if (rc == UGP_OK && !listed) {
        chk(ugp::launch_extract_best(d_res.p, (uint32_t)Q, d_best.p, nullptr), "extract best");
        chk(hipMemsetAsync(d_cnt.p, 0, padded * sizeof(uint32_t), nullptr), "memset");
    }
    if (rc == UGP_OK && !listed) rc = run_place(m, qs, 2, nullptr, nullptr, d_best.p, d_cnt.p, d_j.p, d_hu.p, cap, nullptr, false, &x.dev);
    if (rc == UGP_OK) {
        chk(hipMemcpy(tie_count, d_cnt.p, Q * sizeof(uint32_t), hipMemcpyDeviceToHost), "copy tie counts");
        if (cap) {
            chk(hipMemcpy(tie_j, d_j.p, (size_t)Q * cap * sizeof(uint32_t), hipMemcpyDeviceToHost), "copy ties");
            chk(hipMemcpy(tie_has_unique, d_hu.p, (size_t)Q * cap, hipMemcpyDeviceToHost), "copy ties");
        }
    }
    if (rc == UGP_OK && cap) {   // ascending index of the caller's node vector
        std::vector<std::pair<uint32_t, uint8_t>> tmp;
        for (uint64_t i = 0; i < Q; i++) {
            const uint32_t k = std::min<uint32_t>(tie_count[i], cap);
            tmp.resize(k);
            for (uint32_t t = 0; t < k; t++) tmp[t] = {tie_j[i * cap + t], tie_has_unique[i * cap + t]};
            std::sort(tmp.begin(), tmp.end());
            for (uint32_t t = 0; t < k; t++) { tie_j[i * cap + t] = tmp[t].first; tie_has_unique[i * cap + t] = tmp[t].second; }
        }
    }
    ugp_qset_destroy(qs);
    return rc;
}

// ---- add mode ---------------------------------------------------------------------------------------------------------------

namespace {
constexpr uint32_t kTouchedCap = 64;   // list entries kept per sample on the device (the true count is kept beside them)

// calls of ugp_place_device_overlapped / _async still in flight on the handle's own streams: the entry points below run on the
// default stream and change what those calls read
int drain(ugp_mat *m) {
    HIP_TRY(hipSetDevice(m->device));
    for (auto &W : m->work) if (W.done) HIP_TRY(hipEventSynchronize(W.done));
    if (m->coarse) for (auto &W : m->coarse->work) if (W.done) HIP_TRY(hipEventSynchronize(W.done));
    return UGP_OK;
}

int or_words(ugp_mat *m, uint32_t *stream, const std::vector<uint32_t> &pos, uint32_t bits) {
    if (pos.empty()) return UGP_OK;
    HIP_TRY(m->upd.d_tmp.upload(pos));
    HIP_TRY(ugp::launch_or_words(stream, m->upd.d_tmp.p, (uint32_t)pos.size(), bits, nullptr));
    HIP_TRY(hipStreamSynchronize(nullptr));   // (d_tmp is reused by the next list)
    return UGP_OK;
}

ugp::TouchedArgs touched_args(ugp_mat *m, uint32_t id0, uint32_t id1, uint32_t q0, uint32_t q1) {
    auto &U = m->upd;
    ugp::TouchedArgs a{};
    a.rec = U.d_rec.p; a.ent = U.d_ent.p; a.alive = U.d_alive.p; a.id0 = id0; a.id1 = id1;
    a.dense = U.d_dense.p; a.n_pos = U.n_pos; a.qpad = U.qpad; a.dbot = U.d_dbot.p; a.q0 = q0; a.q1 = q1;
    a.best = U.d_best.p; a.cnt = U.d_cnt.p; a.ids = U.d_ids.p; a.hu = U.d_hu.p; a.cap = kTouchedCap;
    return a;
}
}  // namespace

int ugp_mat_update(ugp_mat *m, const ugp_touched *recs, const uint32_t *retired, uint64_t n_retired, uint32_t *first_id) {
    if (!m || (n_retired && !retired)) return fail(UGP_ERR_INVALID, "null argument");
    auto &U = m->upd;
    const uint64_t N = m->flat.n_nodes, n_new = recs ? recs->n : 0;
    if (first_id) *first_id = (uint32_t)U.n_rec;
    if (n_new && (!recs->flat_j || !recs->flags || !recs->n_path || !recs->ent_off)) return fail(UGP_ERR_INVALID, "null record arrays");
    if (N >= (1ull << 30)) return fail(UGP_ERR_UNSUPPORTED, "ugp_mat_update needs a tree of fewer than 2^30 nodes");
    if (U.rec.size() != N) return fail(UGP_ERR_UNSUPPORTED, "this handle was flattened without the update maps (UGP_NO_UPDATE_MAPS)");
    if (U.n_rec + n_new >= (1ull << 32)) return fail(UGP_ERR_UNSUPPORTED, "more than 2^32 records");
    if (int rc = drain(m)) return rc;
    try {
        // retired records
        // (everything is validated before anything is committed: a call that fails leaves the handle as it was)
        for (uint64_t i = 0; i < n_retired; i++) if (retired[i] >= U.n_rec) return fail(UGP_ERR_INVALID, "retired record id out of range");
        auto retire = [&]() -> int {   // (alive is a byte per record)
            for (uint64_t i = 0; i < n_retired; i++) HIP_TRY(hipMemsetAsync(U.d_alive.p + retired[i], 0, 1, nullptr));
            return UGP_OK;
        };
        if (!n_new) { if (int rc = retire()) return rc; HIP_TRY(hipStreamSynchronize(nullptr)); return UGP_OK; }
        const uint64_t n_ent = recs->ent_off[n_new];
        if (n_ent && (!recs->pos || !recs->allele || !recs->prev || !recs->ref)) return fail(UGP_ERR_INVALID, "null entry arrays");
        std::vector<ugp::TouchedRec> hr(n_new);
        std::vector<ugp::TouchedEnt> he(n_ent);
        std::vector<uint32_t> p8, pr, pt, c8, cr, ct, new_excl;
        auto one_hot = [](uint8_t a) { return a == 1 || a == 2 || a == 4 || a == 8; };
        for (uint64_t i = 0; i < n_new; i++) {
            const uint64_t b = recs->ent_off[i], e = recs->ent_off[i + 1];
            if (e < b || e > n_ent || recs->n_path[i] > e - b) return fail(UGP_ERR_INVALID, "record offsets are not monotone");
            if (U.n_ent + b >= (1ull << 32)) return fail(UGP_ERR_UNSUPPORTED, "more than 2^32 record entries");
            hr[i] = {(uint32_t)(U.n_ent + b), recs->n_path[i], (uint32_t)(e - b) - recs->n_path[i], (uint32_t)recs->flags[i]};
            for (uint64_t k = b; k < e; k++) {
                const bool own = k - b >= recs->n_path[i];
                if (!one_hot(recs->allele[k]) || !one_hot(recs->ref[k]) || (own && !one_hot(recs->prev[k]))) return fail(UGP_ERR_UNSUPPORTED, "record alleles must be single bases");
                he[k] = {recs->pos[k], (uint32_t)recs->allele[k] | ((uint32_t)(own ? recs->prev[k] : 0) << 8) | ((uint32_t)recs->ref[k] << 16)};
            }
            const uint32_t j = recs->flat_j[i];
            if (j == UINT32_MAX) continue;
            if (j == 0 || j >= N) return fail(j == 0 ? UGP_ERR_UNSUPPORTED : UGP_ERR_INVALID, j == 0 ? "the root cannot be taken out of the candidate set" : "flat_j out of range");
            if (U.hdr8[j] != UINT32_MAX) p8.push_back(U.hdr8[j]);
            pr.push_back(U.rec[j] + 1u);
            if (U.post[j] != UINT32_MAX) pt.push_back(U.post[j] + 1u);
            if (m->coarse) {
                auto it = std::lower_bound(U.coarse2bfs.begin(), U.coarse2bfs.end(), j);
                if (it != U.coarse2bfs.end() && *it == j) {
                    const size_t k = (size_t)(it - U.coarse2bfs.begin());
                    auto &C = m->coarse->upd;
                    if (C.rec.size() == m->coarse->flat.n_nodes && k != 0) {
                        if (C.hdr8[k] != UINT32_MAX) c8.push_back(C.hdr8[k]);
                        cr.push_back(C.rec[k] + 1u);
                        if (C.post[k] != UINT32_MAX) ct.push_back(C.post[k] + 1u);
                    }
                }
            }
            new_excl.push_back(j);
        }
        if (int rc = retire()) return rc;
        U.n_excluded += new_excl.size();
        U.excluded_j.insert(U.excluded_j.end(), new_excl.begin(), new_excl.end());
        HIP_TRY(U.d_rec.grow_keep(U.n_rec + n_new, U.n_rec));
        HIP_TRY(U.d_alive.grow_keep(U.n_rec + n_new, U.n_rec));
        HIP_TRY(U.d_ent.grow_keep(U.n_ent + n_ent, U.n_ent));
        // (round 6) Everything the call uploads -- the records, their entries, the six lists of word positions -- goes through ONE pinned
        // staging buffer with asynchronous copies and ONE wait at the end (the driver calls this once per round of 64 insertions:
        // eight blocking copies and six waits each were 0.19 ms, 0.3 s per 100 000 insertions).
        const std::vector<uint32_t> *lists[6] = {&p8, &pr, &pt, &c8, &cr, &ct};
        size_t n_words = 0;
        for (auto *l : lists) n_words += l->size();
        const size_t o_rec = 0, o_ent = o_rec + n_new * sizeof(ugp::TouchedRec), o_lists = (o_ent + n_ent * sizeof(ugp::TouchedEnt) + 15) & ~(size_t)15;
        if (int rc = U.stage_up.reserve(o_lists + n_words * 4 + 64)) return rc;
        char *st = (char *)U.stage_up.p;
        memcpy(st + o_rec, hr.data(), n_new * sizeof(ugp::TouchedRec));
        if (n_ent) memcpy(st + o_ent, he.data(), n_ent * sizeof(ugp::TouchedEnt));
        { size_t o = o_lists; for (auto *l : lists) { if (!l->empty()) memcpy(st + o, l->data(), l->size() * 4); o += l->size() * 4; } }
        HIP_TRY(hipMemcpyAsync(U.d_rec.p + U.n_rec, st + o_rec, n_new * sizeof(ugp::TouchedRec), hipMemcpyHostToDevice, nullptr));
        if (n_ent) HIP_TRY(hipMemcpyAsync(U.d_ent.p + U.n_ent, st + o_ent, n_ent * sizeof(ugp::TouchedEnt), hipMemcpyHostToDevice, nullptr));
        HIP_TRY(hipMemsetAsync(U.d_alive.p + U.n_rec, 1, n_new, nullptr));
        U.n_rec += n_new; U.n_ent += n_ent;
        // the flattened nodes among them leave the candidate set: one bit in their words of the packed stream (H_NOSCORE), of the
        // 32-bit stream and of the tie stream (bit 31 of the key word), here and in the coarse tree of the locality pre-pass
        if (n_words) {
            HIP_TRY(U.d_tmp.reserve(n_words));
            HIP_TRY(hipMemcpyAsync(U.d_tmp.p, st + o_lists, n_words * 4, hipMemcpyHostToDevice, nullptr));
            uint32_t *streams[6] = {m->d_stream8.p, m->d_stream.p, m->d_stream_t.p, m->coarse ? m->coarse->d_stream8.p : nullptr, m->coarse ? m->coarse->d_stream.p : nullptr,
                                    m->coarse ? m->coarse->d_stream_t.p : nullptr};
            const uint32_t bits[6] = {ugp::H_NOSCORE, ugp::KEY_EXCLUDED, ugp::KEY_EXCLUDED, ugp::H_NOSCORE, ugp::KEY_EXCLUDED, ugp::KEY_EXCLUDED};
            size_t o = 0;
            for (int i = 0; i < 6; i++) {
                if (!lists[i]->empty() && streams[i]) HIP_TRY(ugp::launch_or_words(streams[i], U.d_tmp.p + o, (uint32_t)lists[i]->size(), bits[i], nullptr));
                o += lists[i]->size();
            }
        }
        HIP_TRY(hipStreamSynchronize(nullptr));   // (the staging buffer is the next call's; and the caller's next search sees the edits)
    } catch (const std::bad_alloc &) { return fail(UGP_ERR_NOMEM, "out of host memory"); }
    return UGP_OK;
}

int ugp_touched_open(ugp_mat *m, const ugp_queries *q) {
    if (!m || !q) return fail(UGP_ERR_INVALID, "null argument");
    auto &U = m->upd;
    U.open = false;
    if (int rc = drain(m)) return rc;
    if (!U.qs) { U.qs = new (std::nothrow) ugp_qset(); if (!U.qs) return fail(UGP_ERR_NOMEM, "out of host memory"); }
    if (int rc = qset_fill(m, q, U.qs)) return rc;
    const uint64_t Q = q->n_queries, n_ent = U.qs->n_ent;
    if (Q >= (1ull << 31)) return fail(UGP_ERR_UNSUPPORTED, "batch too large");
    int32_t max_pos = (int32_t)m->flat.max_pos;
    for (uint64_t e = 0; e < n_ent; e++) max_pos = std::max(max_pos, q->pos[e]);
    U.n_pos = (uint32_t)max_pos + 1u;
    U.qpad = (uint32_t)((Q + 63) / 64 * 64);
    U.Q = Q;
    if (!Q) { U.open = true; return UGP_OK; }
    HIP_TRY(U.d_dense.reserve((size_t)U.n_pos * U.qpad));
    HIP_TRY(U.d_dbot.reserve(U.qpad)); HIP_TRY(U.d_best.reserve(U.qpad)); HIP_TRY(U.d_list_best.reserve(U.qpad)); HIP_TRY(U.d_cnt.reserve(U.qpad));
    HIP_TRY(U.d_ids.reserve((size_t)U.qpad * kTouchedCap)); HIP_TRY(U.d_hu.reserve((size_t)U.qpad * kTouchedCap));
    HIP_TRY(hipMemsetAsync(U.d_dense.p, 0, (size_t)U.n_pos * U.qpad, nullptr));
    HIP_TRY(hipMemsetAsync(U.d_dbot.p, 0, (size_t)U.qpad * 4, nullptr));
    HIP_TRY(hipMemsetAsync(U.d_cnt.p, 0, (size_t)U.qpad * 4, nullptr));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)U.d_best.p, INT32_MAX, U.qpad, nullptr));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)U.d_list_best.p, INT32_MAX, U.qpad, nullptr));
    if (n_ent)
        HIP_TRY(ugp::launch_dense_scatter(U.d_dense.p, U.n_pos, U.qpad, U.d_dbot.p, U.qs->d_pos.p, U.qs->d_ref.p, U.qs->d_nuc.p, U.qs->d_missing.p, U.qs->d_ent_q.p,
                                          n_ent, nullptr));
    U.open = true;
    if (U.n_rec) HIP_TRY(ugp::launch_touched(touched_args(m, 0, (uint32_t)U.n_rec, 0, (uint32_t)Q), U.d_list_best.p, nullptr));
    return UGP_OK;
}

int ugp_touched_score(ugp_mat *m, uint32_t first_id, uint64_t first_sample) {
    if (!m) return fail(UGP_ERR_INVALID, "null argument");
    auto &U = m->upd;
    if (!U.open) return fail(UGP_ERR_INVALID, "no batch is open (ugp_touched_open)");
    if (first_id > U.n_rec || first_sample > U.Q) return fail(UGP_ERR_INVALID, "record id / sample out of range");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(ugp::launch_touched(touched_args(m, first_id, (uint32_t)U.n_rec, (uint32_t)first_sample, (uint32_t)U.Q), U.d_list_best.p, nullptr));
    return UGP_OK;
}

int ugp_touched_rescore(ugp_mat *m, uint64_t sample) {
    if (!m) return fail(UGP_ERR_INVALID, "null argument");
    auto &U = m->upd;
    if (!U.open || sample >= U.Q) return fail(UGP_ERR_INVALID, "no batch is open / sample out of range");
    HIP_TRY(hipSetDevice(m->device));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(U.d_best.p + sample), INT32_MAX, 1, nullptr));
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)(U.d_list_best.p + sample), INT32_MAX, 1, nullptr));
    HIP_TRY(hipMemsetAsync(U.d_cnt.p + sample, 0, 4, nullptr));
    if (U.n_rec) HIP_TRY(ugp::launch_touched(touched_args(m, 0, (uint32_t)U.n_rec, (uint32_t)sample, (uint32_t)sample + 1u), U.d_list_best.p, nullptr));
    return UGP_OK;
}

int ugp_touched_fetch(ugp_mat *m, uint64_t first_sample, uint64_t n, uint32_t cap, int32_t *best, uint32_t *count, uint32_t *ids, uint8_t *has_unique) {
    if (!m || (n && (!best || !count || (cap && (!ids || !has_unique))))) return fail(UGP_ERR_INVALID, "null argument");
    auto &U = m->upd;
    if (!U.open || first_sample + n > U.Q) return fail(UGP_ERR_INVALID, "no batch is open / samples out of range");
    if (!n) return UGP_OK;
    HIP_TRY(hipSetDevice(m->device));
    // (round 6) One wait instead of four blocking copies into pageable memory: the driver fetches once per round of 64 samples, 1 562
    // times for 100 000 insertions -- 0.46 ms each, 0.7 s of the run, nearly all of it the copies' fixed cost.
    const uint32_t k = cap ? std::min(cap, kTouchedCap) : 0u;
    const size_t o_best = 0, o_cnt = o_best + n * 4, o_ids = o_cnt + n * 4, o_hu = o_ids + n * (size_t)k * 4, total = o_hu + n * (size_t)k;
    if (int rc = U.stage.reserve(total + 64)) return rc;
    char *st = (char *)U.stage.p;
    HIP_TRY(hipMemcpyAsync(st + o_best, U.d_best.p + first_sample, n * 4, hipMemcpyDeviceToHost, nullptr));
    HIP_TRY(hipMemcpyAsync(st + o_cnt, U.d_cnt.p + first_sample, n * 4, hipMemcpyDeviceToHost, nullptr));
    if (k) {
        if (k == kTouchedCap) {
            HIP_TRY(hipMemcpyAsync(st + o_ids, U.d_ids.p + first_sample * kTouchedCap, n * (size_t)k * 4, hipMemcpyDeviceToHost, nullptr));
            HIP_TRY(hipMemcpyAsync(st + o_hu, U.d_hu.p + first_sample * kTouchedCap, n * (size_t)k, hipMemcpyDeviceToHost, nullptr));
        } else {
            HIP_TRY(hipMemcpy2DAsync(st + o_ids, (size_t)k * 4, U.d_ids.p + first_sample * kTouchedCap, (size_t)kTouchedCap * 4, (size_t)k * 4, n, hipMemcpyDeviceToHost, nullptr));
            HIP_TRY(hipMemcpy2DAsync(st + o_hu, k, U.d_hu.p + first_sample * kTouchedCap, kTouchedCap, k, n, hipMemcpyDeviceToHost, nullptr));
        }
    }
    HIP_TRY(hipStreamSynchronize(nullptr));
    memcpy(best, st + o_best, n * 4);
    memcpy(count, st + o_cnt, n * 4);
    if (k) {
        if (k == cap) { memcpy(ids, st + o_ids, n * (size_t)k * 4); memcpy(has_unique, st + o_hu, n * (size_t)k); }
        else for (uint64_t i = 0; i < n; i++) {   // (a caller's rows wider than the device keeps: the first k of each)
            memcpy(ids + i * cap, st + o_ids + i * (size_t)k * 4, (size_t)k * 4);
            memcpy(has_unique + i * cap, st + o_hu + i * (size_t)k, k);
        }
    }
    return UGP_OK;
}

// third pruning bound: the sub-batches that have completed by now, in the order they were issued, go to the handle's tuner (no
// waiting: calls on a handle come from one thread)
static void tuner_poll(ugp_mat *m) {
    B3Tuner &T = m->b3_tuner;
    for (;;) {
        EventSet *hit = nullptr;
        for (auto &W : m->work)
            for (auto &G : W.gens)
                if (G.timing_pending)
                    for (size_t i = 0; i < G.events_used; i++)
                        if (G.events[i].b3_class >= 0 && G.events[i].b3_seq == T.next_seq) hit = &G.events[i];
        if (!hit && T.seq > T.next_seq + 16) { T.resync(T.next_seq + 1); continue; }   // (a call that failed half way left a hole)
        if (!hit || hipEventQuery(hit->ev[3]) != hipSuccess) break;
        if (hit->b3_pos == 0) { T.acc = 0; T.acc_n = 0; }
        float gap = 0;
        if (T.prev_done && hit->b3_pos >= B3Tuner::kSkip && hit->b3_tiles && hipEventElapsedTime(&gap, (hipEvent_t)T.prev_done, hit->ev[3]) == hipSuccess && gap > 0) {
            T.acc += (double)gap / hit->b3_tiles; T.acc_n++;
        }
        if (hit->b3_pos == B3Tuner::kBlock - 1 && T.acc_n && !hit->b3_first) T.record(hit->b3_class, hit->b3_used, T.acc / T.acc_n);
        T.prev_done = (void *)hit->ev[3];
        T.next_seq++;
        hit->b3_class = -1;
    }
    (void)hipGetLastError();   // (hipErrorNotReady of a query is not an error of the call)
}
// ... before a ring entry's events are recorded again: whatever of it the tuner has not seen is dropped
static void tuner_release(ugp_mat *m, ugp_mat::Work::Gen &G) {
    tuner_poll(m);
    B3Tuner &T = m->b3_tuner;
    for (size_t i = 0; i < G.events_used; i++) {
        EventSet &es = G.events[i];
        if (es.b3_class >= 0) { T.resync(es.b3_seq + 1); es.b3_class = -1; }
        if (T.prev_done == (void *)es.ev[3]) T.prev_done = nullptr;
    }
}

// Durations of the set's last call from its HIP events (waits for that call), added to the handle's running totals.
static int harvest_timing(ugp_mat *m, ugp_mat::Work &W, ugp_mat::Work::Gen &G) {
    if (G.timing_pending) {
        float table = 0, place = 0, merge = 0;
        for (size_t i = 0; i < G.events_used; i++) {
            EventSet &es = G.events[i];
            HIP_TRY(hipEventSynchronize(es.ev[3]));
            float t;
            HIP_TRY(hipEventElapsedTime(&t, es.ev[0], es.ev[1])); table += t;
            HIP_TRY(hipEventElapsedTime(&t, es.ev[1], es.ev[2])); place += t;
            HIP_TRY(hipEventElapsedTime(&t, es.ev[2], es.ev[3])); merge += t;
        }
        G.last.table_ms = table; G.last.place_ms = place; G.last.merge_ms = merge;
        G.last.coarse_ms = 0;
        if (G.coarse_timed) { HIP_TRY(hipEventSynchronize(G.ev_coarse[1])); HIP_TRY(hipEventElapsedTime(&G.last.coarse_ms, G.ev_coarse[0], G.ev_coarse[1])); }
        G.last.words_total = W.last_words_total;
        G.last.words_skipped = 0;
        if (W.last_used_best8 && W.d_stats.p && m->knobs.stats) {   // (debug counters: a blocking copy)
            uint64_t v[96] = {0};
            HIP_TRY(hipMemcpy(v, W.d_stats.p, sizeof v, hipMemcpyDeviceToHost));
            G.last.words_skipped = v[0];
            G.last.reserved = (uint32_t)std::min<uint64_t>(v[1], 0xFFFFFFFFull);   // pipeline (re)starts
            if (const char *tf = m->knobs.trace.empty() ? nullptr : m->knobs.trace.c_str()) {
                if (W.d_trace.p) {
                    uint64_t n = 0;
                    HIP_TRY(hipMemcpy(&n, W.d_trace.p, 8, hipMemcpyDeviceToHost));
                    n = std::min<uint64_t>(n, 1u << 20);
                    std::vector<uint64_t> rec(n * 6);
                    if (n) HIP_TRY(hipMemcpy(rec.data(), W.d_trace.p + 8, n * 48, hipMemcpyDeviceToHost));
                    if (FILE *fp = fopen(tf, "wb")) { fwrite(rec.data(), 8, rec.size(), fp); fclose(fp); }
                    fprintf(stderr, "[ugp stats] %llu unit records written to %s\n", (unsigned long long)n, tf);
                }
            }
            {
                fprintf(stderr, "[ugp stats] restarts=%llu restart_cycles=%llu wave_cycles=%llu max_wave=%llu hist:", (unsigned long long)v[1],
                        (unsigned long long)v[2], (unsigned long long)v[3], (unsigned long long)v[4]);
                for (int i = 0; i < 16; i++) fprintf(stderr, " %llu", (unsigned long long)v[5 + i]);
                fprintf(stderr, "\n[ugp stats] jumps decided by the first node after a restart=%llu", (unsigned long long)v[26]);
                fprintf(stderr, "\n[ugp stats] own-region units=%llu cycles=%llu   other units=%llu cycles=%llu\n", (unsigned long long)v[29],
                        (unsigned long long)v[27], (unsigned long long)v[30], (unsigned long long)v[28]);
                fprintf(stderr, "[ugp stats] other units by what the preamble records decided: whole unit skipped %llu (cycles %llu), body entered late %llu (cycles %llu), "
                                "nothing %llu (cycles %llu)\n[ugp stats] units skipped whole: cycles pulling %llu, replaying the preamble %llu, closing %llu\n", (unsigned long long)v[32], (unsigned long long)v[33],
                        (unsigned long long)v[34], (unsigned long long)v[35], (unsigned long long)v[36], (unsigned long long)v[37], (unsigned long long)v[38],
                        (unsigned long long)v[39], (unsigned long long)v[40]);
                fprintf(stderr, "[ugp stats] restarts by cause (own-region / other units): jump %llu / %llu, sibling jump %llu / %llu, chunk end %llu / %llu, slow header %llu / %llu\n",
                        (unsigned long long)v[48], (unsigned long long)v[49], (unsigned long long)v[50], (unsigned long long)v[51], (unsigned long long)v[52],
                        (unsigned long long)v[53], (unsigned long long)v[54], (unsigned long long)v[55]);
                fprintf(stderr, "[ugp stats] units split while running: %llu\n", (unsigned long long)v[31]);
                fprintf(stderr, "[ugp stats] nodes evaluated: %llu in unit bodies = %.3f %% of nodes x tiles (%llu x %u), %llu in preamble replays\n", (unsigned long long)v[66],
                        G.last.n_tiles ? 100.0 * (double)v[66] / ((double)m->flat.n_nodes * G.last.n_tiles) : 0.0, (unsigned long long)m->flat.n_nodes, G.last.n_tiles,
                        (unsigned long long)v[67]);
                fprintf(stderr, "[ugp stats] third bound: asked at a restart %llu times, decided the jump %llu times\n", (unsigned long long)v[64], (unsigned long long)v[65]);
                fprintf(stderr, "[ugp stats] third bound by the record's hsub - hsec (1 2 3 more), decided/asked: %llu/%llu %llu/%llu %llu/%llu %llu/%llu\n", (unsigned long long)v[92],
                        (unsigned long long)v[88], (unsigned long long)v[93], (unsigned long long)v[89], (unsigned long long)v[94], (unsigned long long)v[90], (unsigned long long)v[95],
                        (unsigned long long)v[91]);
                fprintf(stderr, "[ugp stats] third bound by jump length (<16 <32 <64 <128 <256 <512 <1024 more), decided/asked:");
                for (int i = 0; i < 8; i++) fprintf(stderr, " %llu/%llu", (unsigned long long)v[80 + i], (unsigned long long)v[72 + i]);
                fprintf(stderr, "\n");
                fprintf(stderr, "[ugp stats] jump lengths in words (<8 <16 <32 <64 <128 <512 <4096 more):");
                for (int i = 0; i < 8; i++) fprintf(stderr, " %llu", (unsigned long long)v[56 + i]);
                fprintf(stderr, "\n");
                if (W.last_nitems) {
                    uint32_t ni = 0;
                    HIP_TRY(hipMemcpy(&ni, W.last_nitems, 4, hipMemcpyDeviceToHost));
                    fprintf(stderr, "[ugp stats] phase 2: %u (chunk, 64-sample sub-tile) pairs walked by k_ties\n", ni);
                }
                if (W.last_list_n && W.last_list_tiles) {
                    std::vector<uint32_t> ln(W.last_list_tiles);
                    HIP_TRY(hipMemcpy(ln.data(), W.last_list_n, ln.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
                    uint64_t rec = 0;
                    for (uint32_t x : ln) rec += x;
                    fprintf(stderr, "[ugp stats] chunk records stored by phase 1: %llu of %llu (chunk, tile) pairs = %.1f MB\n", (unsigned long long)rec,
                            (unsigned long long)m->flat.n_chunks * W.last_list_tiles, rec * 1024.0 / 1e6);
                }
            }
        }
        m->tsum.table_ms += G.last.table_ms; m->tsum.place_ms += G.last.place_ms; m->tsum.merge_ms += G.last.merge_ms;
        m->tsum.coarse_ms += G.last.coarse_ms; m->tsum.place_launches += G.last.place_launches; m->tsum.bound3 += G.last.bound3;
        m->tsum_calls++;
        tuner_release(m, G);
        G.timing_pending = false;
    }
    return UGP_OK;
}

int ugp_get_timing(ugp_mat *m, ugp_timing *out) {
    if (!m || !out) return fail(UGP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    ugp_mat::Work &W = m->work[m->last_work];
    if (int rc = harvest_timing(m, W, W.gens[W.cur])) return rc;
    *out = W.gens[W.cur].last;
    return UGP_OK;
}

int ugp_debug_bound3_tables(ugp_mat *m, uint32_t tile, uint16_t *over, uint16_t *under, uint64_t cap, uint64_t *n_blocks) {
    if (!m || !n_blocks) return fail(UGP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    ugp_mat::Work &W = m->work[m->last_work];
    *n_blocks = W.b3_host.n_blocks;
    if (!W.b3_host.over) { *n_blocks = 0; return UGP_OK; }   // (the last call did not build them)
    if (!over || !under) return UGP_OK;
    if (cap < W.b3_host.n_blocks) return fail(UGP_ERR_INVALID, "buffer too small");
    HIP_TRY(hipDeviceSynchronize());
    // (the device keeps one byte per block since round 6, 255 = "255 or more"; the hook's arrays stayed 16 bits wide)
    const size_t nb = W.b3_host.n_blocks;
    std::vector<uint8_t> tmp(nb);
    HIP_TRY(hipMemcpy(tmp.data(), W.b3_host.over + (uint64_t)tile * nb, nb, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nb; i++) over[i] = tmp[i];
    HIP_TRY(hipMemcpy(tmp.data(), W.b3_host.under + (uint64_t)tile * nb, nb, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nb; i++) under[i] = tmp[i];
    return UGP_OK;
}

int ugp_get_timing_sum(ugp_mat *m, ugp_timing *sum, uint32_t *n_calls) {
    if (!m || !sum || !n_calls) return fail(UGP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(m->device));
    for (auto &W : m->work)
        for (auto &G : W.gens)
            if (int rc = harvest_timing(m, W, G)) return rc;
    const ugp_timing &l = m->work[m->last_work].gens[m->work[m->last_work].cur].last;   // the sizes of the last call; the durations summed
    *sum = l;
    sum->table_ms = m->tsum.table_ms; sum->place_ms = m->tsum.place_ms; sum->merge_ms = m->tsum.merge_ms; sum->coarse_ms = m->tsum.coarse_ms;
    sum->place_launches = m->tsum.place_launches;
    sum->bound3 = m->tsum.bound3;   // (calls that used the third bound)
    *n_calls = m->tsum_calls;
    m->tsum = {};
    m->tsum_calls = 0;
    return UGP_OK;
}

// Test / tuning hook: read the handle's per-call tuning switches from the environment again (tools that sweep knobs on one
// handle; the switches that shape the flattening -- chunk size, pruning records, LDS slots -- need a new handle).
int ugp_mat_reload_knobs(ugp_mat *m) {
    if (!m) return fail(UGP_ERR_INVALID, "null argument");
    m->knobs = ugp::Knobs::from_env();
    if (m->coarse) m->coarse->knobs = m->knobs;
    return UGP_OK;
}

// 1 when this library was built with the experiments and diagnostics (UGP_STATS, UGP_TRACE, UGP_SEED_PREV / _CHECK,
// UGP_PHASE2_PACKED, UGP_KBEST_EXCLUSIVE), 0 for the release build, which ignores those variables.
int ugp_has_experiments(void) {
#ifdef UGP_EXPERIMENTS
    return 1;
#else
    return 0;
#endif
}

// Test / tuning hook (not part of the drop-in surface): ugp_mat_create with an
// explicit chunk size so small fixtures exercise multi-chunk launches.
int ugp_mat_create_chunked(const ugp_tree_desc *tree, int device, uint32_t chunk_nodes, ugp_mat **out) {
    ugp::Options opt = default_options();
    opt.chunk_nodes = chunk_nodes;
    return mat_create_impl(tree, device, opt, out);
}

// Host-only view of the flattened tree (no device needed) so the CPU test
// suite can check the record stream, slots, preambles and tables.
struct ugp_flat { ugp::FlatMat f; };

int ugp_flat_create(const ugp_tree_desc *tree, uint32_t chunk_nodes, ugp_flat **out) {
    if (!tree || !out) return fail(UGP_ERR_INVALID, "null argument");
    *out = nullptr;
    ugp_flat *fl = new (std::nothrow) ugp_flat();
    if (!fl) return fail(UGP_ERR_NOMEM, "out of host memory");
    ugp::Options opt = default_options();
    opt.chunk_nodes = chunk_nodes;
    std::string err;
    int rc;
    try {
        rc = ugp::flatten(*tree, opt, fl->f, err);
    } catch (const std::bad_alloc &) {
        delete fl;
        return fail(UGP_ERR_NOMEM, "out of host memory while flattening the tree");
    }
    if (rc != UGP_OK) { delete fl; return fail(rc, err); }
    *out = fl;
    return UGP_OK;
}

void ugp_flat_destroy(ugp_flat *fl) { delete fl; }

int ugp_flat_get(const ugp_flat *fl, int which, const void **ptr, uint64_t *count) {
    if (!fl || !ptr || !count) return fail(UGP_ERR_INVALID, "null argument");
    const auto &f = fl->f;
    switch (which) {
        case UGP_FLAT_STREAM: *ptr = f.stream.data(); *count = f.stream.size(); break;
        case UGP_FLAT_PRE_STREAM: *ptr = f.pre_stream.data(); *count = f.pre_stream.size(); break;
        case UGP_FLAT_CHUNK_BODY_OFF: *ptr = f.chunk_body_off.data(); *count = f.chunk_body_off.size(); break;
        case UGP_FLAT_CHUNK_PRE_OFF: *ptr = f.chunk_pre_off.data(); *count = f.chunk_pre_off.size(); break;
        case UGP_FLAT_CHUNK_NODE_OFF: *ptr = f.chunk_node_off.data(); *count = f.chunk_node_off.size(); break;
        case UGP_FLAT_POS2SITE: *ptr = f.pos2site.data(); *count = f.pos2site.size(); break;
        case UGP_FLAT_SITE_REF: *ptr = f.site_ref.data(); *count = f.site_ref.size(); break;
        case UGP_FLAT_RANK2BFS: *ptr = f.rank2bfs.data(); *count = f.rank2bfs.size(); break;
        case UGP_FLAT_DFS2BFS: *ptr = f.dfs2bfs.data(); *count = f.dfs2bfs.size(); break;
        case UGP_FLAT_MAX_SLOTS: *ptr = nullptr; *count = f.max_slots; break;
        case UGP_FLAT_STREAM8: *ptr = f.stream8.data(); *count = f.stream8.size(); break;
        case UGP_FLAT_PRE8_STREAM: *ptr = f.pre8_stream.data(); *count = f.pre8_stream.size(); break;
        case UGP_FLAT_CHUNK8_BODY_OFF: *ptr = f.chunk8_body_off.data(); *count = f.chunk8_body_off.size(); break;
        case UGP_FLAT_CHUNK8_PRE_OFF: *ptr = f.chunk8_pre_off.data(); *count = f.chunk8_pre_off.size(); break;
        case UGP_FLAT_STREAM_T: *ptr = f.stream_t.data(); *count = f.stream_t.size(); break;
        case UGP_FLAT_CHUNK_T_OFF: *ptr = f.chunk_t_off.data(); *count = f.chunk_t_off.size(); break;
        case UGP_FLAT_MAX_PATH_MUTS: *ptr = nullptr; *count = f.max_path_muts; break;
        case UGP_FLAT_LDS_SLOTS: *ptr = nullptr; *count = f.lds_slots; break;
        case UGP_FLAT_B3_GROUP_OFF: *ptr = f.b3_group_off.data(); *count = f.b3_group_off.size(); break;
        case UGP_FLAT_B3_EVENTS: *ptr = f.b3_events.data(); *count = f.b3_events.size(); break;
        default: return fail(UGP_ERR_INVALID, "unknown array id");
    }
    return UGP_OK;
}

}  // extern "C"
