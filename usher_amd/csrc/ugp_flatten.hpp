// ugp_flatten.hpp -- host-side flattening of a mutation-annotated tree into the
// DFS record stream the HIP kernels walk.  Pure C++ (no HIP), so the CPU test
// suite can exercise it without a GPU.
#pragma once
#include <cstdint>
#include <memory>
#include <new>
#include <algorithm>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "usher_amd.h"

namespace ugp {

// ---- record stream encoding (shared with ugp_kernels.hip) -----------------
//
// One record per node, in the library's own DFS preorder (children reordered
// so the largest subtree comes last, which bounds the D stack by log2 N):
//   w0  [15:0]  number of mutation words that follow
//       [21:16] rslot: slot to read D(parent) from; RS_REG = D of the
//               previous record (first child), RS_BOTTOM = D_bottom (root)
//       [27:22] wslot: slot to save D(node) into, WS_NONE = not needed
//       [28] leaf  [29] noscore (preamble copy)  [30] root  [31] has_masked
//   w1  tie key: (rank of (n_leaves, bfs_j) among all nodes) << 1
//   w2.. one word per non-masked mutation:
//       [21:0]  site index (row of the allele tile)
//       [23:22] mutated allele index (0..3 = A,C,G,T)
//       [25:24] true parent-state allele index
//       [27:26] reference allele index
//       [31]    after_mask: mutation sits behind the node's first masked
//               mutation (usher_mapper.cpp:197-200): changes D only
constexpr uint32_t RS_REG = 63, RS_BOTTOM = 62, WS_NONE = 63;
constexpr uint32_t F_LEAF = 1u << 28, F_NOSCORE = 1u << 29, F_ROOT = 1u << 30, F_MASKED = 1u << 31;
constexpr uint32_t M_AFTER_MASK = 1u << 31;
constexpr uint32_t MAX_SLOTS = 40;        // > log2(2^32) + 1

constexpr uint32_t T_INFO_MARK = 0xFFFFu;   // tie stream: mutation-count field of a pruning pseudo-record
constexpr uint32_t T_PRUNE_MIN_DWORDS = 8;  // subtrees shorter than this are not worth a record (24 until round 5: k_ties 273 -> 222 us at 16,384 x 10M; 4: the same)

// ---- packed stream ("stream8") walked by k_best8: 8 samples per lane -------
//
// One 32-bit word per event.  Leaves without mutation words never influence a
// result (not eligible, no descendants) and are dropped.  The layout is chosen
// so that the walk needs almost no scalar-ALU work per word (k_best8 is bound
// by the one scalar unit of a CU, not by its four vector units): every flag
// the fast path branches on is one bit (s_bitcmp), slot numbers sit where a
// vector AND yields the LDS byte address, and everything that needs more
// decoding is flagged H_RARE and handled outside the pipelined loop.
//   HDR word (bit 31 set), opens a node:
//       H_REG     D(parent) is the D of the previous node (first effective child, preamble copies)
//       [15:10]   rslot: saved-D slot holding D(parent) when !H_REG  ((w & 0xFC00) = byte offset of the slot row in LDS)
//       H_STORE   D(node) is saved for later children, into
//       [25:20]   wslot  (((w >> 10) & 0xFC00) = byte offset)
//       H_NOSCORE not a candidate here: preamble copy, node carrying a masked
//                 mutation (never eligible), or the root's D record (see below)
//       H_END     node has no mutation words: finish it now
//       H_FREE    every sample is eligible (internal node without mutations)
//       H_SKIPD   D(node) is not needed (no effective children); informational -- computing it anyway is harmless
//       H_RARE    the word leaves the fast path: a pruning record, a chunk end, padding, or a header that
//                 needs the general code (H_SLOW: one of its slots is not among the LDS-resident ones, or the
//                 node has more than 15 mutation words)
//       H_CHUNK_END (with H_RARE) closes a chunk: publish the chunk-local minimum and reset;
//                 [28:10] = words of the chunk that follows, its own end word included (0: none / does not fit)
//       H_NOP     (with H_RARE) padding
//       H_INFO    (with H_RARE, in front of the header of a node with a large subtree) pruning record:
//                 [17:0] jump = stream words occupied by the node's descendants,
//                 [28:22] hsub = largest number of mutation words on any path node -> descendant,
//                 [20:18] hrev = largest number of SECOND HITS on any such path: mutations of a site that is not at its
//                 reference base in the parent (INFO_HR_NONE = 7: more than 6, or a tree with more than 255 mutations on
//                 some root path; the second bound below is not available).
//                 Two lower bounds hold for every descendant d and sample s:
//                   cost(d,s) >= D(node,s) - hsub          each mutation lowers D by at most 1
//                   cost(d,s) >= B(node,s) - hrev          B(n,s) = sites where the sample's set holds the reference base
//                     but not the state at n (D = A + B, A = mismatching sites whose set excludes the reference base).
//                     Along a path a site lowers D at most once more often than it raises it, and only a site whose
//                     state at `node` lies outside the sample's set can do that: one of the A(node,s) sites, or one of
//                     the B(node,s) sites -- whose state at `node` is not the reference base, so that its first mutation
//                     on the path is a second hit.  Hence cost(d,s) >= D - (A + hrev) = B - hrev.
//                 so the subtree is skipped when max(D - hsub, B - hrev) > upper bound of best(s) for
//                 all s.  B counts the mutations that the node has and the sample has not, whatever the sample's own
//                 variants are: a branch leaves a sample's neighbourhood a few foreign mutations away from its lineage,
//                 however deep the subtree below (hsub) is.
//       H_INFO | H_SIB (in front of a non-last child c_i of a node p, before c_i's own H_INFO):
//                 sibling record: [17:0] jump = stream words from c_i's header to the start of p's last child,
//                 [28:22] hs = max over the remaining non-last children c_j (j >= i) of (mutation words of c_j +
//                 hsub(c_j)), [20:18] hr = the same maximum over (second hits of c_j + hrev(c_j)); every node d of
//                 those subtrees has cost(d) >= max(D(p) - hs, B(p) - hr), so when that exceeds the upper bound of
//                 best(s) for all s they are all skipped with one jump.  The non-last children are emitted in
//                 descending order of hs, so hs only shrinks.
//     The root is emitted as two records: its D record (H_REG, H_NOSCORE: "the previous node's D" is D_bottom, which
//     k_best8 loads at the start of every work unit -- the root is the first record of any unit that contains or
//     replays it) followed by a scoring pseudo-node (H_REG | H_SKIPD | H_FREE | H_END) whose cost is D(parent) =
//     D(root): cost(root) = D(root), always eligible (usher_mapper.cpp:454).
//   MUT word (bit 31 clear):
//       [21:0] site  [23:22] mutated allele  [25:24] parent-state allele  [27:26] reference allele
//       M_FLUSH  15 mutations accumulated in the 4-bit counters: spill them (only in nodes whose header is H_SLOW)
//       M_END    last mutation word of the node
constexpr uint32_t H_TAG = 1u << 31, H_INFO = 1u << 30, H_RARE = 1u << 29, H_SIB = 1u << 21;
constexpr uint32_t H_REG = 1u << 0, H_STORE = 1u << 1, H_NOSCORE = 1u << 2, H_END = 1u << 3, H_FREE = 1u << 4,
                   H_SKIPD = 1u << 5, /* bit 6: unused */ H_SLOW = 1u << 7, H_CHUNK_END = 1u << 8, H_NOP = 1u << 9;
constexpr uint32_t H_RSLOT_SHIFT = 10, H_WSLOT_SHIFT = 20;
constexpr uint32_t CE_LEN_SHIFT = 10, CE_LEN_MASK = (1u << 19) - 1u;   // chunk-end word: length of the next chunk
constexpr uint32_t INFO_JUMP_MASK = (1u << 18) - 1u, INFO_HS_SHIFT = 22, INFO_HS_MAX = 127;
constexpr uint32_t INFO_HR_SHIFT = 18, INFO_HR_NONE = 7;   // hrev field; 7 = not available (more than 6 second hits on some path)
// Preamble copies (pre8) carry a pruning record too, in front of every path node below the root: same hs / hr fields
// (hs = PRE_HS_NONE: hsub too large for the field, first bound not available), and [17:0] = the position, relative to
// the chunk's first body word, where the body goes on behind that node's subtree (INFO_JUMP_MASK = beyond any unit).
// When the record's test holds during the replay, the rest of the preamble and the body up to that position are skipped.
constexpr uint32_t PRE_HS_NONE = 127;
constexpr uint32_t PRUNE_MIN_WORDS = 4;     // only subtrees at least this long carry a pruning record
#ifndef UGP_HOT_SLOTS
#define UGP_HOT_SLOTS 16
#endif
constexpr uint32_t MAX_HOT_SLOTS = UGP_HOT_SLOTS;      // the B halves of the hot slots are two register vectors of this many elements in k_best8 (8 or 16)
constexpr uint32_t LDS_SLOTS = 9;           // saved (D, B) slots k_best8 keeps in LDS (1.5 KB each per wave); the colder ones live in a global scratch
constexpr uint32_t M_FLUSH = 1u << 28, M_END = 1u << 30;
// ---- third pruning bound ("B3", round 5) --------------------------------------------------------------------------------------
// A mutation lowers D for sample s only if  (a) its allele is in the sample's set and the reference base is not -- it matches a
// variant of the sample: "useful" for s --  or  (b) the sample's set holds the reference base and the parent state is not in it: a
// second hit.  So along any path n -> d at most  hU_T(n) + hsec(n)  mutations lower D for a sample of tile T, where hU_T(n) = the
// largest number, over the paths below n, of mutations that are useful for SOME sample of T -- and
//     cost(d, s) >= D(n, s) - (hU_T(n) + hsec(n)).
// hsub(n) counts every mutation of the deepest path (3 at the median record, 40-60 near the top); only ~4 % of the mutations are
// useful for a given tile, hU_T is 0 at the median record (tools/analysis/bound_sim2.py: 0.9 % of the tree visited instead of 2.1 %).
// hU_T depends on the tile, so it cannot sit in the records.  With cum(y) = useful events on the root path of y (y included),
// hU_T(n) = max over descendants d of cum(d) - cum(n).  The events of a tile are found through posting lists per (site, allele)
// (FlatMat::b3_events) for the pairs the tile's samples make useful; each raises a counter over the BLOCK range of its subtree:
//     cum_over[b]  = events whose block range contains b          >= cum(y) for every node y with a word in block b
//     cum_under[b] = events whose range contains b strictly inside <= cum(y) for every such y
// and the walk tests, for a record at n with descendants in blocks [q0, q1] and its own last word in block bn,
//     hU_T(n) <= max(cum_over[q0 .. q1]) - cum_under[bn].
// The events are kept in BLOCK order, in groups of B3_GROUP_BLOCKS blocks (one workgroup of ugp_bound3.hip builds the tables of a
// group for 32 tiles at a time: it reads the group's events once and asks a per-batch bit mask "which tiles make this pair
// useful"), as four lists per group: events inside one block, range starts and range ends of the events that span blocks, and the
// events open at the group's first block.
constexpr uint32_t B3_BLOCK_SHIFT = 4, B3_BLOCK_WORDS = 1u << B3_BLOCK_SHIFT;
constexpr uint32_t B3_GROUP_SHIFT = 8, B3_GROUP_BLOCKS = 1u << B3_GROUP_SHIFT;
// blocks per tile row of the tables (whole groups; one block more than the stream has words for)
inline uint32_t b3_blocks(uint64_t stream8_words) { return (uint32_t)((((stream8_words + B3_BLOCK_WORDS - 1) >> B3_BLOCK_SHIFT) + 1u + B3_GROUP_BLOCKS - 1u) & ~(uint64_t)(B3_GROUP_BLOCKS - 1u)); }
constexpr uint32_t MAX_SITES = 1u << 22;
// Version of everything flatten() produces (stream encodings, record fields, event lists): BUMP IT with every change of a layout.
// ugp_flat_save stamps its files with it (next to the flattening switches), ugp_mat_create_from_flat refuses any other -- a build
// time stamp of one translation unit cannot see a change made in another.  6: events listed under their own word's block.
constexpr uint32_t FLAT_FORMAT_VERSION = 6;
constexpr uint32_t MAX_NODE_MUTS = 65534;   // 0xFFFF marks a pruning pseudo-record of the tie stream

struct Options {
    uint32_t chunk_nodes = 0;   // 0 = automatic (about N/32768, at least 128)
    uint32_t prune_min_words = PRUNE_MIN_WORDS;   // subtrees at least this long (stream words) carry a pruning record
    bool sibling_records = true;   // emit H_INFO | H_SIB records
    bool second_bound = true;      // records carry the second-hit counts of the second pruning bound (false: "not available" everywhere)
    bool keep_node_pos8 = false;   // export FlatMat::node_pos8 / rank_dfs (k_best8 names nodes by stream position: the coarse pass's winners, phase 2's ties)
    uint32_t lds_slots = LDS_SLOTS; // headers whose (renumbered) slots are >= this are flagged H_SLOW
    uint32_t pre_weight = 32;   // weight of the preambles' slot accesses when the hot (LDS) slots are chosen: every unit replays one,
                                // the body is mostly skipped (0: body counts only)
    bool keep_update_maps = false;  // export FlatMat::hdr8_of_bfs / rec_of_bfs / post_of_bfs: where each node's record sits in the three streams
                                    // (ugp_mat_update takes rewritten nodes out of the candidate set by patching those words on the device)
    bool keep_b3_events = false;    // export FlatMat::b3_pair_off / b3_events: the posting lists of the third pruning bound (the main tree of a handle)
    uint32_t threads = 0;       // host threads (0 = UGP_FLATTEN_THREADS, else min(32, hardware threads)); the output does not depend on it
};

// Vector whose resize() leaves new elements uninitialised: the big streams are sized first and then written in
// full by the flattening threads, so their pages are first touched in parallel instead of being zeroed by one thread.
template <class T>
struct DefaultInit : std::allocator<T> {
    template <class U> struct rebind { using other = DefaultInit<U>; };
    using std::allocator<T>::allocator;
    template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using UVec = std::vector<T, DefaultInit<T>>;

struct FlatMat {
    uint64_t n_nodes = 0, n_muts = 0, n_sites = 0;
    uint32_t max_pos = 0, max_slots = 0, n_chunks = 0;
    UVec<uint32_t> stream;          // node records, DFS order
    UVec<uint32_t> pre_stream;      // per-chunk preamble records (root path of the chunk's first node)
    std::vector<uint32_t> chunk_body_off;  // [n_chunks+1] dword offsets into stream
    std::vector<uint32_t> chunk_pre_off;   // [n_chunks+1] dword offsets into pre_stream
    std::vector<uint32_t> chunk_node_off;  // [n_chunks+1] DFS index of the chunk's first node
    std::vector<int32_t> pos2site;         // [max_pos+1], -1 = position never mutated in the tree
    std::vector<uint8_t> site_ref;         // [n_sites] one-hot reference base
    UVec<uint32_t> rank2bfs;        // [n_nodes] tie rank -> BFS index
    UVec<uint32_t> dfs2bfs;         // [n_nodes]
    // packed stream for k_best8 (same chunk cut points, by DFS node index)
    UVec<uint32_t> stream8, pre8_stream;
    std::vector<uint32_t> chunk8_body_off, chunk8_pre_off;   // [n_chunks+1]
    // Tie stream (phase 2): the chunk bodies of `stream` without the leaves that can never be eligible and
    // with pruning pseudo-records {w0 = T_INFO_MARK | second hits below << 16 (255: not available), w1 = hsub << 24 | jump}: the node that follows may be
    // skipped together with its descendants (`jump` dwords behind its own record, inside the chunk)
    // when D(node) - hsub exceeds the wanted score of every sample that still looks for ties here.
    UVec<uint32_t> stream_t, chunk_t_off;             // chunk_t_off: [n_chunks+1]
    uint32_t max_path_muts = 0;            // max over nodes of the mutation count on the root path
    std::vector<uint32_t> rank_dfs;        // (Options::keep_node_pos8) [n_nodes] by DFS index: the node's tie rank
    std::vector<uint32_t> node_pos8;       // (Options::keep_node_pos8) [n_nodes + 1] by DFS index: packed-stream position where the node's words begin
    // (Options::keep_update_maps) by BFS index: position of the node's header word in stream8, of its record (w0) in stream and in
    // stream_t; UINT32_MAX = the node has no record there (leaves without mutation words are dropped from stream8 / stream_t)
    UVec<uint32_t> hdr8_of_bfs, rec_of_bfs, post_of_bfs;
    // (Options::keep_b3_events) third pruning bound: the mutation events (one per mutation word of the packed body) in block order.
    // b3_events[i] = 4 * site + allele index (bits 23:0) | block within its group (31:24); list k of group g =
    // [b3_group_off[k * (n_groups + 1) + g], ...[.. + g + 1]) with k = 0: events whose range -- from the mutation word itself to the
    // end of its node's subtree -- lies inside one block of B3_BLOCK_WORDS packed-stream words, listed under that block; 1: the
    // others, under the block of the mutation word (round 6; the node's header word until then: a long branch put all its events
    // into one block and overflowed the table kernel's 8-bit counter);
    // 2: the same events again, under the last block of the subtree; 3: the events that are open at the group's first block (word
    // in an earlier group, subtree not over before this one: the root path's mutation words there) -- what a group needs to know of
    // everything in front of it.  n_groups = b3_blocks(stream8 words) / B3_GROUP_BLOCKS.
    std::vector<uint32_t> b3_group_off;
    UVec<uint32_t> b3_events;
    uint32_t max_chunk8_words = 0;         // longest chunk of the packed stream (a work unit must stay below the reach of a preamble record's jump field)
    uint32_t lds_slots = 0;                // the Options value the packed stream was encoded for (<= max_slots)
    bool mask_not_first = false;           // some non-root node lists a masked mutation behind an ordinary one: only the
                                           // 32-bit walk (M_AFTER_MASK) scores such a node the way usher_mapper.cpp:190-270 does
};

// By-products of the flattening that the caller may want (both by BFS index): subtree sizes, DFS position.
struct FlatExtras {
    UVec<uint32_t> sub, dfsidx;
    // only when the input is a breadth-first expansion (parent[] ascending), else empty: node_pair[2j] = child_begin[j]
    // (the children of node j are the nodes child_begin[j] + 1 .. child_begin[j + 1]), node_pair[2j + 1] = dword offset of
    // j's record in FlatMat::stream; n_nodes + 1 pairs
    UVec<uint32_t> node_pair;
    uint64_t children_of_wide_nodes = 0;   // nodes that are a child of a node with more than 16 children
};

// Returns UGP_OK or a negative UGP_ERR_* with `err` filled.
int flatten(const ugp_tree_desc &t, const Options &opt, FlatMat &out, std::string &err, FlatExtras *extras = nullptr);

// Contiguous ranges of [0, n) on up to T host threads; fn(begin, end, thread index).  A pass with fewer than
// min_per_thread items per thread uses fewer threads (one: it runs inline).
struct Par {
    unsigned T;
    uint64_t grain = 0;   // != 0: overrides min_per_thread (UGP_FLATTEN_GRAIN; the tests set 1 to split even tiny passes)
    template <class F>
    void run(uint64_t n, F fn, uint64_t min_per_thread = 8192) const;
    uint64_t exclusive_scan(uint32_t *a, uint64_t n) const;   // a[i] <- sum of a[0..i); returns the total
};
unsigned flatten_threads(const Options &opt);   // Options.threads, else UGP_FLATTEN_THREADS, else min(32, hardware threads)
Par flatten_par(const Options &opt);

template <class F>
void Par::run(uint64_t n, F fn, uint64_t min_per_thread) const {
    if (grain) min_per_thread = grain;
    const unsigned t = (unsigned)std::min<uint64_t>(T, std::max<uint64_t>(1, n / std::max<uint64_t>(1, min_per_thread)));
    if (t <= 1) { fn((uint64_t)0, n, 0u); return; }
    std::vector<std::thread> th;
    th.reserve(t - 1);
    for (unsigned i = 1; i < t; i++) th.emplace_back([&fn, n, t, i] { fn(n * i / t, n * (i + 1) / t, i); });
    fn((uint64_t)0, n / t, 0u);
    for (auto &x : th) x.join();
}

}  // namespace ugp
