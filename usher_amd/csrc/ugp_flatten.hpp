// ugp_flatten.hpp -- host-side flattening of a mutation-annotated tree into the
// DFS record stream the HIP kernels walk.  Pure C++ (no HIP), so the CPU test
// suite can exercise it without a GPU.
#pragma once
#include <cstdint>
#include <string>
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
constexpr uint32_t T_PRUNE_MIN_DWORDS = 24;  // subtrees shorter than this are not worth a record

// ---- packed stream ("stream8") walked by k_best8: 8 samples per lane -------
//
// One 32-bit word per event.  Leaves without mutation words never influence a
// result (not eligible, no descendants) and are dropped.
//   HDR word (bit 31 set), opens a node:
//       [5:0] rslot  [11:6] wslot   (same meaning as above, on the effective tree)
//       H_SKIPD   D(node) is not needed (no effective children)
//       H_NOSCORE not a candidate here: preamble copy, node carrying a masked
//                 mutation (never eligible), or the root's D record (see below)
//       H_FREE    every sample is eligible (internal node without mutations)
//       H_END     node has no mutation words: finish it now
//       H_CHUNK_END (alone) closes a chunk: publish the chunk-local minimum and reset
//       H_NOP     padding
//       H_INFO    (alone, in front of the header of a node with a large subtree) pruning record:
//                 [20:0] jump = stream words occupied by the node's descendants,
//                 [29:22] hsub = largest number of mutation words on any path node -> descendant;
//                 every descendant d has cost(d) >= D(node) - hsub (each mutation lowers D by at most 1),
//                 so the subtree can be skipped when D(node,s) - hsub > upper bound of best(s) for all s
//       H_INFO | H_SIB (alone, in front of a non-last child c_i of a node p, before c_i's own H_INFO):
//                 sibling record: [20:0] jump = stream words from c_i's header to the start of p's last child,
//                 [29:22] hs = max over the remaining non-last children c_j (j >= i) of (mutation words of c_j +
//                 hsub(c_j)); every node d of those subtrees has cost(d) >= D(p) - hs, so when
//                 D(p,s) - hs > upper bound of best(s) for all s they are all skipped with one jump.  The
//                 non-last children are emitted in descending order of that quantity, so hs only shrinks.
//                 (Both kinds of record keep their jump in 21 bits.)
//     The root is emitted as two records: its D record (rslot = RS_BOTTOM,
//     H_NOSCORE) followed by a scoring pseudo-node (RS_REG, H_SKIPD | H_FREE |
//     H_END) whose cost is D(parent) = D(root): cost(root) = D(root), always
//     eligible (usher_mapper.cpp:454).
//   MUT word (bit 31 clear):
//       [21:0] site  [23:22] mutated allele  [25:24] parent-state allele  [27:26] reference allele
//       M_FLUSH  15 mutations accumulated in the 4-bit counters: spill them
//       M_END    last mutation word of the node
constexpr uint32_t H_TAG = 1u << 31, H_SKIPD = 1u << 12, H_NOSCORE = 1u << 13, H_END = 1u << 16,
                   H_FREE = 1u << 17, H_CHUNK_END = 1u << 18, H_NOP = 1u << 19, H_INFO = 1u << 30,
                   H_SIB = 1u << 21;   // with H_INFO: sibling record
constexpr uint32_t INFO_JUMP_MASK = (1u << 21) - 1u;
constexpr uint32_t PRUNE_MIN_WORDS = 4;     // only subtrees at least this long carry a pruning record
constexpr uint32_t M_FLUSH = 1u << 28, M_END = 1u << 30;
constexpr uint32_t MAX_SITES = 1u << 22;
constexpr uint32_t MAX_NODE_MUTS = 65534;   // 0xFFFF marks a pruning pseudo-record of the tie stream

struct Options {
    uint32_t chunk_nodes = 0;   // 0 = automatic (about N/32768, at least 128)
    uint32_t prune_min_words = PRUNE_MIN_WORDS;   // subtrees at least this long (stream words) carry a pruning record
    bool sibling_records = true;   // emit H_INFO | H_SIB records
};

struct FlatMat {
    uint64_t n_nodes = 0, n_muts = 0, n_sites = 0;
    uint32_t max_pos = 0, max_slots = 0, n_chunks = 0;
    std::vector<uint32_t> stream;          // node records, DFS order
    std::vector<uint32_t> pre_stream;      // per-chunk preamble records (root path of the chunk's first node)
    std::vector<uint32_t> chunk_body_off;  // [n_chunks+1] dword offsets into stream
    std::vector<uint32_t> chunk_pre_off;   // [n_chunks+1] dword offsets into pre_stream
    std::vector<uint32_t> chunk_node_off;  // [n_chunks+1] DFS index of the chunk's first node
    std::vector<int32_t> pos2site;         // [max_pos+1], -1 = position never mutated in the tree
    std::vector<uint8_t> site_ref;         // [n_sites] one-hot reference base
    std::vector<uint32_t> rank2bfs;        // [n_nodes] tie rank -> BFS index
    std::vector<uint32_t> dfs2bfs;         // [n_nodes]
    // packed stream for k_best8 (same chunk cut points, by DFS node index)
    std::vector<uint32_t> stream8, pre8_stream;
    std::vector<uint32_t> chunk8_body_off, chunk8_pre_off;   // [n_chunks+1]
    // Tie stream (phase 2): the chunk bodies of `stream` without the leaves that can never be eligible and
    // with pruning pseudo-records {w0 = T_INFO_MARK, w1 = hsub << 24 | jump}: the node that follows may be
    // skipped together with its descendants (`jump` dwords behind its own record, inside the chunk)
    // when D(node) - hsub exceeds the wanted score of every sample that still looks for ties here.
    std::vector<uint32_t> stream_t, chunk_t_off;             // chunk_t_off: [n_chunks+1]
    uint32_t max_path_muts = 0;            // max over nodes of the mutation count on the root path
    bool mask_not_first = false;           // some non-root node lists a masked mutation behind an ordinary one: only the
                                           // 32-bit walk (M_AFTER_MASK) scores such a node the way usher_mapper.cpp:190-270 does
};

// Returns UGP_OK or a negative UGP_ERR_* with `err` filled.
int flatten(const ugp_tree_desc &t, const Options &opt, FlatMat &out, std::string &err);

}  // namespace ugp
