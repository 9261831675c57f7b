// ugp_flatten.cpp -- BFS-ordered tree arrays -> DFS record stream (host, C++).
//
// What the reference does per sample (rebuild the BFS vector, allocate 2N
// vectors, walk parent pointers to the root for every node:
// usher_common.cpp:342-365, usher_mapper.cpp:275-286) is hoisted here and done
// once per tree: the true parent state of every mutation, the descendant-leaf
// counts (Tree::get_num_leaves, mutation_annotated_tree.cpp:866-879), the
// (n_leaves, j) tie rank (usher_mapper.cpp:483-486) and a traversal order whose
// running state fits a log2(N)-deep per-lane stack.
#include "ugp_flatten.hpp"

#include <algorithm>
#include <unordered_map>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace ugp {

// Intermediate encoding of the packed stream while it is being emitted (slot numbers are renumbered by use
// afterwards, and only then can a header be told apart as fast-path or H_SLOW): rslot [5:0], wslot [11:6],
// flags below; pruning records {E_INFO, hs [29:22], jump [20:0]}.  finalize8() converts to the layout of
// ugp_flatten.hpp.
namespace {
constexpr uint32_t E_SKIPD = 1u << 12, E_NOSCORE = 1u << 13, E_END = 1u << 16, E_FREE = 1u << 17, E_CHUNK_END = 1u << 18,
                   E_NOP = 1u << 19, E_SIB = 1u << 21, E_INFO = 1u << 30;
}

static inline int nuc_index(uint8_t onehot) {
    switch (onehot) {
        case 1: return 0;
        case 2: return 1;
        case 4: return 2;
        case 8: return 3;
        default: return -1;
    }
}

int flatten(const ugp_tree_desc &t, const Options &opt, FlatMat &out, std::string &err) {
    const uint64_t N = t.n_nodes;
    if (N == 0 || !t.parent || !t.mut_off) { err = "empty tree or null arrays"; return UGP_ERR_INVALID; }
    if (N >= (1ull << 31)) { err = "more than 2^31 nodes"; return UGP_ERR_UNSUPPORTED; }
    if (t.parent[0] != UINT32_MAX) { err = "parent[0] must be UINT32_MAX (root first, BFS order)"; return UGP_ERR_INVALID; }
    const uint64_t M = t.mut_off[N];
    if (M && (!t.mut_pos || !t.mut_ref || !t.mut_nuc)) { err = "null mutation arrays"; return UGP_ERR_INVALID; }

    const bool verbose = getenv("UGP_FLATTEN_VERBOSE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto flat_lap = [&](const char *what) {
        if (!verbose) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[ugp flatten] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };
    // ---- children (BFS order keeps each node's children contiguous & ordered)
    std::vector<uint32_t> child_off(N + 1, 0);
    for (uint64_t j = 1; j < N; j++) {
        if (t.parent[j] >= j) { err = "parent[j] must be < j (BFS order)"; return UGP_ERR_INVALID; }
        child_off[t.parent[j] + 1]++;
    }
    for (uint64_t j = 0; j < N; j++) child_off[j + 1] += child_off[j];
    std::vector<uint32_t> children(N > 1 ? N - 1 : 0);
    {
        std::vector<uint32_t> fill(child_off.begin(), child_off.end() - 1);
        for (uint64_t j = 1; j < N; j++) children[fill[t.parent[j]]++] = (uint32_t)j;
    }
    flat_lap("children");
    // ---- subtree sizes and descendant-leaf counts
    std::vector<uint32_t> sub(N, 1), leaves(N, 0);
    for (uint64_t j = N; j-- > 0;) {
        if (child_off[j + 1] == child_off[j]) leaves[j] = 1;
        if (j > 0) { sub[t.parent[j]] += sub[j]; leaves[t.parent[j]] += leaves[j]; }
    }
    // hdown[j] = own mutation words of j + the largest number of mutation words on a path j -> descendant:
    // no node of j's subtree (j included) costs less than D(parent(j)) - hdown[j]
    std::vector<uint32_t> hdown(N, 0);
    for (uint64_t j = N; j-- > 0;) {
        uint32_t nw = 0;
        for (uint64_t i = t.mut_off[j]; i < t.mut_off[j + 1] && i < M; i++) nw += t.mut_pos[i] >= 0;
        hdown[j] += nw;
        if (j > 0) hdown[t.parent[j]] = std::max(hdown[t.parent[j]], hdown[j]);   // (the parent's own words are added when it is visited)
    }
    // largest subtree last; the others by descending hdown (the sibling pruning records rely on it)
    for (uint64_t j = 0; j < N; j++) {
        uint32_t b = child_off[j], e = child_off[j + 1];
        if (e - b < 2) continue;
        uint32_t best = b;
        for (uint32_t k = b + 1; k < e; k++) if (sub[children[k]] > sub[children[best]]) best = k;
        uint32_t c = children[best];
        for (uint32_t k = best; k + 1 < e; k++) children[k] = children[k + 1];
        children[e - 1] = c;
        if (e - b > 2) std::stable_sort(children.begin() + b, children.begin() + e - 1, [&](uint32_t x, uint32_t y) { return hdown[x] > hdown[y]; });
    }
    flat_lap("subtree sizes");
    // ---- tie rank: ascending (n_leaves, j)
    // (stable LSD radix sort of j by leaves[j], 11 bits per pass; as many passes as the largest count needs)
    out.rank2bfs.resize(N);
    std::iota(out.rank2bfs.begin(), out.rank2bfs.end(), 0u);
    {
        std::vector<uint32_t> tmp(N);
        const uint32_t max_leaves = leaves[0];
        for (uint32_t shift = 0; shift < 32 && (max_leaves >> shift) != 0; shift += 11) {
            uint64_t count[2049] = {0};
            for (uint64_t i = 0; i < N; i++) count[((leaves[out.rank2bfs[i]] >> shift) & 2047u) + 1]++;
            for (int b = 0; b < 2048; b++) count[b + 1] += count[b];
            for (uint64_t i = 0; i < N; i++) {
                const uint32_t j = out.rank2bfs[i];
                tmp[count[(leaves[j] >> shift) & 2047u]++] = j;
            }
            out.rank2bfs.swap(tmp);
        }
    }
    std::vector<uint32_t> rank(N);
    for (uint64_t r = 0; r < N; r++) rank[out.rank2bfs[r]] = (uint32_t)r;

    flat_lap("tie rank");
    // ---- sites
    int32_t max_pos = 0;
    for (uint64_t i = 0; i < M; i++) max_pos = std::max(max_pos, t.mut_pos[i]);
    out.max_pos = (uint32_t)max_pos;
    out.pos2site.assign((size_t)max_pos + 1, -1);
    out.site_ref.clear();
    uint64_t n_real = 0;
    for (uint64_t i = 0; i < M; i++) {
        int32_t p = t.mut_pos[i];
        if (p < 0) continue;
        n_real++;
        if (nuc_index(t.mut_nuc[i]) < 0) { err = "tree mutation allele is not one-hot (ambiguous MAT alleles are unsupported)"; return UGP_ERR_UNSUPPORTED; }
        if (nuc_index(t.mut_ref[i]) < 0) { err = "tree mutation reference base is not one-hot"; return UGP_ERR_UNSUPPORTED; }
        if (out.pos2site[p] < 0) {
            out.pos2site[p] = (int32_t)out.site_ref.size();
            out.site_ref.push_back(t.mut_ref[i]);
        } else if (out.site_ref[out.pos2site[p]] != t.mut_ref[i]) {
            err = "tree mutations disagree on the reference base at position " + std::to_string(p);
            return UGP_ERR_UNSUPPORTED;
        }
    }
    // number sites by position so a tile row index grows with the genome coordinate
    {
        std::vector<uint8_t> ref_by_pos;
        uint32_t s = 0;
        std::vector<uint8_t> new_ref(out.site_ref.size());
        for (int32_t p = 0; p <= max_pos; p++) {
            if (out.pos2site[p] < 0) continue;
            new_ref[s] = out.site_ref[out.pos2site[p]];
            out.pos2site[p] = (int32_t)s++;
        }
        out.site_ref.swap(new_ref);
    }
    out.n_sites = out.site_ref.size();
    if (out.n_sites > MAX_SITES) { err = "more than 2^22 mutated positions"; return UGP_ERR_UNSUPPORTED; }
    out.n_nodes = N;
    out.n_muts = n_real;

    flat_lap("sites");
    // ---- DFS emission
    out.stream.clear();
    out.mask_not_first = false;
    out.stream.reserve(2 * N + n_real);
    out.dfs2bfs.resize(N);
    std::vector<uint32_t> rec_off(N);        // by BFS index: dword offset of the node's record
    std::vector<uint8_t> node_masked(N, 0);  // by BFS index: carries a masked mutation (non-root)
    std::vector<uint8_t> slot(N, 0);         // by BFS index
    std::vector<uint8_t> cur(out.n_sites);   // running allele index per site
    for (uint64_t s = 0; s < out.n_sites; s++) cur[s] = (uint8_t)nuc_index(out.site_ref[s]);
    struct Undo { uint32_t site; uint8_t old; };
    std::vector<Undo> undo;
    struct Frame { uint32_t node, next, undo_mark; };
    std::vector<Frame> stack;
    uint32_t max_slot_used = 0;
    uint64_t dfs_idx = 0;

    auto emit = [&](uint32_t j, bool first_child) -> int {
        const uint64_t b = t.mut_off[j], e = t.mut_off[j + 1];
        if (e < b || e > M) { err = "mut_off is not monotone"; return UGP_ERR_INVALID; }
        const uint32_t nch = child_off[j + 1] - child_off[j];
        uint32_t w0 = 0;
        const bool root = (j == 0);
        uint32_t rslot = root ? RS_BOTTOM : (first_child ? RS_REG : slot[t.parent[j]]);
        uint32_t wslot = WS_NONE;
        if (nch >= 2) { wslot = slot[j]; max_slot_used = std::max<uint32_t>(max_slot_used, slot[j] + 1u); }
        if (nch == 0) w0 |= F_LEAF;
        if (root) w0 |= F_ROOT;
        rec_off[j] = (uint32_t)out.stream.size();
        out.stream.push_back(0);
        out.stream.push_back(rank[j] << 1);
        uint32_t nwords = 0;
        bool masked = false;
        int32_t last_pos = -1;
        for (uint64_t i = b; i < e; i++) {
            int32_t p = t.mut_pos[i];
            if (p < 0) { if (nwords > 0 && !root) out.mask_not_first = true; masked = true; continue; }
            if (p == last_pos) { err = "node carries two mutations at position " + std::to_string(p); return UGP_ERR_UNSUPPORTED; }
            last_pos = p;   // (adjacent duplicates; the loader keeps lists sorted)
            uint32_t site = (uint32_t)out.pos2site[p];
            uint32_t mi = (uint32_t)nuc_index(t.mut_nuc[i]);
            uint32_t pi = cur[site];
            uint32_t ri = (uint32_t)nuc_index(out.site_ref[site]);
            uint32_t w = site | (mi << 22) | (pi << 24) | (ri << 26);
            if (masked && !root) w |= M_AFTER_MASK;
            out.stream.push_back(w);
            undo.push_back({site, cur[site]});
            cur[site] = (uint8_t)mi;
            nwords++;
        }
        if (nwords > MAX_NODE_MUTS) { err = "node with more than 65534 mutations"; return UGP_ERR_UNSUPPORTED; }
        if (masked && !root) node_masked[j] = 1;
        if (masked && !root) w0 |= F_MASKED;   // root: masked mutations are inert (usher_mapper.cpp:266-269, 309-311, 401-403)
        w0 |= nwords | (rslot << 16) | (wslot << 22);
        out.stream[rec_off[j]] = w0;
        out.dfs2bfs[dfs_idx++] = j;
        return UGP_OK;
    };

    stack.push_back({0u, 0u, 0u});
    slot[0] = 0;
    { int rc = emit(0, false); if (rc) return rc; }
    while (!stack.empty()) {
        Frame &f = stack.back();
        const uint32_t b = child_off[f.node], e = child_off[f.node + 1];
        if (b + f.next < e) {
            const uint32_t k = f.next++;
            const uint32_t c = children[b + k];
            const bool last = (b + k + 1 == e);
            uint32_t s = last ? slot[f.node] : slot[f.node] + 1u;
            if (s >= MAX_SLOTS) { err = "D stack deeper than MAX_SLOTS"; return UGP_ERR_UNSUPPORTED; }
            slot[c] = (uint8_t)s;
            const uint32_t mark = (uint32_t)undo.size();
            int rc = emit(c, k == 0);
            if (rc) return rc;
            stack.push_back({c, 0u, mark});
        } else {
            // leaving: restore the running state
            const uint32_t mark = f.undo_mark;
            while (undo.size() > mark) { cur[undo.back().site] = undo.back().old; undo.pop_back(); }
            stack.pop_back();
        }
    }
    out.max_slots = std::max<uint32_t>(max_slot_used, 1u);

    flat_lap("DFS emission");
    // ---- chunks: equal dword budgets, cut at node boundaries
    // ~300 nodes per chunk on a 10M-node tree: chunks are the granule of the phase-1 minima (short phase-2
    // re-walks) and of the work units (16 / 32 chunks each outside / inside a tile's own region)
    uint64_t chunk_nodes = opt.chunk_nodes ? opt.chunk_nodes : std::max<uint64_t>(128, N / 32768);
    uint64_t want_chunks = std::max<uint64_t>(1, (N + chunk_nodes - 1) / chunk_nodes);
    const uint64_t total = out.stream.size();
    out.chunk_body_off.clear(); out.chunk_node_off.clear(); out.chunk_pre_off.clear(); out.pre_stream.clear();
    {
        uint64_t next_cut = 0, c = 0;
        for (uint64_t d = 0; d < N; d++) {
            uint32_t off = rec_off[out.dfs2bfs[d]];
            if (off >= next_cut && c < want_chunks) {
                out.chunk_body_off.push_back(off);
                out.chunk_node_off.push_back((uint32_t)d);
                c++;
                next_cut = total * c / want_chunks;
                if (next_cut <= off) next_cut = off + 1;
            }
        }
        out.chunk_body_off.push_back((uint32_t)total);
        out.chunk_node_off.push_back((uint32_t)N);
    }
    out.n_chunks = (uint32_t)out.chunk_body_off.size() - 1;
    flat_lap("chunks");
    // ---- preambles: the root path of each chunk's first node, replayed without scoring
    std::vector<uint32_t> path;
    for (uint32_t c = 0; c < out.n_chunks; c++) {
        out.chunk_pre_off.push_back((uint32_t)out.pre_stream.size());
        uint32_t a = out.dfs2bfs[out.chunk_node_off[c]];
        path.clear();
        for (uint32_t q = a; q != 0;) { q = t.parent[q]; path.push_back(q); }
        for (size_t i = path.size(); i-- > 0;) {
            uint32_t q = path[i];
            uint32_t off = rec_off[q];
            uint32_t w0 = out.stream[off];
            uint32_t nwords = w0 & 0xFFFF;
            uint32_t rs = (q == 0) ? RS_BOTTOM : RS_REG;
            w0 = (w0 & ~(63u << 16)) | (rs << 16) | F_NOSCORE;
            out.pre_stream.push_back(w0);
            out.pre_stream.push_back(out.stream[off + 1]);
            for (uint32_t k = 0; k < nwords; k++) out.pre_stream.push_back(out.stream[off + 2 + k]);
        }
    }
    out.chunk_pre_off.push_back((uint32_t)out.pre_stream.size());

    flat_lap("preambles");
    // ---- packed stream (stream8) on the effective tree -----------------------
    // dropped[j]: leaf whose record has no mutation words (never eligible, no descendants)
    std::vector<uint8_t> dropped(N, 0);
    for (uint64_t j = 1; j < N; j++)
        if (child_off[j + 1] == child_off[j] && (out.stream[rec_off[j]] & 0xFFFFu) == 0) dropped[j] = 1;
    std::vector<uint32_t> eff_children(N, 0);
    std::vector<uint8_t> first_eff(N, 0);     // j is the first effective child of its parent
    for (uint64_t d = 0; d < N; d++) {        // DFS order: children of a node appear in emission order
        if (d + 24 < N) {   // nodes are visited in DFS order but stored by BFS index: fetch ahead
            const uint32_t jn = out.dfs2bfs[d + 24];
            __builtin_prefetch(&dropped[jn]); __builtin_prefetch(&t.parent[jn]); __builtin_prefetch(&first_eff[jn]);
        }
        if (d + 12 < N) __builtin_prefetch(&eff_children[t.parent[out.dfs2bfs[d + 12]]]);
        uint32_t j = out.dfs2bfs[d];
        if (j == 0 || dropped[j]) continue;
        if (eff_children[t.parent[j]]++ == 0) first_eff[j] = 1;
    }
    flat_lap("  eff children");
    // mutation count on the root path (bound for the 16-bit counters)
    {
        std::vector<uint32_t> path(N, 0);
        uint32_t mx = 0;
        for (uint64_t j = 0; j < N; j++) {   // BFS order: parents first
            uint32_t own = out.stream[rec_off[j]] & 0xFFFFu;
            path[j] = (j ? path[t.parent[j]] : 0) + own;
            mx = std::max(mx, path[j]);
        }
        out.max_path_muts = mx;
    }
    flat_lap("  path muts");
    auto emit8 = [&](std::vector<uint32_t> &dst, uint32_t j, bool preamble) {
        const uint32_t off = rec_off[j];
        const uint32_t w0 = out.stream[off];
        const uint32_t nwords = w0 & 0xFFFFu;
        const bool root = (j == 0);
        const uint32_t nch = child_off[j + 1] - child_off[j];
        uint32_t rslot = root ? RS_BOTTOM : ((preamble || first_eff[j]) ? RS_REG : slot[t.parent[j]]);
        uint32_t wslot = (eff_children[j] >= 2) ? slot[j] : WS_NONE;
        uint32_t h = H_TAG | rslot | (wslot << 6);
        if (eff_children[j] == 0 && !root) h |= E_SKIPD;
        if (preamble || root || node_masked[j]) h |= E_NOSCORE;
        if (nwords == 0) h |= E_END;
        if (!root && nch > 0 && nwords == 0 && !node_masked[j]) h |= E_FREE;
        dst.push_back(h);
        for (uint32_t k = 0; k < nwords; k++) {
            uint32_t w = out.stream[off + 2 + k] & 0x0FFFFFFFu;   // site, mutated / parent-state / reference allele
            if (k + 1 == nwords) w |= M_END;
            else if ((k + 1) % 15 == 0) w |= M_FLUSH;
            dst.push_back(w);
        }
        // the root scores through a pseudo-node right behind its D record: cost = D(parent) = D(root)
        if (root && !preamble) dst.push_back(H_TAG | RS_REG | (WS_NONE << 6) | E_SKIPD | E_FREE | E_END);
    };
    // pruning records: hsub[j] = max mutation words on a path j -> descendant (excluding j's own),
    // subw[j] = stream words of j's descendants (approximate: without the records themselves)
    std::vector<uint32_t> hsub(N, 0), subw(N, 0), dfsidx(N, 0);
    for (uint64_t d = 0; d < N; d++) dfsidx[out.dfs2bfs[d]] = (uint32_t)d;
    for (uint64_t j = N; j-- > 1;) {
        const uint32_t nw = out.stream[rec_off[j]] & 0xFFFFu;
        const uint32_t p = t.parent[j];
        hsub[p] = std::max(hsub[p], nw + hsub[j]);
        subw[p] += subw[j] + (dropped[j] ? 0u : 1u + nw);
    }
    flat_lap("  hsub/subw");
    // sibling records.  last_eff[p] = p's last effective child; for a non-last effective child j:
    // suffix_h[j] = max hdown over j and the non-last effective siblings after it, big_after[j] = how many of
    // those later siblings carry a pruning record of their own (a jump that the sibling record can save)
    std::vector<uint32_t> last_eff(N, UINT32_MAX), suffix_h(N, 0), big_after(N, 0), sib_slot(N, UINT32_MAX);
    for (uint64_t p = 0; p < N; p++) {
        const uint32_t b = child_off[p], e = child_off[p + 1];
        uint32_t last = UINT32_MAX;
        for (uint32_t k = e; k-- > b;) if (!dropped[children[k]]) { last = children[k]; break; }
        last_eff[p] = last;
        uint32_t run_h = 0, run_big = 0;
        for (uint32_t k = e; k-- > b;) {
            const uint32_t c = children[k];
            if (dropped[c] || c == last) continue;
            big_after[c] = run_big;
            run_h = std::max(run_h, hdown[c]);
            suffix_h[c] = run_h;
            if (subw[c] >= opt.prune_min_words && hsub[c] <= INFO_HS_MAX) run_big++;
        }
    }
    std::vector<std::vector<uint32_t>> sib_open;   // per open parent: positions of its unpatched sibling records
    std::vector<uint32_t> free_sib;
    std::unordered_map<uint32_t, uint32_t> sib_hdr_pos;   // record position -> position of the header it precedes
    uint32_t sib_pending_hdr = UINT32_MAX;
    const bool preamble_only = !opt.sibling_records;   // (no sibling records at all)
    struct OpenBig { uint32_t info_pos, own_end, dfs_end; };
    std::vector<OpenBig> open_big;
    auto close_big = [&](uint32_t next_dfs) {   // patch the records of subtrees that end before DFS node next_dfs
        while (!open_big.empty() && open_big.back().dfs_end <= next_dfs) {
            const OpenBig &b = open_big.back();
            const uint64_t jump = out.stream8.size() - b.own_end;
            const uint32_t hs = (out.stream8[b.info_pos] >> 22) & 0xFFu;
            out.stream8[b.info_pos] = (jump <= INFO_JUMP_MASK && jump > 0) ? (H_TAG | E_INFO | (hs << 22) | (uint32_t)jump) : (H_TAG | E_NOP);
            open_big.pop_back();
        }
    };
    out.stream8.clear(); out.pre8_stream.clear(); out.chunk8_body_off.clear(); out.chunk8_pre_off.clear();
    out.stream8.reserve(N + n_real + out.n_chunks);
    // by DFS index: stream position when the node is reached, of its first word behind a sibling record, of its header
    std::vector<uint32_t> pos8_at(N + 1, 0), pos8_start(N, 0), pos8_hdr(N, 0);
    for (uint32_t c = 0; c < out.n_chunks; c++) {
        out.chunk8_body_off.push_back((uint32_t)out.stream8.size());
        for (uint32_t d = out.chunk_node_off[c]; d < out.chunk_node_off[c + 1]; d++) {
            if ((uint64_t)d + 24 < N) {
                const uint32_t jn = out.dfs2bfs[d + 24];
                __builtin_prefetch(&dropped[jn]); __builtin_prefetch(&subw[jn]); __builtin_prefetch(&hsub[jn]); __builtin_prefetch(&sub[jn]);
                __builtin_prefetch(&rec_off[jn]); __builtin_prefetch(&child_off[jn]); __builtin_prefetch(&first_eff[jn]);
                __builtin_prefetch(&slot[jn]); __builtin_prefetch(&eff_children[jn]); __builtin_prefetch(&node_masked[jn]);
                __builtin_prefetch(&t.parent[jn]);
            }
            if ((uint64_t)d + 12 < N) __builtin_prefetch(&slot[t.parent[out.dfs2bfs[d + 12]]]);
            uint32_t j = out.dfs2bfs[d];
            close_big(d);
            pos8_at[d] = (uint32_t)out.stream8.size();
            if (dropped[j]) continue;
            // sibling record: j is a non-last effective child and at least one more non-last sibling with a
            // pruning record of its own follows (otherwise there is no jump to save)
            if (j != 0) {
                const uint32_t p = t.parent[j];
                if (last_eff[p] == j) {   // p's last child starts here: the pending sibling records of p jump to this word
                  if (sib_slot[p] != UINT32_MAX) {
                    for (uint32_t pos8 : sib_open[sib_slot[p]]) {
                        const uint64_t jump = out.stream8.size() - sib_hdr_pos[pos8];
                        uint32_t &w = out.stream8[pos8];
                        w = (jump <= INFO_JUMP_MASK && jump > 0) ? (w | (uint32_t)jump) : (H_TAG | E_NOP);
                    }
                    sib_open[sib_slot[p]].clear();
                    free_sib.push_back(sib_slot[p]);
                    sib_slot[p] = UINT32_MAX;
                  }
                } else if (big_after[j] >= 1 && suffix_h[j] <= INFO_HS_MAX && !preamble_only) {
                    if (sib_slot[p] == UINT32_MAX) {
                        if (free_sib.empty()) { free_sib.push_back((uint32_t)sib_open.size()); sib_open.emplace_back(); }
                        sib_slot[p] = free_sib.back(); free_sib.pop_back();
                    }
                    sib_open[sib_slot[p]].push_back((uint32_t)out.stream8.size());
                    sib_pending_hdr = (uint32_t)out.stream8.size();
                    out.stream8.push_back(H_TAG | E_INFO | E_SIB | (suffix_h[j] << 22));   // jump patched when the last child starts
                }
            }
            const bool big = j != 0 && subw[j] >= opt.prune_min_words && hsub[j] <= INFO_HS_MAX;
            pos8_start[d] = (uint32_t)out.stream8.size();   // (behind the sibling record, if any)
            if (big) {
                open_big.push_back({(uint32_t)out.stream8.size(), 0u, d + sub[j]});
                out.stream8.push_back(H_TAG | E_INFO | (hsub[j] << 22));   // jump patched when the subtree closes
            }
            if (sib_pending_hdr != UINT32_MAX) { sib_hdr_pos[sib_pending_hdr] = (uint32_t)out.stream8.size(); sib_pending_hdr = UINT32_MAX; }
            pos8_hdr[d] = (uint32_t)out.stream8.size();
            emit8(out.stream8, j, false);
            if (big) open_big.back().own_end = (uint32_t)out.stream8.size();
        }
        close_big(out.chunk_node_off[c + 1]);
        out.stream8.push_back(H_TAG | E_CHUNK_END);
        out.chunk8_pre_off.push_back((uint32_t)out.pre8_stream.size());
        uint32_t a = out.dfs2bfs[out.chunk_node_off[c]];
        path.clear();
        for (uint32_t q = a; q != 0;) { q = t.parent[q]; path.push_back(q); }
        for (size_t i = path.size(); i-- > 0;) emit8(out.pre8_stream, path[i], true);
    }
    out.chunk8_body_off.push_back((uint32_t)out.stream8.size());
    out.chunk8_pre_off.push_back((uint32_t)out.pre8_stream.size());
    pos8_at[N] = (uint32_t)out.stream8.size();

    flat_lap("packed stream");
    // ---- tie stream (phase 2 walks it one chunk at a time)
    {
        out.stream_t.clear(); out.chunk_t_off.clear();
        out.stream_t.reserve(out.stream.size());
        std::vector<uint32_t> subd(N, 0);   // dwords of the kept records of j's descendants
        for (uint64_t j = N; j-- > 1;)
            subd[t.parent[j]] += subd[j] + (dropped[j] ? 0u : 2u + (out.stream[rec_off[j]] & 0xFFFFu));
        struct Open { uint32_t info_pos, own_end, dfs_end; };
        std::vector<Open> open;
        auto close = [&](uint32_t next_dfs) {
            while (!open.empty() && open.back().dfs_end <= next_dfs) {
                const Open &b = open.back();
                const uint64_t jump = out.stream_t.size() - b.own_end;
                out.stream_t[b.info_pos + 1] = (out.stream_t[b.info_pos + 1] & 0xFF000000u) | (uint32_t)std::min<uint64_t>(jump, 0xFFFFFFu);
                open.pop_back();
            }
        };
        for (uint32_t c = 0; c < out.n_chunks; c++) {
            out.chunk_t_off.push_back((uint32_t)out.stream_t.size());
            for (uint32_t d = out.chunk_node_off[c]; d < out.chunk_node_off[c + 1]; d++) {
                if ((uint64_t)d + 24 < N) {
                    const uint32_t jn = out.dfs2bfs[d + 24];
                    __builtin_prefetch(&dropped[jn]); __builtin_prefetch(&rec_off[jn]); __builtin_prefetch(&subd[jn]);
                    __builtin_prefetch(&hsub[jn]); __builtin_prefetch(&sub[jn]);
                }
                const uint32_t j = out.dfs2bfs[d];
                close(d);
                if (dropped[j]) continue;
                const uint32_t off = rec_off[j];
                const uint32_t nwords = out.stream[off] & 0xFFFFu;
                if (j != 0 && subd[j] >= T_PRUNE_MIN_DWORDS && hsub[j] <= 255) {
                    open.push_back({(uint32_t)out.stream_t.size(), 0u, d + sub[j]});
                    out.stream_t.push_back(T_INFO_MARK);
                    out.stream_t.push_back(hsub[j] << 24);
                    open.back().own_end = (uint32_t)out.stream_t.size() + 2u + nwords;
                }
                for (uint32_t k = 0; k < 2u + nwords; k++) out.stream_t.push_back(out.stream[off + k]);
            }
            close(UINT32_MAX);   // a jump never leaves its chunk
        }
        out.chunk_t_off.push_back((uint32_t)out.stream_t.size());
    }

    flat_lap("tie stream");
    // Renumber the packed stream's slots by access frequency, hottest first: the
    // kernel keeps the first few in LDS and the cold remainder in a global scratch
    // (slot use is bell-shaped over the index, a handful of slots take ~95 %).  Then
    // convert every header to its final layout (ugp_flatten.hpp): only now is it known
    // which headers touch a cold slot and have to leave the fast path (H_SLOW).
    {
        uint64_t freq[64] = {0};
        for (uint32_t w : out.stream8) {
            if (!(w & H_TAG) || (w & (E_CHUNK_END | E_NOP | E_INFO))) continue;
            uint32_t rs = w & 63u, ws = (w >> 6) & 63u;
            if (rs < RS_BOTTOM) freq[rs]++;
            if (ws != WS_NONE) freq[ws]++;
        }
        uint32_t order[64], remap[64];
        for (uint32_t i = 0; i < 64; i++) order[i] = i;
        std::stable_sort(order, order + out.max_slots, [&](uint32_t a, uint32_t b) { return freq[a] > freq[b]; });
        for (uint32_t i = 0; i < 64; i++) remap[i] = i;
        for (uint32_t i = 0; i < out.max_slots; i++) remap[order[i]] = i;
        out.lds_slots = std::max<uint32_t>(1, std::min<uint32_t>(opt.lds_slots, out.max_slots));
        const uint32_t hot = out.lds_slots;
        auto finalize8 = [&](std::vector<uint32_t> &v) {
            for (uint32_t &w : v) {
                if (!(w & H_TAG)) continue;
                if (w & E_INFO) {   // (first: the jump length overlaps the other intermediate flag bits)
                    w = H_TAG | H_INFO | H_RARE | (w & E_SIB ? H_SIB : 0u) | (((w >> 22) & 0x7Fu) << INFO_HS_SHIFT) | (w & INFO_JUMP_MASK);
                    continue;
                }
                if (w & E_CHUNK_END) { w = H_TAG | H_RARE | H_CHUNK_END; continue; }
                if (w & E_NOP) { w = H_TAG | H_RARE | H_NOP; continue; }
                uint32_t rs = w & 63u, ws = (w >> 6) & 63u;
                uint32_t h = H_TAG;
                bool slow = false;
                if (rs == RS_REG) h |= H_REG;
                else if (rs == RS_BOTTOM) { h |= H_BOTTOM; slow = true; }
                else { rs = remap[rs]; h |= rs << H_RSLOT_SHIFT; slow |= rs >= hot; }
                if (ws != WS_NONE) { ws = remap[ws]; h |= H_STORE | (ws << H_WSLOT_SHIFT); slow |= ws >= hot; }
                if (w & E_SKIPD) h |= H_SKIPD;
                if (w & E_NOSCORE) h |= H_NOSCORE;
                if (w & E_END) h |= H_END;
                if (w & E_FREE) h |= H_FREE;
                if (slow) h |= H_SLOW | H_RARE;
                w = h;
            }
        };
        finalize8(out.stream8);
        finalize8(out.pre8_stream);
    }
    // ---- summaries: the top-level subtrees of every run of super_chunks chunks, as one dense stream ----------
    // Seen from the first node f of such a run, every later node of the run lies in the subtree of f, of a later
    // sibling of f, or of a later sibling of one of f's ancestors: the "top-level" nodes, whose parents are on the
    // root path of f (their D is in the slots once the preamble has been replayed).  A wave far from the tile's
    // samples used to reach them one by one -- evaluate, find the subtree prunable, jump, refill the pipeline:
    // three dependent memory round trips per node, which is what bound the kernel.  The summary lists them
    // back to back ({SUM_A, SUM_B, header copy, mutation words} each) so they are evaluated in the pipelined
    // loop without a single restart; only the survivors' subtrees are walked in the main stream afterwards.
    {
        out.super_chunks = std::max<uint32_t>(1, opt.super_chunks);
        const uint32_t SC = out.super_chunks;
        out.sum8.clear(); out.sum8_off.clear();
        for (uint32_t c0 = 0; c0 < out.n_chunks; c0 += SC) {
            const uint32_t c1 = std::min<uint32_t>(out.n_chunks, c0 + SC);
            out.sum8_off.push_back((uint32_t)out.sum8.size());
            const uint32_t base = out.chunk8_body_off[c0], body_end = out.chunk8_body_off[c1];
            const uint32_t d_end = out.chunk_node_off[c1];
            for (uint32_t d = out.chunk_node_off[c0]; d < d_end;) {
                const uint32_t j = out.dfs2bfs[d];
                const uint32_t d_next = (uint32_t)std::min<uint64_t>((uint64_t)d + sub[j], d_end);
                if (!dropped[j]) {
                    const uint32_t hpos = pos8_hdr[d];
                    uint32_t h = out.stream8[hpos] & ~H_STORE;
                    const uint32_t nw = out.stream[rec_off[j]] & 0xFFFFu;
                    // the end of the subtree inside this run (a subtree that reaches beyond it ends with the run)
                    const uint32_t end_pos = ((uint64_t)d + sub[j] >= d_end ? body_end : pos8_at[d_next]) - base;
                    const bool forced = hsub[j] > INFO_HS_MAX || nw >= 15 || nw > SUM_W_MAX;   // (no test: always walked)
                    if (forced) h |= H_END;
                    out.sum8.push_back(SUM_A | (std::min<uint32_t>(hsub[j], INFO_HS_MAX) << INFO_HS_SHIFT) | (pos8_start[d] - base));
                    out.sum8.push_back(SUM_B | (forced ? SUM_FORCED : 0u) | (std::min<uint32_t>(nw, SUM_W_MAX) << SUM_W_SHIFT) | end_pos);
                    out.sum8.push_back(h);
                    if (!forced) for (uint32_t k = 0; k < nw; k++) out.sum8.push_back(out.stream8[hpos + 1 + k]);
                }
                d = d_next;
            }
        }
        out.sum8_off.push_back((uint32_t)out.sum8.size());
    }
    flat_lap("slot renumbering");
    return UGP_OK;
}

}  // namespace ugp
