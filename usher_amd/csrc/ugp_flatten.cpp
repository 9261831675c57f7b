// ugp_flatten.cpp -- BFS-ordered tree arrays -> DFS record streams (host, C++, multi-threaded).
//
// What the reference does per sample (rebuild the BFS vector, allocate 2N
// vectors, walk parent pointers to the root for every node:
// usher_common.cpp:342-365, usher_mapper.cpp:275-286) is hoisted here and done
// once per tree: the true parent state of every mutation, the descendant-leaf
// counts (Tree::get_num_leaves, mutation_annotated_tree.cpp:866-879), the
// (n_leaves, j) tie rank (usher_mapper.cpp:483-486) and a traversal order whose
// running state fits a log2(N)-deep per-lane stack.
//
// Structure.  Nothing here walks the tree recursively.  Per-node quantities are computed by passes over the
// breadth-first levels (children pull from / push to their parent's row; the nodes of a level are independent),
// the DFS position of every node follows in closed form from the subtree sizes (dfs(c_k) = dfs(p) + 1 + the sizes
// of the earlier siblings), and every stream is written in two steps -- sizes, prefix sum, then the words -- so
// that contiguous DFS segments (32-bit stream) or chunks (packed stream, tie stream, preambles, summaries) are
// independent units of work for the host threads.  The one piece of running state, the allele at every site on
// the current root path, is rebuilt per segment by replaying the root path of its first node.
// The output does not depend on the number of threads (tools/flatten_digest.py).
#include "ugp_flatten.hpp"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <numeric>
#include <thread>

namespace ugp {

// Intermediate encoding of the packed stream while it is being emitted (slot numbers are renumbered by use
// afterwards, and only then can a header be told apart as fast-path or H_SLOW): rslot [5:0], wslot [11:6],
// flags below; pruning records {E_INFO, hs [29:22], E_SIB, hr [20:18], jump [17:0]}.  finalize8() converts to the
// layout of ugp_flatten.hpp.
namespace {
constexpr uint32_t E_SKIPD = 1u << 12, E_NOSCORE = 1u << 13, E_LONG = 1u << 14, E_END = 1u << 16, E_FREE = 1u << 17, E_CHUNK_END = 1u << 18,
                   E_NOP = 1u << 19, E_SIB = 1u << 21, E_INFO = 1u << 30;

inline int nuc_index(uint8_t onehot) {
    switch (onehot) {
        case 1: return 0;
        case 2: return 1;
        case 4: return 2;
        case 8: return 3;
        default: return -1;
    }
}

// Scratch array without initialisation: every element is written by the (parallel) pass that produces it before it is
// read, so the pages are first touched by the threads that fill them.
template <class V>
struct Buf {
    V *p;
    explicit Buf(uint64_t n) : p(static_cast<V *>(malloc(std::max<uint64_t>(1, n) * sizeof(V)))) { if (!p) throw std::bad_alloc(); }
    ~Buf() { free(p); }
    Buf(const Buf &) = delete;
    Buf &operator=(const Buf &) = delete;
    V &operator[](uint64_t i) { return p[i]; }
    const V &operator[](uint64_t i) const { return p[i]; }
    V *data() { return p; }
};

// The error the serial order of checks would have met first: smallest (stage, index).
struct FirstError {
    std::mutex m;
    uint64_t key = UINT64_MAX;
    int code = UGP_OK;
    std::string msg;
    std::atomic<bool> any{false};
    void report(uint64_t stage, uint64_t index, int c, std::string text) {
        std::lock_guard<std::mutex> g(m);
        const uint64_t k = (stage << 56) | (index & ((1ull << 56) - 1));
        if (k < key) { key = k; code = c; msg = std::move(text); }
        any.store(true, std::memory_order_relaxed);
    }
};
}  // namespace

uint64_t Par::exclusive_scan(uint32_t *a, uint64_t n) const {
    std::vector<uint64_t> part(T + 1, 0);
    run(n, [&](uint64_t b, uint64_t e, unsigned tid) { uint64_t s = 0; for (uint64_t i = b; i < e; i++) s += a[i]; part[tid + 1] = s; }, 1u << 16);
    for (unsigned i = 0; i < T; i++) part[i + 1] += part[i];
    run(n, [&](uint64_t b, uint64_t e, unsigned tid) { uint64_t s = part[tid]; for (uint64_t i = b; i < e; i++) { const uint32_t v = a[i]; a[i] = (uint32_t)s; s += v; } }, 1u << 16);
    return part[T];
}

unsigned flatten_threads(const Options &opt) {
    unsigned n = opt.threads;
    if (!n) {
        if (const char *e = getenv("UGP_FLATTEN_THREADS")) n = (unsigned)std::max(1, atoi(e));
        else n = std::min(32u, std::max(1u, std::thread::hardware_concurrency()));
    }
    return std::min(n, 256u);
}

Par flatten_par(const Options &opt) {
    const char *g = getenv("UGP_FLATTEN_GRAIN");
    return Par{flatten_threads(opt), g ? (uint64_t)std::max(1, atoi(g)) : 0u};
}

int flatten(const ugp_tree_desc &t, const Options &opt, FlatMat &out, std::string &err, FlatExtras *extras) {
    const uint64_t N = t.n_nodes;
    if (N == 0 || !t.parent || !t.mut_off) { err = "empty tree or null arrays"; return UGP_ERR_INVALID; }
    if (N >= (1ull << 31)) { err = "more than 2^31 nodes"; return UGP_ERR_UNSUPPORTED; }
    if (t.parent[0] != UINT32_MAX) { err = "parent[0] must be UINT32_MAX (root first, BFS order)"; return UGP_ERR_INVALID; }
    const uint64_t M = t.mut_off[N];
    if (M && (!t.mut_pos || !t.mut_ref || !t.mut_nuc)) { err = "null mutation arrays"; return UGP_ERR_INVALID; }

    const Par par = flatten_par(opt);
    const unsigned T = par.T;
    FirstError ferr;
    auto failed = [&]() -> int { err = ferr.msg; return ferr.code; };

    const bool verbose = getenv("UGP_FLATTEN_VERBOSE") != nullptr;
    auto t_prev = std::chrono::steady_clock::now();
    auto flat_lap = [&](const char *what) {
        if (!verbose) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[ugp flatten] %-28s %8.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_prev).count());
        t_prev = now;
    };

    // ---- children (a breadth-first expansion lists them contiguously, parents in ascending order) -----------
    std::atomic<bool> bad_parent{false}, not_monotone{false};
    par.run(N, [&](uint64_t b, uint64_t e, unsigned) {
        bool bad = false, nm = false;
        for (uint64_t j = std::max<uint64_t>(b, 1); j < e; j++) {
            bad |= t.parent[j] >= j;
            nm |= j > 1 && t.parent[j] < t.parent[j - 1];
        }
        if (bad) bad_parent = true;
        if (nm) not_monotone = true;
    });
    if (bad_parent) { err = "parent[j] must be < j (BFS order)"; return UGP_ERR_INVALID; }
    const bool bfs_levels = !not_monotone;   // parent[] ascending: levels are index ranges and children[k] = k + 1 before reordering
    Buf<uint32_t> child_off(N + 1), children(N);
    if (bfs_levels) {
        par.run(N, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t j = std::max<uint64_t>(b, 1); j < e; j++) {
                children[j - 1] = (uint32_t)j;
                const uint64_t lo = (j == 1) ? 0 : (uint64_t)t.parent[j - 1] + 1;
                for (uint64_t p = lo; p <= t.parent[j]; p++) child_off[p] = (uint32_t)(j - 1);
            }
        });
        for (uint64_t p = (N > 1 ? (uint64_t)t.parent[N - 1] + 1 : 0); p <= N; p++) child_off[p] = (uint32_t)(N - 1);
    } else {
        memset(child_off.data(), 0, (N + 1) * sizeof(uint32_t));
        for (uint64_t j = 1; j < N; j++) child_off[t.parent[j] + 1]++;
        for (uint64_t j = 0; j < N; j++) child_off[j + 1] += child_off[j];
        std::vector<uint32_t> fill(child_off.data(), child_off.data() + N);
        for (uint64_t j = 1; j < N; j++) children[fill[t.parent[j]]++] = (uint32_t)j;
    }
    // levels: [level_off[l], level_off[l + 1]) (index ranges when bfs_levels; otherwise one pseudo-level per node,
    // which makes the level passes below plain serial loops in ascending / descending node order)
    std::vector<uint64_t> level_off;
    if (bfs_levels) {
        level_off.push_back(0);
        uint64_t lo = 0, hi = 1;
        while (lo < hi) {
            level_off.push_back(hi);
            const uint64_t nlo = hi, nhi = (uint64_t)child_off[hi] + 1;   // children of [lo, hi) are nodes child_off[lo]+1 .. child_off[hi]
            lo = nlo; hi = nhi;
        }
    }
    auto top_down = [&](auto fn) {      // fn(p, tid): p's row is final, writes the rows of its children
        if (!bfs_levels) { for (uint64_t p = 0; p < N; p++) fn(p, 0u); return; }
        for (size_t l = 0; l + 1 < level_off.size(); l++) {
            const uint64_t b0 = level_off[l], n = level_off[l + 1] - b0;
            par.run(n, [&](uint64_t b, uint64_t e, unsigned tid) { for (uint64_t p = b0 + b; p < b0 + e; p++) fn(p, tid); }, 4096);
        }
    };
    auto bottom_up = [&](auto fn) {     // fn(p, tid): the rows of p's children are final, writes p's row
        if (!bfs_levels) { for (uint64_t p = N; p-- > 0;) fn(p, 0u); return; }
        for (size_t l = level_off.size() - 1; l-- > 0;) {
            const uint64_t b0 = level_off[l], n = level_off[l + 1] - b0;
            par.run(n, [&](uint64_t b, uint64_t e, unsigned tid) { for (uint64_t p = b0 + b; p < b0 + e; p++) fn(p, tid); }, 4096);
        }
    };
    flat_lap("children");

    // ---- own mutation words, never-eligible leaves --------------------------------------------------------------
    Buf<uint32_t> nw(N);            // non-masked mutations of the node
    Buf<uint8_t> rev(N);            // ... of which hit a site that is NOT at the reference base in the parent (saturating at 255; filled by the DFS emission, which tracks the states)
    Buf<uint8_t> dropped(N);        // leaf without mutation words: never eligible, no descendants (packed / tie streams skip it)
    par.run(N, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t j = b; j < e; j++) {
            uint32_t c = 0;
            for (uint64_t i = t.mut_off[j]; i < t.mut_off[j + 1] && i < M; i++) c += t.mut_pos[i] >= 0;
            nw[j] = c;
            rev[j] = 0;
            dropped[j] = j != 0 && child_off[j + 1] == child_off[j] && c == 0;
        }
    });
    // ---- subtree sizes, descendant-leaf counts, pruning bounds (one bottom-up pass) ------------------------------
    // hsub[j] = largest number of mutation words on a path j -> descendant, without j's own (hdown = nw + hsub: no
    // node of j's subtree, j included, costs less than D(parent(j)) - hdown[j]); subw / subd = packed-stream words /
    // tie-stream dwords of j's descendants (without pruning records)
    UVec<uint32_t> sub(N);
    Buf<uint32_t> leaves(N), hsub(N), subw(N), subd(N);
    Buf<uint8_t> hrev(N);   // largest number of second hits (rev) on a path j -> descendant, without j's own (saturating at 255; second pass below)
    bottom_up([&](uint64_t p, unsigned) {
        uint32_t s = 1, lv = 0, h = 0, w = 0, d = 0;
        for (uint32_t k = child_off[p]; k < child_off[p + 1]; k++) {
            const uint32_t c = children[k];
            s += sub[c]; lv += leaves[c];
            h = std::max(h, nw[c] + hsub[c]);
            w += subw[c] + (dropped[c] ? 0u : 1u + nw[c]);
            d += subd[c] + (dropped[c] ? 0u : 2u + nw[c]);
        }
        sub[p] = s; leaves[p] = (child_off[p + 1] == child_off[p]) ? 1u : lv; hsub[p] = h; subw[p] = w; subd[p] = d;
    });
    // largest subtree last; the others by descending hdown (the sibling pruning records rely on it)
    par.run(N, [&](uint64_t b0, uint64_t e0, unsigned) {
        for (uint64_t j = b0; j < e0; j++) {
            const uint32_t b = child_off[j], e = child_off[j + 1];
            if (e - b < 2) continue;
            uint32_t best = b;
            for (uint32_t k = b + 1; k < e; k++) if (sub[children[k]] > sub[children[best]]) best = k;
            const uint32_t c = children[best];
            for (uint32_t k = best; k + 1 < e; k++) children[k] = children[k + 1];
            children[e - 1] = c;
            if (e - b > 2) std::stable_sort(children.data() + b, children.data() + e - 1, [&](uint32_t x, uint32_t y) { return nw[x] + hsub[x] > nw[y] + hsub[y]; });
        }
    });
    flat_lap("subtree sizes");

    // ---- tie rank: ascending (n_leaves, j) -------------------------------------------------------------------------
    // (stable LSD radix sort of j by leaves[j], 11 bits per pass; as many passes as the largest count needs)
    out.rank2bfs.resize(N);
    Buf<uint32_t> rank(N);
    {
        UVec<uint32_t> tmp(N);
        uint32_t *a = out.rank2bfs.data(), *bq = tmp.data();
        par.run(N, [&](uint64_t b, uint64_t e, unsigned) { for (uint64_t i = b; i < e; i++) a[i] = (uint32_t)i; });
        const uint32_t max_leaves = leaves[0];
        std::vector<uint64_t> hist((size_t)T * 2048);
        for (uint32_t shift = 0; shift < 32 && (max_leaves >> shift) != 0; shift += 11) {
            std::fill(hist.begin(), hist.end(), 0);
            par.run(N, [&](uint64_t b, uint64_t e, unsigned tid) {
                uint64_t *h = &hist[(size_t)tid * 2048];
                for (uint64_t i = b; i < e; i++) h[(leaves[a[i]] >> shift) & 2047u]++;
            }, 1u << 16);
            uint64_t run = 0;
            for (uint32_t bk = 0; bk < 2048; bk++)
                for (unsigned th = 0; th < T; th++) { const uint64_t c = hist[(size_t)th * 2048 + bk]; hist[(size_t)th * 2048 + bk] = run; run += c; }
            par.run(N, [&](uint64_t b, uint64_t e, unsigned tid) {
                uint64_t *h = &hist[(size_t)tid * 2048];
                for (uint64_t i = b; i < e; i++) { const uint32_t j = a[i]; bq[h[(leaves[j] >> shift) & 2047u]++] = j; }
            }, 1u << 16);
            std::swap(a, bq);
        }
        if (a != out.rank2bfs.data()) out.rank2bfs.swap(tmp);
        const uint32_t *r2b = out.rank2bfs.data();
        par.run(N, [&](uint64_t b, uint64_t e, unsigned) { for (uint64_t r = b; r < e; r++) rank[r2b[r]] = (uint32_t)r; });
    }
    flat_lap("tie rank");

    // ---- sites -------------------------------------------------------------------------------------------------------
    int32_t max_pos = 0;
    {
        std::vector<int32_t> mx(T, 0);
        par.run(M, [&](uint64_t b, uint64_t e, unsigned tid) { int32_t m = 0; for (uint64_t i = b; i < e; i++) m = std::max(m, t.mut_pos[i]); mx[tid] = m; }, 1u << 16);
        for (int32_t m : mx) max_pos = std::max(max_pos, m);
    }
    out.max_pos = (uint32_t)max_pos;
    uint64_t n_real = 0;
    {
        // per-thread "reference base seen at this position" tables, merged afterwards
        const Par spar{(uint64_t)max_pos > (1ull << 26) ? 1u : T, par.grain};
        std::vector<std::vector<uint8_t>> seen(spar.T);
        std::vector<uint64_t> cnt(spar.T, 0);
        spar.run(M, [&](uint64_t b, uint64_t e, unsigned tid) {
            std::vector<uint8_t> &sv = seen[tid];
            sv.assign((size_t)max_pos + 1, 0);
            uint64_t c = 0;
            for (uint64_t i = b; i < e; i++) {
                const int32_t p = t.mut_pos[i];
                if (p < 0) continue;
                c++;
                if (nuc_index(t.mut_nuc[i]) < 0) { ferr.report(0, i, UGP_ERR_UNSUPPORTED, "tree mutation allele is not one-hot (ambiguous MAT alleles are unsupported)"); continue; }
                if (nuc_index(t.mut_ref[i]) < 0) { ferr.report(0, i, UGP_ERR_UNSUPPORTED, "tree mutation reference base is not one-hot"); continue; }
                if (sv[p] == 0) sv[p] = t.mut_ref[i];
                else if (sv[p] != t.mut_ref[i]) ferr.report(1, (uint64_t)p, UGP_ERR_UNSUPPORTED, "tree mutations disagree on the reference base at position " + std::to_string(p));
            }
            cnt[tid] = c;
        }, 1u << 16);
        for (uint64_t c : cnt) n_real += c;
        out.pos2site.assign((size_t)max_pos + 1, -1);
        out.site_ref.clear();
        for (int32_t p = 0; p <= max_pos; p++) {   // sites numbered by position: a tile row index grows with the genome coordinate
            uint8_t r = 0;
            for (auto &sv : seen) {
                if (sv.empty() || sv[p] == 0) continue;
                if (r == 0) r = sv[p];
                else if (r != sv[p]) ferr.report(1, (uint64_t)p, UGP_ERR_UNSUPPORTED, "tree mutations disagree on the reference base at position " + std::to_string(p));
            }
            if (r == 0) continue;
            out.pos2site[p] = (int32_t)out.site_ref.size();
            out.site_ref.push_back(r);
        }
    }
    if (ferr.any) return failed();
    out.n_sites = out.site_ref.size();
    if (out.n_sites > MAX_SITES) { err = "more than 2^22 mutated positions"; return UGP_ERR_UNSUPPORTED; }
    out.n_nodes = N;
    out.n_muts = n_real;
    flat_lap("sites");

    // ---- DFS positions, saved-D slots, effective children (one top-down pass) -------------------------------------------
    // slot[c] = slot[p] for p's last child, slot[p] + 1 for the others: the number of non-last-child edges on the root
    // path, <= log2 N because the largest subtree goes last
    UVec<uint32_t> dfsidx(N);
    Buf<uint32_t> pathm(N), eff_children(N), last_eff(N), suffix_h(N), big_after(N);   // (suffix_h / big_after: non-last effective children only)
    Buf<uint8_t> slot(N), first_child(N), first_eff(N), suffix_hr(N);   // (suffix_hr: like suffix_h, over second hits)
    slot[0] = 0; first_child[0] = 0; first_eff[0] = 0;
    struct alignas(64) Maxima { uint32_t path = 0, slot = 0; };   // one cache line per thread
    std::vector<Maxima> mx(T);
    dfsidx[0] = 0; pathm[0] = nw[0]; mx[0].path = nw[0];
    top_down([&](uint64_t p, unsigned tid) {
        const uint32_t b = child_off[p], e = child_off[p + 1];
        if (e - b >= 2) mx[tid].slot = std::max<uint32_t>(mx[tid].slot, slot[p] + 1u);
        uint32_t d = dfsidx[p] + 1, n_eff = 0, last = UINT32_MAX, mp = 0;
        for (uint32_t k = b; k < e; k++) {
            const uint32_t c = children[k];
            dfsidx[c] = d; d += sub[c];
            const uint32_t s = (k + 1 == e) ? slot[p] : slot[p] + 1u;
            if (s >= MAX_SLOTS) { ferr.report(2, dfsidx[c], UGP_ERR_UNSUPPORTED, "D stack deeper than MAX_SLOTS"); }
            slot[c] = (uint8_t)std::min<uint32_t>(s, 255);
            first_child[c] = k == b;
            pathm[c] = pathm[p] + nw[c];
            mp = std::max(mp, pathm[c]);
            first_eff[c] = !dropped[c] && n_eff == 0;
            if (!dropped[c]) { n_eff++; last = c; }
        }
        if (mp > mx[tid].path) mx[tid].path = mp;
        eff_children[p] = n_eff;
        last_eff[p] = last;
        // sibling records: for a non-last effective child c, suffix_h[c] = max hdown over c and the non-last effective
        // siblings after it, big_after[c] = how many of those later siblings carry a pruning record of their own
        uint32_t run_h = 0, run_big = 0;
        for (uint32_t k = e; k-- > b;) {
            const uint32_t c = children[k];
            if (dropped[c] || c == last) continue;
            big_after[c] = run_big;
            run_h = std::max(run_h, nw[c] + hsub[c]);
            suffix_h[c] = run_h;
            if (subw[c] >= opt.prune_min_words && hsub[c] <= INFO_HS_MAX) run_big++;
        }
    });
    if (ferr.any) return failed();
    out.max_path_muts = 0; out.max_slots = 1;
    for (const Maxima &m : mx) { out.max_path_muts = std::max(out.max_path_muts, m.path); out.max_slots = std::max(out.max_slots, m.slot); }
    out.dfs2bfs.resize(N);
    par.run(N, [&](uint64_t b, uint64_t e, unsigned) { for (uint64_t j = b; j < e; j++) out.dfs2bfs[dfsidx[j]] = (uint32_t)j; });
    const uint32_t *d2b = out.dfs2bfs.data();
    flat_lap("DFS order");

    // ---- 32-bit stream -------------------------------------------------------------------------------------------------
    Buf<uint32_t> rec_off_d(N + 1);   // by DFS index: dword offset of the node's record
    par.run(N, [&](uint64_t b, uint64_t e, unsigned) { for (uint64_t d = b; d < e; d++) rec_off_d[d] = 2u + nw[d2b[d]]; });
    const uint64_t total = par.exclusive_scan(rec_off_d.data(), N);
    if (total >= (1ull << 32)) { err = "tree too large: the record stream exceeds 2^32 dwords"; return UGP_ERR_UNSUPPORTED; }
    rec_off_d[N] = (uint32_t)total;
    out.stream.resize(total + 4);   // (4 dwords of padding behind chunk_body_off[n_chunks]: k_descend fetches a record's first words before it knows its length)
    for (uint64_t i = total; i < total + 4; i++) out.stream[i] = 0;
    Buf<uint8_t> node_masked(N);   // by BFS index: carries a masked mutation (non-root)
    std::atomic<bool> mask_not_first{false};
    par.run(N, [&](uint64_t d0, uint64_t d1, unsigned) {
        if (d0 >= d1) return;
        std::vector<uint8_t> cur(out.n_sites);   // running allele index per site on the current root path
        for (uint64_t s = 0; s < out.n_sites; s++) cur[s] = (uint8_t)nuc_index(out.site_ref[s]);
        struct Undo { uint32_t site; uint8_t old; };
        struct Frame { uint64_t dfs_end; uint32_t undo_mark; };
        std::vector<Undo> undo;
        std::vector<Frame> stack;
        std::vector<uint32_t> path;
        for (uint32_t q = d2b[d0]; q != 0;) { q = t.parent[q]; path.push_back(q); }
        for (size_t i = path.size(); i-- > 0;) {   // replay the root path of the segment's first node
            const uint32_t a = path[i];
            stack.push_back({(uint64_t)dfsidx[a] + sub[a], (uint32_t)undo.size()});
            for (uint64_t k = t.mut_off[a]; k < t.mut_off[a + 1] && k < M; k++) {
                const int32_t p = t.mut_pos[k];
                if (p < 0) continue;
                const uint32_t site = (uint32_t)out.pos2site[p];
                undo.push_back({site, cur[site]});
                cur[site] = (uint8_t)nuc_index(t.mut_nuc[k]);
            }
        }
        bool mnf = false;
        for (uint64_t d = d0; d < d1; d++) {
            while (!stack.empty() && stack.back().dfs_end <= d) {   // leaving subtrees: restore the running state
                const uint32_t mark = stack.back().undo_mark;
                while (undo.size() > mark) { cur[undo.back().site] = undo.back().old; undo.pop_back(); }
                stack.pop_back();
            }
            const uint32_t j = d2b[d];
            const uint64_t b = t.mut_off[j], e = t.mut_off[j + 1];
            if (e < b || e > M) { ferr.report(3, d, UGP_ERR_INVALID, "mut_off is not monotone"); return; }
            const uint32_t nch = child_off[j + 1] - child_off[j];
            const bool root = (j == 0);
            const uint32_t rslot = root ? RS_BOTTOM : (first_child[j] ? RS_REG : slot[t.parent[j]]);
            const uint32_t wslot = nch >= 2 ? slot[j] : WS_NONE;
            uint32_t w0 = 0;
            if (nch == 0) w0 |= F_LEAF;
            if (root) w0 |= F_ROOT;
            uint32_t *rec = &out.stream[rec_off_d[d]];
            stack.push_back({d + sub[j], (uint32_t)undo.size()});
            uint32_t nwords = 0, second = 0;
            bool masked = false;
            int32_t last_pos = -1;
            for (uint64_t i = b; i < e; i++) {
                const int32_t p = t.mut_pos[i];
                if (p < 0) { if (nwords > 0 && !root) mnf = true; masked = true; continue; }
                if (p == last_pos) { ferr.report(3, d, UGP_ERR_UNSUPPORTED, "node carries two mutations at position " + std::to_string(p)); return; }
                last_pos = p;   // (adjacent duplicates; the loader keeps lists sorted)
                const uint32_t site = (uint32_t)out.pos2site[p];
                const uint32_t mi = (uint32_t)nuc_index(t.mut_nuc[i]), pi = cur[site], ri = (uint32_t)nuc_index(out.site_ref[site]);
                uint32_t w = site | (mi << 22) | (pi << 24) | (ri << 26);
                second += pi != ri;
                if (masked && !root) w |= M_AFTER_MASK;
                rec[2 + nwords] = w;
                undo.push_back({site, cur[site]});
                cur[site] = (uint8_t)mi;
                nwords++;
            }
            if (nwords > MAX_NODE_MUTS) { ferr.report(3, d, UGP_ERR_UNSUPPORTED, "node with more than 65534 mutations"); return; }
            node_masked[j] = masked && !root;
            rev[j] = (uint8_t)std::min<uint32_t>(second, 255);
            if (masked && !root) w0 |= F_MASKED;   // root: masked mutations are inert (usher_mapper.cpp:266-269, 309-311, 401-403)
            rec[0] = w0 | nwords | (rslot << 16) | (wslot << 22);
            rec[1] = rank[j] << 1;
        }
        if (mnf) mask_not_first = true;
    }, 4096);
    if (ferr.any) return failed();
    out.mask_not_first = mask_not_first;
    flat_lap("DFS emission");

    // ---- second pruning bound (ugp_flatten.hpp): hrev[j] = largest number of second hits on a path j -> descendant, and the
    // same over the later non-last siblings (sibling records).  Needs the parent states, hence after the emission.
    // B(n, s) of k_best8 is kept in 8 bits per sample and never exceeds the mutations on a root path: on a tree deeper than
    // that the records say "not available".
    const bool second_bound = opt.second_bound && out.max_path_muts <= 255;
    bottom_up([&](uint64_t p, unsigned) {
        uint32_t hr = 0;
        for (uint32_t k = child_off[p]; k < child_off[p + 1]; k++) { const uint32_t c = children[k]; hr = std::max<uint32_t>(hr, (uint32_t)rev[c] + hrev[c]); }
        hrev[p] = second_bound ? (uint8_t)std::min<uint32_t>(hr, 255) : 255;
    });
    par.run(N, [&](uint64_t b0, uint64_t e0, unsigned) {
        for (uint64_t p = b0; p < e0; p++) {
            uint32_t run_hr = 0;
            for (uint32_t k = child_off[p + 1]; k-- > child_off[p];) {
                const uint32_t c = children[k];
                if (dropped[c] || c == last_eff[p]) continue;
                run_hr = std::max<uint32_t>(run_hr, (uint32_t)rev[c] + hrev[c]);
                suffix_hr[c] = (uint8_t)std::min<uint32_t>(run_hr, 255);
            }
        }
    }, 4096);
    flat_lap("second bound");

    // ---- chunks: equal dword budgets, cut at node boundaries ---------------------------------------------------------
    // ~300 nodes per chunk on a 10M-node tree: chunks are the granule of the phase-1 minima (short phase-2
    // re-walks) and of the work units (8 / 16 chunks each outside / inside a tile's own region)
    const uint64_t chunk_nodes = opt.chunk_nodes ? opt.chunk_nodes : std::max<uint64_t>(128, N / 32768);
    const uint64_t want_chunks = std::max<uint64_t>(1, (N + chunk_nodes - 1) / chunk_nodes);
    out.chunk_body_off.clear(); out.chunk_node_off.clear();
    {
        uint64_t next_cut = 0;
        for (uint64_t c = 0; c < want_chunks;) {   // the first node whose record starts at or behind the cut
            const uint64_t d = std::lower_bound(rec_off_d.data(), rec_off_d.data() + N, (uint32_t)std::min<uint64_t>(next_cut, UINT32_MAX)) - rec_off_d.data();
            if (d >= N || next_cut > UINT32_MAX) break;
            const uint32_t off = rec_off_d[d];
            out.chunk_body_off.push_back(off);
            out.chunk_node_off.push_back((uint32_t)d);
            c++;
            next_cut = total * c / want_chunks;
            if (next_cut <= off) next_cut = (uint64_t)off + 1;
        }
        out.chunk_body_off.push_back((uint32_t)total);
        out.chunk_node_off.push_back((uint32_t)N);
    }
    out.n_chunks = (uint32_t)out.chunk_body_off.size() - 1;
    const uint32_t NC = out.n_chunks;
    const uint32_t *cno = out.chunk_node_off.data();
    std::vector<uint8_t> chunk_start(N + 1, 0);   // by DFS index (N: the end of the last chunk)
    for (uint32_t c = 0; c <= NC; c++) chunk_start[cno[c]] = 1;
    flat_lap("chunks");

    // ---- preambles: the root path of each chunk's first node, replayed without scoring (both encodings) ----------------
    // the packed form of a node's record; `preamble` copies read D(parent) from the register and never score
    auto emit8 = [&](uint32_t *dst, uint32_t j, bool preamble) -> uint32_t {
        const uint32_t *rec = &out.stream[rec_off_d[dfsidx[j]]];
        const uint32_t nwords = rec[0] & 0xFFFFu;
        const bool root = (j == 0);
        const uint32_t nch = child_off[j + 1] - child_off[j];
        // (the root reads "the previous node's D" too: k_best8 starts every unit with D = D(bottom), and the root is the
        // first record of any unit that contains or replays it)
        const uint32_t rslot = (root || preamble || first_eff[j]) ? RS_REG : slot[t.parent[j]];
        const uint32_t wslot = (eff_children[j] >= 2) ? slot[j] : WS_NONE;
        uint32_t h = H_TAG | rslot | (wslot << 6);
        if (eff_children[j] == 0 && !root) h |= E_SKIPD;
        if (preamble || root || node_masked[j]) h |= E_NOSCORE;
        if (nwords == 0) h |= E_END;
        if (nwords > 15) h |= E_LONG;   // (the 4-bit counters of k_best8 overflow: the node takes the general step, which spills them at every M_FLUSH)
        if (!root && nch > 0 && nwords == 0 && !node_masked[j]) h |= E_FREE;
        uint32_t n = 0;
        dst[n++] = h;
        for (uint32_t k = 0; k < nwords; k++) {
            uint32_t w = rec[2 + k] & 0x0FFFFFFFu;   // site, mutated / parent-state / reference allele
            if (k + 1 == nwords) w |= M_END;
            else if ((k + 1) % 15 == 0) w |= M_FLUSH;
            dst[n++] = w;
        }
        // the root scores through a pseudo-node right behind its D record: cost = D(parent) = D(root)
        if (root && !preamble) dst[n++] = H_TAG | RS_REG | (WS_NONE << 6) | E_SKIPD | E_FREE | E_END;
        return n;
    };
    out.chunk_pre_off.assign(NC + 1, 0);
    par.run(NC, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t c = b; c < e; c++) {
            uint32_t l32 = 0;
            for (uint32_t q = d2b[cno[c]]; q != 0;) { q = t.parent[q]; l32 += 2u + nw[q]; }
            out.chunk_pre_off[c] = l32;
        }
    }, 64);
    out.pre_stream.resize(par.exclusive_scan(out.chunk_pre_off.data(), NC));
    out.chunk_pre_off[NC] = (uint32_t)out.pre_stream.size();
    par.run(NC, [&](uint64_t b, uint64_t e, unsigned) {
        std::vector<uint32_t> path;
        for (uint64_t c = b; c < e; c++) {
            path.clear();
            for (uint32_t q = d2b[cno[c]]; q != 0;) { q = t.parent[q]; path.push_back(q); }
            uint32_t *p32 = out.pre_stream.data() + out.chunk_pre_off[c];
            for (size_t i = path.size(); i-- > 0;) {
                const uint32_t q = path[i];
                const uint32_t *rec = &out.stream[rec_off_d[dfsidx[q]]];
                const uint32_t nwords = rec[0] & 0xFFFFu;
                *p32++ = (rec[0] & ~(63u << 16)) | ((q == 0 ? RS_BOTTOM : RS_REG) << 16) | F_NOSCORE;
                *p32++ = rec[1];
                for (uint32_t k = 0; k < nwords; k++) *p32++ = rec[2 + k];
            }
        }
    }, 64);
    flat_lap("preambles");

    // ---- packed stream (stream8) on the effective tree ---------------------------------------------------------------------
    // Pruning records: a node j != root whose descendants occupy >= prune_min_words words carries {E_INFO, hsub, jump}
    // in front of its header; a non-last effective child with >= 1 later non-last sibling that carries one gets a
    // sibling record {E_INFO | E_SIB, suffix_h, jump to the parent's last effective child} in front of that.
    const bool sib_on = opt.sibling_records;
    auto is_big = [&](uint32_t j) { return j != 0 && subw[j] >= opt.prune_min_words && hsub[j] <= INFO_HS_MAX; };
    auto has_sib = [&](uint32_t j) { return sib_on && j != 0 && last_eff[t.parent[j]] != j && big_after[j] >= 1 && suffix_h[j] <= INFO_HS_MAX; };
    // by DFS index: stream position when the node is reached, of its first word behind a sibling record, of its header
    Buf<uint32_t> pos8_at(N + 1), pos8_start(N), pos8_hdr(N);
    out.chunk8_body_off.assign(NC + 1, 0);
    par.run(NC, [&](uint64_t b, uint64_t e, unsigned) {     // sizes: positions relative to the chunk's first word
        for (uint64_t c = b; c < e; c++) {
            uint32_t pos = 0;
            for (uint32_t d = cno[c]; d < cno[c + 1]; d++) {
                const uint32_t j = d2b[d];
                pos8_at[d] = pos;
                if (dropped[j]) { pos8_start[d] = 0; pos8_hdr[d] = 0; continue; }
                if (has_sib(j)) pos++;
                pos8_start[d] = pos;
                if (is_big(j)) pos++;
                pos8_hdr[d] = pos;
                pos += 1u + nw[j] + (j == 0 ? 1u : 0u);
            }
            out.chunk8_body_off[c] = pos + 1;   // + the chunk-end word
        }
    }, 16);
    const uint64_t total8 = par.exclusive_scan(out.chunk8_body_off.data(), NC);
    if (total8 >= (1ull << 32)) { err = "tree too large: the packed stream exceeds 2^32 words"; return UGP_ERR_UNSUPPORTED; }
    out.chunk8_body_off[NC] = (uint32_t)total8;
    out.max_chunk8_words = 0;
    for (uint32_t c = 0; c < NC; c++) out.max_chunk8_words = std::max(out.max_chunk8_words, out.chunk8_body_off[c + 1] - out.chunk8_body_off[c]);
    pos8_at[N] = (uint32_t)total8;
    par.run(NC, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t c = b; c < e; c++) {
            const uint32_t base = out.chunk8_body_off[c];
            for (uint32_t d = cno[c]; d < cno[c + 1]; d++) {
                pos8_at[d] += base;
                if (!dropped[d2b[d]]) { pos8_start[d] += base; pos8_hdr[d] += base; }
            }
        }
    }, 16);
    if (opt.keep_node_pos8) {
        out.node_pos8.assign(pos8_at.data(), pos8_at.data() + N + 1);
        out.rank_dfs.resize(N);
        par.run(N, [&](uint64_t b, uint64_t e, unsigned) { for (uint64_t d = b; d < e; d++) out.rank_dfs[d] = rank[d2b[d]]; });
    } else { out.node_pos8.clear(); out.rank_dfs.clear(); }
    if (opt.keep_update_maps) {
        out.hdr8_of_bfs.resize(N); out.rec_of_bfs.resize(N);
        par.run(N, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t d = b; d < e; d++) { const uint32_t j = d2b[d]; out.hdr8_of_bfs[j] = dropped[j] ? UINT32_MAX : pos8_hdr[d]; out.rec_of_bfs[j] = rec_off_d[d]; }
        });
    } else { out.hdr8_of_bfs.clear(); out.rec_of_bfs.clear(); }
    out.stream8.resize(total8);
    par.run(NC, [&](uint64_t b, uint64_t e, unsigned) {
        uint32_t *s8 = out.stream8.data();
        for (uint64_t c = b; c < e; c++) {
            for (uint32_t d = cno[c]; d < cno[c + 1]; d++) {
                const uint32_t j = d2b[d];
                if (dropped[j]) continue;
                if (has_sib(j)) {   // jump: from this node's header to the first word of the parent's last effective child
                    const uint64_t jump = (uint64_t)pos8_at[dfsidx[last_eff[t.parent[j]]]] - pos8_hdr[d];
                    const uint32_t hr = std::min<uint32_t>(suffix_hr[j], INFO_HR_NONE);
                    s8[pos8_at[d]] = (jump <= INFO_JUMP_MASK && jump > 0) ? (H_TAG | E_INFO | E_SIB | (suffix_h[j] << 22) | (hr << INFO_HR_SHIFT) | (uint32_t)jump) : (H_TAG | E_NOP);
                }
                if (is_big(j)) {    // jump: the words of the descendants (a subtree that ends with a chunk lands on its chunk-end word)
                    const uint64_t d_end = (uint64_t)d + sub[j];
                    const uint64_t own_end = (uint64_t)pos8_hdr[d] + 1u + nw[j];
                    const uint64_t jump = (uint64_t)pos8_at[d_end] - (chunk_start[d_end] ? 1u : 0u) - own_end;
                    const uint32_t hr = std::min<uint32_t>(hrev[j], INFO_HR_NONE);
                    s8[pos8_start[d]] = (jump <= INFO_JUMP_MASK && jump > 0) ? (H_TAG | E_INFO | (hsub[j] << 22) | (hr << INFO_HR_SHIFT) | (uint32_t)jump) : (H_TAG | E_NOP);
                }
                emit8(s8 + pos8_hdr[d], j, false);
            }
            s8[out.chunk8_body_off[c + 1] - 1] = H_TAG | E_CHUNK_END;
        }
    }, 16);
    // Packed preambles, with a pruning record in front of every path node below the root: {E_INFO, hs, hr, jump} with
    // jump = where the body goes on behind the node's subtree, relative to the chunk's first body word (capped: "beyond
    // any unit").  A unit far from the tile's samples is usually inside a subtree that the second bound already rules out
    // a few nodes below the point where its root path leaves the samples' paths: the replay stops there and the body
    // starts behind that subtree -- most of the time behind the whole unit.  hs = PRE_HS_NONE: hsub does not fit the field.
    auto pre_rec = [&](uint32_t q) -> bool { return q != 0 && (hsub[q] < PRE_HS_NONE || hrev[q] < INFO_HR_NONE); };
    out.chunk8_pre_off.assign(NC + 1, 0);
    par.run(NC, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t c = b; c < e; c++) {
            uint32_t l8 = 0;
            for (uint32_t q = d2b[cno[c]]; q != 0;) { q = t.parent[q]; l8 += 1u + nw[q] + (pre_rec(q) ? 1u : 0u); }
            out.chunk8_pre_off[c] = l8;
        }
    }, 64);
    out.pre8_stream.resize(par.exclusive_scan(out.chunk8_pre_off.data(), NC));
    out.chunk8_pre_off[NC] = (uint32_t)out.pre8_stream.size();
    par.run(NC, [&](uint64_t b, uint64_t e, unsigned) {
        std::vector<uint32_t> path;
        for (uint64_t c = b; c < e; c++) {
            path.clear();
            for (uint32_t q = d2b[cno[c]]; q != 0;) { q = t.parent[q]; path.push_back(q); }
            uint32_t *p8 = out.pre8_stream.data() + out.chunk8_pre_off[c];
            for (size_t i = path.size(); i-- > 0;) {
                const uint32_t q = path[i];
                if (pre_rec(q)) {
                    const uint64_t rel = (uint64_t)pos8_at[(uint64_t)dfsidx[q] + sub[q]] - out.chunk8_body_off[c];
                    *p8++ = H_TAG | E_INFO | (std::min<uint32_t>(hsub[q], PRE_HS_NONE) << 22) | (std::min<uint32_t>(hrev[q], INFO_HR_NONE) << INFO_HR_SHIFT) |
                            (uint32_t)std::min<uint64_t>(rel, INFO_JUMP_MASK);
                }
                p8 += emit8(p8, q, true);
            }
        }
    }, 64);
    flat_lap("packed stream");

    // ---- third pruning bound (round 5; ugp_flatten.hpp "B3"): posting lists of the mutation events by (site, mutated allele).
    // An event = one mutation word of the packed body; it raises, for every node of its node's subtree (the node included), the
    // number of mutations of that (site, allele) on the root path -- i.e. over the word range [header of the node, end of its
    // descendants), here as a range of blocks of B3_BLOCK_WORDS words.  Listed in depth-first order (deterministic).
    out.b3_group_off.clear(); out.b3_events.clear();
    // (max_path_muts: the table kernel's per-block end counter is 16 bits wide, bounded by the mutation words of one root path + 16; the tables
    // themselves saturate at 255 = "no bound".  The packed walk -- the only reader -- needs max_path_muts < 0x7F7F anyway.)
    if (opt.keep_b3_events && out.n_sites && total8 > 0 && total8 < (1ull << 32) - 2 * B3_GROUP_BLOCKS * B3_BLOCK_WORDS && out.max_path_muts < 0x7F7Fu) {
        const uint32_t ng = b3_blocks(total8) >> B3_GROUP_SHIFT;
        out.b3_group_off.assign((size_t)4 * (ng + 1), 0);
        uint32_t *off[4];
        for (int k = 0; k < 4; k++) off[k] = out.b3_group_off.data() + (size_t)k * (ng + 1);
        const uint32_t *s8 = out.stream8.data();
        // An event's range starts at the block of ITS OWN word, not of the node's header (round 6): every node that counts the event
        // on its root path -- the node itself, whose cum is read at its last word, and its descendants -- has its last word at or
        // behind that block, so over / under stay bounds (tighter ones), and a block never starts more events than it has words:
        // the 8-bit start / inside counters of k_b3_group_tables cannot overflow, whatever the length of a branch (a node with
        // >= 256 mutation words useful for a tile used to wrap them).  Ends per block <= the mutation words of one root path (the
        // nodes whose subtree ends in a block it did not start in all hold the block's first word): 16 bits, see the guard below.
        auto word_block = [&](uint64_t d, uint32_t k) { return (uint32_t)((pos8_hdr[d] + 1u + k) >> B3_BLOCK_SHIFT); };
        auto end_block = [&](uint64_t d, uint32_t j) { return (uint32_t)((pos8_at[d + sub[j]] - 1u) >> B3_BLOCK_SHIFT); };
        // Each thread owns one run of consecutive depth-first indices (Par::run's static split, the same in both sweeps): it counts
        // its events per (list, group), the counts become cursors (groups in order, within a group the threads in order: the lists come
        // out in depth-first order whatever the number of threads), then it writes its events at its cursors.
        // List 3 = the events that are open at a group's first block (started in an earlier group, not over before this one): the
        // mutation words of the root path there -- a few dozen per group in a tree of logarithmic depth, the whole tree in a
        // caterpillar; counted through a difference array (+ at the first group an event is open at, - behind the last).
        std::vector<std::vector<uint32_t>> cur(T);
        par.run(N, [&](uint64_t b, uint64_t e, unsigned tid) {
            std::vector<uint32_t> &c = cur[tid];
            c.assign((size_t)4 * ng + 1, 0);
            uint32_t *c3 = c.data() + (size_t)3 * ng;
            for (uint64_t d = b; d < e; d++) {
                const uint32_t j = d2b[d];
                if (dropped[j] || !nw[j]) continue;
                const uint32_t b1 = end_block(d, j);
                for (uint32_t k = 0; k < nw[j]; k++) {
                    const uint32_t b0 = word_block(d, k);
                    if (b0 == b1) c[b0 >> B3_GROUP_SHIFT]++;
                    else {
                        c[(size_t)ng + (b0 >> B3_GROUP_SHIFT)]++; c[(size_t)2 * ng + (b1 >> B3_GROUP_SHIFT)]++;
                        const uint32_t g0 = (b0 >> B3_GROUP_SHIFT) + 1u, g1 = b1 >> B3_GROUP_SHIFT;   // open at the first block of groups g0 .. g1
                        if (g0 <= g1) { c3[g0]++; c3[g1 + 1u]--; }
                    }
                }
            }
            uint32_t open = 0;
            for (uint32_t g = 0; g < ng; g++) { open += c3[g]; c3[g] = open; }
        }, 1u << 16);
        uint64_t run = 0;
        for (int k = 0; k < 4; k++) {   // (the four lists one behind the other)
            for (uint32_t g = 0; g < ng; g++) {
                off[k][g] = (uint32_t)std::min<uint64_t>(run, UINT32_MAX);
                for (unsigned t2 = 0; t2 < T; t2++)
                    if (!cur[t2].empty()) { uint32_t &c = cur[t2][(size_t)k * ng + g]; const uint32_t n = c; c = (uint32_t)std::min<uint64_t>(run, UINT32_MAX); run += n; }
            }
            off[k][ng] = (uint32_t)std::min<uint64_t>(run, UINT32_MAX);
        }
        const uint64_t n_ev = off[2][0];   // (lists 0 and 1: every event once)
        if (run >= (1ull << 32) || run > 6 * n_ev + (16ull << 20)) {
            // (a tree so deep that the open lists dwarf the events themselves: no third bound for it)
            out.b3_group_off.clear();
            flat_lap("third bound: event lists (dropped: open lists too long)");
        } else {
        out.b3_events.resize(run);
        uint32_t *ev = out.b3_events.data();
        par.run(N, [&](uint64_t b, uint64_t e, unsigned tid) {
            uint32_t *c = cur[tid].data();
            for (uint64_t d = b; d < e; d++) {
                const uint32_t j = d2b[d];
                if (dropped[j] || !nw[j]) continue;
                const uint32_t b1 = end_block(d, j);
                for (uint32_t k = 0; k < nw[j]; k++) {
                    const uint32_t w = s8[pos8_hdr[d] + 1u + k];
                    const uint32_t pair = (w & 0x3FFFFFu) * 4u + ((w >> 22) & 3u);
                    const uint32_t b0 = word_block(d, k);
                    if (b0 == b1) ev[c[b0 >> B3_GROUP_SHIFT]++] = pair | ((b0 & (B3_GROUP_BLOCKS - 1u)) << 24);
                    else {
                        ev[c[(size_t)ng + (b0 >> B3_GROUP_SHIFT)]++] = pair | ((b0 & (B3_GROUP_BLOCKS - 1u)) << 24);
                        ev[c[(size_t)2 * ng + (b1 >> B3_GROUP_SHIFT)]++] = pair | ((b1 & (B3_GROUP_BLOCKS - 1u)) << 24);
                        for (uint32_t g = (b0 >> B3_GROUP_SHIFT) + 1u; g <= (b1 >> B3_GROUP_SHIFT); g++) ev[c[(size_t)3 * ng + g]++] = pair;
                    }
                }
            }
        }, 1u << 16);
        flat_lap("third bound: event lists");
        }
    }

    // ---- tie stream (phase 2 walks it one chunk at a time) ---------------------------------------------------------------------
    {
        auto t_big = [&](uint32_t j) { return j != 0 && subd[j] >= T_PRUNE_MIN_DWORDS && hsub[j] <= 255; };
        Buf<uint32_t> post_at(N);   // by DFS index: position (relative to the chunk) when the node is reached
        out.chunk_t_off.assign(NC + 1, 0);
        par.run(NC, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t c = b; c < e; c++) {
                uint32_t pos = 0;
                for (uint32_t d = cno[c]; d < cno[c + 1]; d++) {
                    const uint32_t j = d2b[d];
                    post_at[d] = pos;
                    if (dropped[j]) continue;
                    pos += (t_big(j) ? 2u : 0u) + 2u + nw[j];
                }
                out.chunk_t_off[c] = pos;
            }
        }, 16);
        const uint64_t total_t = par.exclusive_scan(out.chunk_t_off.data(), NC);
        out.chunk_t_off[NC] = (uint32_t)total_t;
        out.stream_t.resize(total_t);
        if (opt.keep_update_maps) out.post_of_bfs.resize(N); else out.post_of_bfs.clear();
        par.run(NC, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t c = b; c < e; c++) {
                if (opt.keep_update_maps)
                    for (uint32_t d = cno[c]; d < cno[c + 1]; d++) {
                        const uint32_t j = d2b[d];
                        out.post_of_bfs[j] = dropped[j] ? UINT32_MAX : out.chunk_t_off[c] + post_at[d] + (t_big(j) ? 2u : 0u);
                    }
                uint32_t *st = out.stream_t.data() + out.chunk_t_off[c];
                const uint32_t len = out.chunk_t_off[c + 1] - out.chunk_t_off[c];
                for (uint32_t d = cno[c]; d < cno[c + 1]; d++) {
                    const uint32_t j = d2b[d];
                    if (dropped[j]) continue;
                    uint32_t pos = post_at[d];
                    const uint32_t *rec = &out.stream[rec_off_d[d]];
                    const uint32_t nwords = rec[0] & 0xFFFFu;
                    if (t_big(j)) {   // a jump never leaves its chunk
                        const uint64_t d_end = (uint64_t)d + sub[j];
                        const uint32_t target = d_end >= cno[c + 1] ? len : post_at[d_end];
                        const uint32_t own_end = pos + 4u + nwords;
                        st[pos++] = T_INFO_MARK | ((uint32_t)hrev[j] << 16);   // (second hits below; 255: not available)
                        st[pos++] = (hsub[j] << 24) | std::min<uint32_t>(target - own_end, 0xFFFFFFu);
                    }
                    for (uint32_t k = 0; k < 2u + nwords; k++) st[pos++] = rec[k];
                }
            }
        }, 16);
    }
    flat_lap("tie stream");

    // Renumber the packed stream's slots by access frequency, hottest first: the
    // kernel keeps the first few in LDS and the cold remainder in a global scratch
    // (slot use is bell-shaped over the index, a handful of slots take ~95 %).  Then
    // convert every header to its final layout (ugp_flatten.hpp): only now is it known
    // which headers touch a cold slot and have to leave the fast path (H_SLOW).
    {
        std::vector<uint64_t> freq_t((size_t)T * 64, 0);
        par.run(out.stream8.size(), [&](uint64_t b, uint64_t e, unsigned tid) {
            uint64_t *f = &freq_t[(size_t)tid * 64];
            for (uint64_t i = b; i < e; i++) {
                const uint32_t w = out.stream8[i];
                if (!(w & H_TAG) || (w & (E_CHUNK_END | E_NOP | E_INFO))) continue;
                const uint32_t rs = w & 63u, ws = (w >> 6) & 63u;
                if (rs < RS_BOTTOM) f[rs]++;
                if (ws != WS_NONE) f[ws]++;
            }
        }, 1u << 16);
        uint64_t freq[64] = {0};
        for (unsigned th = 0; th < T; th++) for (int i = 0; i < 64; i++) freq[i] += freq_t[(size_t)th * 64 + i];
        // The preambles count too: every work unit replays one, and with most of the body pruned away the replays are a
        // large share of all slot accesses -- all of them to the slots of the top of the tree.  opt.pre_weight replays per
        // body pass (the body is mostly skipped, a unit's replay is not).
        if (opt.pre_weight) {
            uint64_t fp[64] = {0};
            for (uint32_t w : out.pre8_stream) {
                if (!(w & H_TAG) || (w & (E_CHUNK_END | E_NOP | E_INFO))) continue;
                const uint32_t rs = w & 63u, ws = (w >> 6) & 63u;
                if (rs < RS_BOTTOM) fp[rs]++;
                if (ws != WS_NONE) fp[ws]++;
            }
            for (int i = 0; i < 64; i++) freq[i] += fp[i] * opt.pre_weight;
        }
        uint32_t order[64], remap[64];
        for (uint32_t i = 0; i < 64; i++) order[i] = i;
        std::stable_sort(order, order + out.max_slots, [&](uint32_t a, uint32_t b) { return freq[a] > freq[b]; });
        for (uint32_t i = 0; i < 64; i++) remap[i] = i;
        for (uint32_t i = 0; i < out.max_slots; i++) remap[order[i]] = i;
        out.lds_slots = std::max<uint32_t>(1, std::min<uint32_t>(std::min<uint32_t>(opt.lds_slots, MAX_HOT_SLOTS), out.max_slots));
        const uint32_t hot = out.lds_slots;
        auto finalize8 = [&](UVec<uint32_t> &v) {
            par.run(v.size(), [&](uint64_t b, uint64_t e, unsigned) {
                for (uint64_t i = b; i < e; i++) {
                    uint32_t &w = v[i];
                    if (!(w & H_TAG)) continue;
                    if (w & E_INFO) {   // (first: the jump length overlaps the other intermediate flag bits)
                        w = H_TAG | H_INFO | H_RARE | (w & E_SIB ? H_SIB : 0u) | (((w >> 22) & 0x7Fu) << INFO_HS_SHIFT) | (w & ((7u << INFO_HR_SHIFT) | INFO_JUMP_MASK));
                        continue;
                    }
                    if (w & E_CHUNK_END) { w = H_TAG | H_RARE | H_CHUNK_END; continue; }
                    if (w & E_NOP) { w = H_TAG | H_RARE | H_NOP; continue; }
                    uint32_t rs = w & 63u, ws = (w >> 6) & 63u;
                    uint32_t h = H_TAG;
                    bool slow = (w & E_LONG) != 0;
                    if (rs == RS_REG) h |= H_REG;
                    else { rs = remap[rs]; h |= rs << H_RSLOT_SHIFT; slow |= rs >= hot; }
                    if (ws != WS_NONE) { ws = remap[ws]; h |= H_STORE | (ws << H_WSLOT_SHIFT); slow |= ws >= hot; }
                    if (w & E_SKIPD) h |= H_SKIPD;
                    if (w & E_NOSCORE) h |= H_NOSCORE;
                    if (w & E_END) h |= H_END;
                    if (w & E_FREE) h |= H_FREE;
                    if (slow) h |= H_SLOW | H_RARE;
                    w = h;
                }
            }, 1u << 16);
        };
        finalize8(out.stream8);
        finalize8(out.pre8_stream);
        // every chunk-end word names the length of the chunk behind it (its own end word included; 0: does not fit / none):
        // the walk always knows where the open chunk ends without a table lookup
        par.run(NC, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t c = b; c < e; c++) {
                const uint64_t ln = c + 1 < NC ? (uint64_t)out.chunk8_body_off[c + 2] - out.chunk8_body_off[c + 1] : 0;
                out.stream8[out.chunk8_body_off[c + 1] - 1] |= (ln <= CE_LEN_MASK ? (uint32_t)ln : 0u) << CE_LEN_SHIFT;
            }
        }, 1u << 12);
    }
    flat_lap("slot renumbering");

    if (extras) {
        extras->node_pair.clear();
        if (bfs_levels) {
            extras->node_pair.resize(2 * (N + 1));
            par.run(N, [&](uint64_t b, uint64_t e, unsigned) {
                for (uint64_t j = b; j < e; j++) { extras->node_pair[2 * j] = child_off[j]; extras->node_pair[2 * j + 1] = rec_off_d[dfsidx[j]]; }
            });
            extras->node_pair[2 * N] = child_off[N]; extras->node_pair[2 * N + 1] = 0;
            std::vector<uint64_t> wide(T, 0);
            par.run(N, [&](uint64_t b, uint64_t e, unsigned tid) {
                uint64_t w = 0;
                for (uint64_t j = b; j < e; j++) { const uint32_t c = child_off[j + 1] - child_off[j]; if (c > 16) w += c; }
                wide[tid] = w;
            });
            extras->children_of_wide_nodes = 0;
            for (uint64_t w : wide) extras->children_of_wide_nodes += w;
        }
        extras->sub.swap(sub); extras->dfsidx.swap(dfsidx);
    }
    return UGP_OK;
}

}  // namespace ugp
