// driver.cpp -- usher-compatible placement driver.  Written from scratch; the
// behaviour (messages, file formats, ordering rules) follows the reference's
// src/usher_common.cpp, cited inline.
#include "driver.hpp"
#include "par.hpp"

#include <sys/stat.h>
#include <sys/time.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <numeric>
#include <random>
#include <chrono>
#include <functional>
#include <climits>
#include <cmath>
#include <thread>
#include <unordered_map>
#include <unordered_set>

namespace uh {

namespace {

struct Timer {   // usher_graph.hpp:15-31
    timeval t0;
    void start() { gettimeofday(&t0, nullptr); }
    long stop() const {
        timeval t1;
        gettimeofday(&t1, nullptr);
        return (long)((t1.tv_sec - t0.tv_sec) * 1000 + (t1.tv_usec - t0.tv_usec) / 1000.0 + 0.5);
    }
};

// Flat BFS arrays of the current tree: the ugp_tree_desc the backend takes.
struct FlatTree {
    std::vector<Node *> bfs;
    std::vector<uint32_t> parent;
    std::vector<uint64_t> mut_off;
    std::vector<int32_t> pos;
    std::vector<uint8_t> ref, par, nuc;
    ugp_tree_desc desc{};
    uint32_t epoch = 0;                               // nodes stamped with this epoch carry their index here (Node::flat_index)
    std::vector<uint32_t> leaves;                     // leaves below each node at build time (Tree::get_num_leaves)
    bool has(const Node *n) const { return epoch != 0 && n->flat_epoch == epoch; }   // (epoch 0: never built)
    // Tree -> breadth-first arrays, on the host threads: the expansion level by level (the next level is the
    // concatenation of the children lists of this one: sizes, prefix sum, copy), then per-node fills.
    void build(const Tree &T) {
        static uint32_t next_epoch = 0;
        epoch = ++next_epoch;
        const size_t N = T.all_nodes.size();
        bfs.clear();
        if (!T.root) { parent.clear(); mut_off.assign(1, 0); leaves.clear(); desc = ugp_tree_desc{}; return; }
        bfs.resize(N);
        bfs[0] = T.root;
        std::vector<uint64_t> level_off{0, 1};   // nodes of level l: [level_off[l], level_off[l + 1])
        std::vector<uint64_t> cnt;
        for (size_t lo = 0, hi = 1; lo < hi;) {
            cnt.resize(hi - lo);
            parallel_for(hi - lo, [&](uint64_t b, uint64_t e, unsigned) { for (uint64_t i = b; i < e; i++) cnt[i] = bfs[lo + i]->children.size(); }, 2048);
            const uint64_t total = exclusive_scan(cnt.data(), hi - lo);
            if (hi + total > bfs.size()) bfs.resize(hi + total);   // (all_nodes out of step with the tree: never expected)
            parallel_for(hi - lo, [&](uint64_t b, uint64_t e, unsigned) {
                for (uint64_t i = b; i < e; i++) {
                    Node **dst = &bfs[hi + cnt[i]];
                    for (Node *c : bfs[lo + i]->children) *dst++ = c;
                }
            }, 2048);
            lo = hi; hi += total;
            if (total) level_off.push_back(hi);
        }
        bfs.resize(level_off.back());
        const size_t n = bfs.size();
        parent.resize(n); leaves.resize(n);
        mut_off.resize(n + 1);
        parallel_for(n, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t j = b; j < e; j++) { bfs[j]->flat_index = (uint32_t)j; bfs[j]->flat_epoch = epoch; mut_off[j] = bfs[j]->mutations.size(); }
        });
        mut_off[n] = 0;
        const uint64_t M = exclusive_scan(mut_off.data(), n + 1);
        pos.resize(M); ref.resize(M); par.resize(M); nuc.resize(M);
        parallel_for(n, [&](uint64_t b, uint64_t e, unsigned) {
            for (uint64_t j = b; j < e; j++) {
                parent[j] = bfs[j]->parent ? bfs[j]->parent->flat_index : UINT32_MAX;
                uint64_t k = mut_off[j];
                for (const Mutation &m : bfs[j]->mutations) {
                    pos[k] = m.position; ref[k] = (uint8_t)m.ref_nuc; par[k] = (uint8_t)m.par_nuc; nuc[k] = (uint8_t)m.mut_nuc;
                    k++;
                }
            }
        });
        // leaves below each node: the levels bottom-up, every node summing over its children
        for (size_t l = level_off.size() - 1; l-- > 0;) {
            const uint64_t b0 = level_off[l];
            parallel_for(level_off[l + 1] - b0, [&](uint64_t b, uint64_t e, unsigned) {
                for (uint64_t j = b0 + b; j < b0 + e; j++) {
                    uint32_t v = 0;
                    for (const Node *c : bfs[j]->children) v += leaves[c->flat_index];
                    leaves[j] = bfs[j]->children.empty() ? 1u : v;
                }
            }, 2048);
        }
        desc.n_nodes = n; desc.parent = parent.data(); desc.mut_off = mut_off.data();
        desc.mut_pos = pos.data(); desc.mut_ref = ref.data(); desc.mut_par = par.data(); desc.mut_nuc = nuc.data();
    }
};

// Does `a` come before `b` in the breadth-first expansion of the tree as it is now
// (mutation_annotated_tree.cpp:1225-1251)?  Levels first; within a level the order of the parents, then
// the position among the parent's children.
bool bfs_before(const Node *a, const Node *b) {
    if (a == b) return false;
    if (a->level != b->level) return a->level < b->level;
    while (a->parent != b->parent) { a = a->parent; b = b->parent; }
    for (const Node *c : a->parent->children) {
        if (c == a) return true;
        if (c == b) return false;
    }
    return false;
}

struct FlatQueries {
    std::vector<uint64_t> ent_off{0};
    std::vector<int32_t> pos;
    std::vector<uint8_t> ref, nuc, miss;
    ugp_queries desc{};
    void add(const std::vector<Mutation> &muts) {
        for (const Mutation &m : muts) {
            pos.push_back(m.position); ref.push_back((uint8_t)m.ref_nuc);
            nuc.push_back((uint8_t)m.mut_nuc); miss.push_back(m.is_missing ? 1 : 0);
        }
        ent_off.push_back(pos.size());
    }
    void finish() {
        desc.n_queries = ent_off.size() - 1; desc.ent_off = ent_off.data(); desc.pos = pos.data();
        desc.ref = ref.data(); desc.nuc = nuc.data(); desc.is_missing = miss.data();
    }
};

bool by_pos(const Mutation &a, const Mutation &b) { return a.position < b.position; }

}  // namespace

// ---------------------------------------------------------------------------
// mapper2_body(inp, true, true) for ONE node: used for the winning node only
// (what pass 2 does, usher_common.cpp:426-449) and for the optimal rows of -p.
// ---------------------------------------------------------------------------
void node_vecs(const Node *node, const std::vector<Mutation> &sample, NodeVecs &out) {
    out.excess.clear(); out.imputed.clear(); out.set_difference = 0; out.has_unique = false; out.eligible = false;
    int node_num_mut = 0, num_common = 0;      // :184-185
    std::vector<Mutation> anc;                 // ancestral_mutations (usher_mapper.cpp:178-179)
    auto anc_has = [&](int32_t p) { for (const Mutation &m : anc) if (m.position == p) return true; return false; };
    if (!node->is_root()) {                    // branch loop, :190-264
        size_t start = 0;
        for (const Mutation &m1 : node->mutations) {
            node_num_mut++;
            if (m1.masked()) { out.has_unique = true; break; }                 // :197-200
            bool found = false, found_pos = false;
            for (size_t k = start; k < sample.size(); k++) {
                const Mutation &m2 = sample[k];
                start = k;
                if (m1.position == m2.position) {
                    found_pos = true;
                    if (m2.is_missing) { found = true; num_common++; }        // :209-211
                    else if (m2.mut_nuc & m1.mut_nuc) {                        // :214-235
                        Mutation m = m1; m.is_missing = false;
                        anc.push_back(m); out.excess.push_back(m);
                        found = true; num_common++;
                        break;
                    }
                }
                if (m1.position < m2.position) break;
            }
            if (!found) {
                if (!found_pos && m1.mut_nuc == m1.ref_nuc) {                  // :244-259
                    Mutation m = m1; m.is_missing = false;
                    anc.push_back(m); out.excess.push_back(m);
                    num_common++;
                } else out.has_unique = true;
            }
        }
    } else {
        for (const Mutation &m : node->mutations) anc.push_back(m);            // :266-269
    }
    for (const Node *a = node->parent; a; a = a->parent)                       // :275-286
        for (const Mutation &m : a->mutations)
            if (!m.masked() && !anc_has(m.position)) anc.push_back(m);
    std::sort(anc.begin(), anc.end(), by_pos);                                 // :289
    for (const Mutation &m1 : sample) {                                        // :292-388
        if (m1.is_missing) continue;
        bool found_pos = false, found = false;
        const bool has_ref = (m1.mut_nuc & m1.ref_nuc) != 0;
        int8_t anc_nuc = m1.ref_nuc;
        for (const Mutation &m2 : anc) {
            if (m2.masked()) continue;
            if (m1.position == m2.position) {
                found_pos = true; anc_nuc = m2.mut_nuc;
                if (m1.mut_nuc & anc_nuc) found = true;
                break;
            }
        }
        const bool ambiguous = (m1.mut_nuc & (m1.mut_nuc - 1)) != 0;
        Mutation m = m1; m.is_missing = false; m.par_nuc = anc_nuc;
        if (found) {
            if (ambiguous) { m.mut_nuc = anc_nuc; out.imputed.push_back(m); }              // :322-335
        } else if (!found_pos && has_ref) {
            if (ambiguous) { m.mut_nuc = m1.ref_nuc; out.imputed.push_back(m); }           // :341-351
        } else {                                                                             // :356-387
            if (has_ref) m.mut_nuc = m1.ref_nuc;
            else { m.mut_nuc = 0; for (int b = 0; b < 4; b++) if (m1.mut_nuc & (1 << b)) { m.mut_nuc = (int8_t)(1 << b); break; } }
            if (ambiguous) out.imputed.push_back(m);
            if (m.mut_nuc != m.par_nuc) { out.excess.push_back(m); out.set_difference++; }
        }
    }
    for (const Mutation &m1 : anc) {                                           // back-mutations, :393-445
        if (m1.masked()) continue;   // masked root entries: ref == par == 0, never counted (:428-436)
        bool found = false, found_pos = false;
        for (const Mutation &m2 : sample) {
            if (m1.position == m2.position) {
                found_pos = true;
                if (m2.is_missing) { found = true; break; }
                if (m2.mut_nuc & m1.mut_nuc) found = true;
            }
        }
        if (found || found_pos) continue;
        if (m1.mut_nuc == m1.ref_nuc) continue;
        Mutation m = m1; m.par_nuc = m1.mut_nuc; m.mut_nuc = m1.ref_nuc; m.is_missing = false;
        out.excess.push_back(m);
        out.set_difference++;
    }
    const bool leaf = node->is_leaf();                                          // :454-455
    out.eligible = node->is_root() || (out.has_unique && !leaf && num_common > 0 && node_num_mut != num_common) ||
                   (leaf && num_common > 0) || (!out.has_unique && !leaf && node_num_mut == num_common);
}

// ---------------------------------------------------------------------------

// ---------------------------------------------------------------------------
// The search of usher_common.cpp:389-449 for ONE sample on the host, with the literal routine above at every node.
// For inputs outside the preconditions of the device algorithm (rows of a sample not strictly increasing by position:
// the reference's scans then depend on the row order, usher_mapper.cpp:204-242, 393-445; tree alleles that are not
// one base): such samples never reach the backend, their answers come from here -- slow (N literal evaluations)
// but the reference's own answer.
// ---------------------------------------------------------------------------
struct HostSearch {
    int best = 0;
    size_t num_best = 0;
    Node *best_node = nullptr;
    bool best_has_unique = false;
    std::vector<std::pair<Node *, bool>> ties;   // breadth-first order
    std::vector<int32_t> scores;                 // (want_scores) breadth-first order, +1 where not eligible
};

static void host_search(const std::vector<Node *> &bfs, const std::vector<Mutation> &sample, bool want_scores, HostSearch &out) {
    const size_t n = bfs.size();
    std::unordered_map<const Node *, size_t> idx;
    idx.reserve(n * 2);
    for (size_t j = 0; j < n; j++) idx[bfs[j]] = j;
    std::vector<uint32_t> leaves(n, 0);
    for (size_t j = n; j-- > 0;) {
        if (bfs[j]->is_leaf()) leaves[j] = 1;
        if (bfs[j]->parent) leaves[idx[bfs[j]->parent]] += leaves[j];
    }
    out = HostSearch();
    out.best = INT32_MAX;
    if (want_scores) out.scores.assign(n, 0);
    size_t best_j = 0;
    NodeVecs nv;
    for (size_t j = 0; j < n; j++) {
        node_vecs(bfs[j], sample, nv);
        if (want_scores) out.scores[j] = nv.set_difference + (nv.eligible ? 0 : 1);
        if (!nv.eligible || nv.set_difference > out.best) continue;
        if (nv.set_difference < out.best) {
            out.best = nv.set_difference; out.ties.clear(); out.num_best = 0;
            out.best_node = bfs[j]; out.best_has_unique = nv.has_unique; best_j = j;
        } else if (leaves[j] > leaves[best_j] || (leaves[j] == leaves[best_j] && j > best_j)) {   // usher_mapper.cpp:483-486
            out.best_node = bfs[j]; out.best_has_unique = nv.has_unique; best_j = j;
        }
        out.num_best++;
        out.ties.push_back({bfs[j], nv.has_unique});
    }
}

// ---------------------------------------------------------------------------
// Closed-form evaluation of ONE node for a sample whose rows are in order (the same formulas the
// device kernels evaluate, proven equal to the literal routine in tests/test_closed_form.py): used by the add mode to
// re-derive batched answers on the nodes touched since the batch was placed -- a few thousand (sample, node) pairs
// per insertion, for which the literal routine's list scans (~1 us each) were the second largest cost.
//   D(parent) = D(bottom) + sum over positions where the parent's state differs from the reference of
//               ([state not in S] - [ref not in S]);   cost = D(parent) + sum over own mutations of min(delta, 0)
// The parent's state is cached per touched node; an insertion never changes the state at an existing node
// (usher_common.cpp:652-765 only splits one branch in two), so the cache stays valid until the node is touched again.
// ---------------------------------------------------------------------------
struct TouchedInfo {
    struct PathEnt { int32_t pos; uint8_t allele, ref; };
    struct Own { int32_t pos; uint8_t mut, prev, ref; };
    std::vector<PathEnt> path;   // positions where the parent's state is not the reference base
    std::vector<Own> own;        // the node's mutations in front of its first masked one
    bool masked = false;
};

struct DenseSample {             // S(p) of one sample as an array over positions: 0 = no row (reference base)
    std::vector<uint8_t> s;
    int dbot = 0;
    void set(const std::vector<Mutation> &rows) {
        dbot = 0;
        for (const Mutation &m : rows) {
            if (m.position < 0) continue;
            if ((size_t)m.position >= s.size()) s.resize((size_t)m.position * 2 + 1024, 0);
            const uint8_t a = m.is_missing ? 15 : (uint8_t)m.mut_nuc;
            s[m.position] = a;
            if (!m.is_missing && (a & (uint8_t)m.ref_nuc) == 0) dbot++;
        }
    }
    void clear(const std::vector<Mutation> &rows) { for (const Mutation &m : rows) if (m.position >= 0) s[m.position] = 0; }
    uint8_t at(int32_t p, uint8_t ref) const { return ((size_t)p < s.size() && s[p]) ? s[p] : ref; }
};

struct TouchedCache {
    std::unordered_map<const Node *, TouchedInfo> info;
    std::vector<uint32_t> mark;      // per position: stamp of the walk that set `state`
    std::vector<uint8_t> state;
    uint32_t stamp = 0;
    void build(const Node *x) {
        TouchedInfo &ti = info[x];
        ti.path.clear(); ti.own.clear(); ti.masked = false;
        if (++stamp == 0) { std::fill(mark.begin(), mark.end(), 0u); stamp = 1; }
        auto touch = [&](int32_t p) { if ((size_t)p >= mark.size()) { mark.resize((size_t)p * 2 + 1024, 0); state.resize(mark.size(), 0); } };
        for (const Node *a = x->parent; a; a = a->parent)
            for (const Mutation &m : a->mutations) {
                if (m.masked()) continue;
                touch(m.position);
                if (mark[m.position] == stamp) continue;          // a more recent mutation at this position was seen below
                mark[m.position] = stamp; state[m.position] = (uint8_t)m.mut_nuc;
                if (m.mut_nuc != m.ref_nuc) ti.path.push_back({m.position, (uint8_t)m.mut_nuc, (uint8_t)m.ref_nuc});
            }
        for (const Mutation &m : x->mutations) {
            if (m.masked()) { ti.masked = true; break; }
            touch(m.position);
            const uint8_t prev = mark[m.position] == stamp ? state[m.position] : (uint8_t)m.ref_nuc;
            ti.own.push_back({m.position, (uint8_t)m.mut_nuc, prev, (uint8_t)m.ref_nuc});
        }
    }
    // (eligible, cost, has_unique) of usher_mapper.cpp:454-455 / :172-445 for node x (not the root)
    void eval(const Node *x, const DenseSample &S, bool &eligible, int &cost, bool &has_unique) const {
        const TouchedInfo &ti = info.at(x);
        int D = S.dbot;
        for (const auto &e : ti.path) {
            const uint8_t sp = S.at(e.pos, e.ref);
            D += ((sp & e.allele) == 0) - ((sp & e.ref) == 0);
        }
        int neg = 0, common = 0;
        const int num_mut = (int)ti.own.size() + (ti.masked ? 1 : 0);
        for (const auto &o : ti.own) {
            const uint8_t sp = S.at(o.pos, o.ref);
            const int c = (sp & o.mut) != 0, pr = (sp & o.prev) != 0;
            common += c;
            if (pr - c < 0) neg += pr - c;
        }
        cost = D + neg;
        eligible = common > 0 || (!x->is_leaf() && num_mut == 0);
        has_unique = ti.masked || common != num_mut;
    }
};

static bool rows_in_order(const std::vector<Mutation> &rows) {
    for (size_t i = 1; i < rows.size(); i++) if (rows[i].position <= rows[i - 1].position) return false;
    return true;
}
// Rows the backend accepts in place of an out-of-order sample (its answer for them is never used).
static std::vector<Mutation> ordered_rows(const std::vector<Mutation> &rows) {
    std::vector<Mutation> v(rows);
    std::stable_sort(v.begin(), v.end(), by_pos);
    v.erase(std::unique(v.begin(), v.end(), [](const Mutation &a, const Mutation &b) { return a.position == b.position; }), v.end());
    return v;
}
static bool tree_alleles_are_bases(const Tree &T) {
    auto one = [](int8_t a) { return a == 1 || a == 2 || a == 4 || a == 8; };
    bool ok = true;
    T.all_nodes.for_each([&](const Node *n) {
        for (const Mutation &m : n->mutations)
            if (!m.masked() && (!one(m.mut_nuc) || !one(m.ref_nuc))) ok = false;
    });
    return ok;
}

static bool write_text(const std::string &path, const std::string &text) {
    FILE *f = fopen(path.c_str(), "w");
    if (!f) return false;
    fwrite(text.data(), 1, text.size(), f);
    fclose(f);
    return true;
}

// Insert `sample` next to / below `best` (usher_common.cpp:652-765).
static void insert_sample(Tree &T, Node *best, bool as_sibling, const std::string &sample, const std::vector<Mutation> &excess,
                          std::vector<Node *> &touched /* nodes created or whose branch changed */) {
    auto matches = [](const Mutation &a, const Mutation &b) { return a.position == b.position && a.mut_nuc == b.mut_nuc; };
    if (as_sibling) {                                                           // :654-729
        const std::string nid = T.new_internal_node_id();
        Node *mid = T.create_node(nid, best->parent);
        Node *leaf = T.create_node(sample, mid);
        T.reattach(best, mid);                                                  // move_node(best, nid): children = [sample, best]
        const std::vector<Mutation> branch = best->mutations;
        best->mutations.clear();
        std::vector<Mutation> common, l1, l2;
        for (const Mutation &m1 : branch) {                                     // :677-694
            bool found = false;
            if (!m1.masked()) for (const Mutation &m2 : excess) if (matches(m1, m2)) { found = true; break; }
            if (!found) l1.push_back(m1);
        }
        for (const Mutation &m1 : excess) {                                     // :696-715
            bool found = false;
            if (!m1.masked()) for (const Mutation &m2 : branch) if (matches(m1, m2)) { found = true; break; }
            (found ? common : l2).push_back(m1);
        }
        for (const Mutation &m : common) mid->add_mutation(m);
        for (const Mutation &m : l1) best->add_mutation(m);
        for (const Mutation &m : l2) leaf->add_mutation(m);
        touched.push_back(mid); touched.push_back(leaf); touched.push_back(best);
    } else {                                                                    // :731-764
        Node *leaf = T.create_node(sample, best);
        for (const Mutation &m1 : excess) {
            bool found = false;
            if (!m1.masked()) for (const Mutation &m2 : best->mutations) if (matches(m1, m2)) { found = true; break; }
            if (!found) leaf->add_mutation(m1);
        }
        touched.push_back(leaf);
    }
}

// ---------------------------------------------------------------------------
// -K / -k: subtrees around the newly placed samples (get_random_single_subtree / get_random_sample_subtrees,
// mutation_annotated_tree.cpp:1693-1990).  The reference draws with std::rand() (default seed for -K, srand(0)
// for -k) and std::shuffle(std::default_random_engine{}); the same calls are made here, in the same order.
// ---------------------------------------------------------------------------
static void write_subtree_files(Tree &T, Tree &sub, const std::string &stem, const char *what, int number) {
    sub.rotate_for_display();
    const std::string nh = stem + ".nh";
    if (number < 0) fprintf(stderr, "%s %s.\n", what, nh.c_str());
    else fprintf(stderr, "Writing subtree %d to file %s.\n", number, nh.c_str());
    write_text(nh, newick(sub, sub.root, true, true));
    const std::string mf = stem + "-mutations.txt";
    if (number < 0) fprintf(stderr, "Writing list of mutations at the nodes of the single subtree to file %s\n", mf.c_str());
    else fprintf(stderr, "Writing list of mutations at the nodes of subtree %d to file %s\n", number, mf.c_str());
    std::string text;
    for (Node *n : sub.dfs()) {
        text += n->id + ": ";
        for (size_t i = 0; i < n->mutations.size(); i++) text += n->mutations[i].str() + (i + 1 < n->mutations.size() ? "," : "");
        text += "\n";
    }
    write_text(mf, text);
    std::string expanded;
    for (Node *l : sub.leaves()) {
        auto it = T.condensed_nodes.find(l->id);
        if (it == T.condensed_nodes.end()) continue;
        expanded += l->id + ": ";
        for (const std::string &c : it->second) expanded += c + " ";
        expanded += "\n";
    }
    if (!expanded.empty()) {
        const std::string ef = stem + "-expanded.txt";
        if (number < 0) fprintf(stderr, "Subtree has condensed nodes.\nExpanding the condensed nodes for the single subtree in file %s\n", ef.c_str());
        else fprintf(stderr, "Subtree %d has condensed nodes.\nExpanding the condensed nodes for subtree %d in file %s\n", number, number, ef.c_str());
        write_text(ef, expanded);
    }
}

static void write_single_subtree(Tree &T, const std::vector<MissingSample> &missing, const std::string &outdir, size_t subtree_size,
                                 size_t tree_idx, bool use_tree_idx) {   // :1693-1786
    const std::string pre = use_tree_idx ? "/tree-" + std::to_string(tree_idx) + "-" : "/";
    std::unordered_set<Node *> keep;
    std::vector<Node *> order;   // (insertion order; the reference's std::set<Node*> order does not influence the result)
    auto add = [&](Node *n) { if (n && keep.insert(n).second) order.push_back(n); };
    for (const auto &ms : missing) add(T.get_node(ms.name));
    const std::vector<Node *> all = T.leaves();
    for (size_t i = 0; i < all.size(); i++) {
        add(all[(size_t)std::rand() % all.size()]);
        if (keep.size() >= subtree_size + missing.size()) break;
    }
    std::vector<std::string> ids;
    for (Node *n : order) ids.push_back(n->id);
    Tree sub;
    std::string err;
    if (!get_subtree(T, ids, sub, err)) { fprintf(stderr, "ERROR: %s\n", err.c_str()); return; }
    char what[160];
    snprintf(what, sizeof what, "Writing single subtree with %zu randomly added leaves to file", subtree_size);
    write_subtree_files(T, sub, outdir + pre + "single-subtree", what, -1);
}

static void write_sample_subtrees(Tree &T, const std::vector<MissingSample> &missing, const std::string &outdir, size_t subtree_size,
                                  size_t tree_idx, bool use_tree_idx) {   // :1788-1989
    fprintf(stderr, "Computing subtrees for %ld samples. \n\n", (long)missing.size());
    const std::string pre = use_tree_idx ? "/tree-" + std::to_string(tree_idx) + "-" : "/";
    const size_t random_size = subtree_size / 5, nearest_size = subtree_size - random_size;
    std::srand(0);
    {   // (the reference draws a set of random leaves here that it never uses; the draws advance the generator all the same)
        const std::vector<Node *> all = T.leaves();
        std::unordered_set<Node *> seen;
        for (size_t i = 0; i < all.size(); i++) {
            seen.insert(all[(size_t)std::rand() % all.size()]);
            if (seen.size() >= subtree_size) break;
        }
    }
    std::vector<bool> displayed(missing.size(), false);
    for (size_t i = 0; i < missing.size(); i++) if (!T.get_node(missing[i].name)) displayed[i] = true;
    int num_subtrees = 0;
    for (size_t i = 0; i < missing.size(); i++) {
        if (displayed[i]) continue;
        Node *last_anc = T.get_node(missing[i].name);
        std::vector<std::string> keep;
        for (Node *anc : T.rsearch(last_anc, true)) {
            const size_t nl = T.num_leaves(anc);
            if (nl < subtree_size) { last_anc = anc; continue; }
            if (nl > subtree_size) {
                for (Node *l : T.leaves(last_anc)) keep.push_back(l->id);
                struct NodeDist { Node *node; uint32_t num_mut; };
                std::vector<NodeDist> dist;
                for (Node *l : T.leaves(anc)) {
                    if (T.is_ancestor(last_anc, l)) continue;
                    uint32_t d = 0;
                    for (Node *a : T.rsearch(l, true)) { if (a == anc) break; d += (uint32_t)a->mutations.size(); }
                    dist.push_back({l, d});
                }
                std::sort(dist.begin(), dist.end(), [](const NodeDist &a, const NodeDist &b) { return a.num_mut < b.num_mut; });
                for (const NodeDist &n : dist) { if (keep.size() >= nearest_size) break; keep.push_back(n.node->id); }
                if (nearest_size < subtree_size && nearest_size < dist.size()) {
                    std::vector<NodeDist> rest(dist.begin() + nearest_size, dist.end());
                    std::shuffle(rest.begin(), rest.end(), std::default_random_engine{});
                    for (const NodeDist &n : rest) { if (keep.size() == subtree_size) break; keep.push_back(n.node->id); }
                }
            } else {
                for (Node *l : T.leaves(anc)) { if (keep.size() == subtree_size) break; keep.push_back(l->id); }
            }
            Tree sub;
            std::string err;
            if (!get_subtree(T, keep, sub, err)) { fprintf(stderr, "ERROR: %s\n", err.c_str()); return; }
            for (size_t j = i + 1; j < missing.size(); j++) if (!displayed[j] && sub.get_node(missing[j].name)) displayed[j] = true;
            ++num_subtrees;
            write_subtree_files(T, sub, outdir + pre + "subtree-" + std::to_string(num_subtrees), "", num_subtrees);
            break;
        }
    }
}

// Every output of usher_common.cpp:808-1044 for the final tree(s); with several trees (--multiple-placements) the
// file names carry the tree number (:836-838, :859-861, :893-895, :918-919) and only the first tree is saved (:1027-1034).
// The final tree's text and parsimony score computed ahead of time (-n: the tree never changes, so the 10M-node newick is produced on
// a thread of its own while the samples are placed and their statistics written)
struct FinalText {
    std::string text;
    size_t parsimony = 0;
    std::thread th;
    void start(const Tree &T, bool uncondensed) {
        th = std::thread([this, &T, uncondensed]() { text = newick(T, T.root, true, true, uncondensed); parsimony = T.parsimony_score(); });
    }
    void wait() { if (th.joinable()) th.join(); }
    ~FinalText() { wait(); }
};

static int write_outputs(const Options &opt, std::vector<Tree *> &trees, std::vector<MissingSample> &missing,
                         const std::vector<std::string> &low_confidence, FinalText *ready = nullptr) {
    Timer timer;
    const std::string &outdir = opt.outdir;
    const size_t num_trees = trees.size();
    auto numbered = [&](const std::string &stem, const std::string &ext, size_t t, bool dash) {
        return num_trees > 1 ? outdir + "/" + stem + (dash ? "-" : "") + std::to_string(t + 1) + ext : outdir + "/" + stem + ext;
    };
    if (opt.collapse_output_tree) {                                             // :808-822
        for (size_t t = 0; t < num_trees; t++) {
            timer.start();
            if (num_trees > 1) fprintf(stderr, "Collapsing output tree %zu.\n", t + 1);
            else fprintf(stderr, "Collapsing output tree.\n");
            trees[t]->collapse_tree();
            fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
        }
    }
    for (size_t t = 0; t < num_trees; t++) {                                    // :828-881
        Tree &T = *trees[t];
        timer.start();
        if (opt.write_uncondensed) {
            const std::string fn = numbered("uncondensed-final-tree", ".nh", t, true);
            if (num_trees > 1) fprintf(stderr, "Writing uncondensed final tree %zu to file %s \n", t + 1, fn.c_str());
            else fprintf(stderr, "Writing uncondensed final tree to file %s \n", fn.c_str());
            if (ready) ready->wait();
            fprintf(stderr, "The parsimony score for this tree is: %zu \n", ready ? ready->parsimony : T.parsimony_score());
            write_text(fn, ready ? ready->text : newick(T, T.root, true, true, true));
        } else {
            const std::string fn = numbered("final-tree", ".nh", t, true);
            if (num_trees > 1) fprintf(stderr, "Writing final tree %zu to file %s \n", t + 1, fn.c_str());
            else fprintf(stderr, "Writing final tree to file %s \n", fn.c_str());
            if (ready) ready->wait();
            fprintf(stderr, "The parsimony score for this tree is: %zu \n", ready ? ready->parsimony : T.parsimony_score());
            write_text(fn, ready ? ready->text : newick(T, T.root, true, true));
        }
        fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
    }
    if (!missing.empty()) {
        for (size_t t = 0; t < num_trees; t++) {                                // :883-907
            Tree &T = *trees[t];
            timer.start();
            const std::string fn = numbered("mutation-paths", ".txt", t, true);   // get_sample_mutation_paths, mutation_annotated_tree.cpp:1991-2050
            if (num_trees > 1) fprintf(stderr, "Writing mutation paths for tree %zu to file %s \n", t + 1, fn.c_str());
            else fprintf(stderr, "Writing mutation paths to file %s \n", fn.c_str());
            std::string text;
            for (auto &ms : missing) {
                Node *n = T.get_node(ms.name);
                if (!n) continue;
                std::vector<std::string> parts;
                for (Node *a : T.rsearch(n, true)) {
                    if (a->mutations.empty()) continue;
                    std::string p = a->id + ":";
                    for (size_t k = 0; k < a->mutations.size(); k++) p += a->mutations[k].str() + (k + 1 < a->mutations.size() ? "," : " ");
                    parts.push_back(std::move(p));
                }
                text += ms.name + "\t";
                for (size_t i = parts.size(); i-- > 0;) text += parts[i];
                text += "\n";
            }
            write_text(fn, text);
            fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
        }
        for (size_t t = 0; t < num_trees; t++) {                                // clades.txt, :909-970
            Tree &T = *trees[t];
            const size_t n_ann = T.num_annotations();
            if (n_ann == 0) continue;
            timer.start();
            const std::string cf = numbered("clades", ".txt", t, false);
            if (num_trees > 1) fprintf(stderr, "Writing clade annotations for tree %zu to file %s \n", t + 1, cf.c_str());
            else fprintf(stderr, "Writing clade annotations to file %s \n", cf.c_str());
            std::string ctext;
            for (auto &ms : missing) {
                if (ms.best_clade_assignment.empty()) continue;
                ctext += ms.name + "\t";
                for (size_t k = 0; k < n_ann; k++) {
                    ctext += ms.best_clade_assignment[k];
                    if (opt.max_trees == 1 && opt.detailed_clades) {
                        ctext += "*|";
                        std::string cur; int cnt = 0;
                        const auto &all = ms.clade_assignments[k];
                        std::vector<std::string> segs;
                        for (const std::string &c : all) {
                            if (c == cur) cnt++;
                            else { if (cnt > 0) segs.push_back(cur + "(" + std::to_string(cnt) + "/" + std::to_string(all.size()) + ")"); cur = c; cnt = 1; }
                        }
                        for (auto &sg : segs) ctext += sg + ",";
                        if (cnt > 0) ctext += cur + "(" + std::to_string(cnt) + "/" + std::to_string(all.size()) + ")";
                    }
                    if (k + 1 < n_ann) ctext += "\t";
                }
                ctext += "\n";
            }
            write_text(cf, ctext);
            fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
        }
    }
    if (opt.subtrees_single > 1 && !missing.empty()) {                          // :973-990
        fprintf(stderr, "Computing the single subtree for added samples with %zu random leaves. \n\n", opt.subtrees_single);
        timer.start();
        for (size_t t = 0; t < num_trees; t++) {
            trees[t]->uncondense_leaves();
            write_single_subtree(*trees[t], missing, outdir, opt.subtrees_single, t, num_trees > 1);
        }
        fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
    }
    if (opt.subtrees_size > 1 && !missing.empty()) {                            // :992-1012
        fprintf(stderr, "Computing subtrees for added samples. \n\n");
        timer.start();
        for (size_t t = 0; t < num_trees; t++) {
            trees[t]->uncondense_leaves();
            write_sample_subtrees(*trees[t], missing, outdir, opt.subtrees_size, t, num_trees > 1);
        }
        fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
    }
    if (!low_confidence.empty()) {                                              // :1016-1021
        fprintf(stderr, "WARNING: Following samples had multiple possibilities of parsimony-optimal placements:\n");
        for (auto &l : low_confidence) fprintf(stderr, "%s\n", l.c_str());
    }
    if (!opt.save_mat.empty()) {                                                // :1024-1044
        timer.start();
        fprintf(stderr, "Saving mutation-annotated tree object to file (after condensing identical sequences) %s\n", opt.save_mat.c_str());
        if (num_trees > 1) fprintf(stderr, "WARNING: --multiple-placements option was used but only the first mutation-annotated tree object will be saved to file.\n");
        Tree &T = *trees[0];
        if (!T.condensed_nodes.empty()) T.uncondense_leaves();
        T.condense_leaves();
        std::string err;
        if (!save_mat(T, opt.save_mat, err)) { fprintf(stderr, "ERROR: %s\n", err.c_str()); return 1; }
        fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
    }
    return 0;
}

// ---------------------------------------------------------------------------
// --multiple-placements > 1 (usher_common.cpp:310-792 with max_trees > 1): every tree alive at the start of a
// sample's turn is searched; the sample goes to the first optimal node on that tree and, while the tree budget
// lasts, to each further optimal node on a copy of the tree as it was before the insertion (get_tree_copy).
// Every (sample, tree) pair is a search on a different tree, so this path flattens per search, as the reference
// re-expands per search; clade assignments are not produced in this mode (:598).
// ---------------------------------------------------------------------------
static int run_multi(const Options &opt, Tree &T0, std::vector<MissingSample> &missing, const std::vector<size_t> &indexes,
                     const Backend &be, uint64_t &tree_version) {
    Timer timer;
    std::vector<std::unique_ptr<Tree>> owned;
    std::vector<Tree *> trees{&T0};
    std::vector<std::string> low_confidence;   // (only filled when max_trees == 1, :457-459)
    auto be_fail = [&](const char *what) {
        fprintf(stderr, "ERROR: %s failed: %s\n", what, be.last_error ? be.last_error(be.ctx) : "?");
        return 1;
    };
    FILE *stats = fopen((opt.outdir + "/placement_stats.tsv").c_str(), "w");
    if (!stats) { fprintf(stderr, "ERROR: cannot write to %s\n", opt.outdir.c_str()); return 1; }
    FlatTree flat;
    for (size_t ii = 0; ii < indexes.size(); ii++) {
        MissingSample &ms = missing[indexes[ii]];
        const size_t num_trees = trees.size();
        for (size_t t_idx = 0; t_idx < num_trees; t_idx++) {
            timer.start();
            Tree *T = trees[t_idx];
            if (num_trees > 1) fprintf(stderr, "==Tree %zu=== \n", t_idx + 1);
            if (T->get_node(ms.name)) { fprintf(stderr, "WARNING: Sample %s already in the tree! Ignoring.\n\n", ms.name.c_str()); continue; }
            flat.build(*T);
            const size_t total_nodes = flat.bfs.size();
            FlatQueries q1;
            q1.add(ms.mutations);
            q1.finish();
            ugp_result r{};
            if (be.place(be.ctx, &flat.desc, ++tree_version, &q1.desc, &r) != 0) { fclose(stats); return be_fail("placement"); }
            int best = r.best_set_difference;
            size_t num_best = r.num_best;
            std::vector<uint32_t> tj(std::max<size_t>(1, std::min<size_t>(num_best, (size_t)1 << 20))), tc(1, 0);
            std::vector<uint8_t> th(tj.size());
            if (num_best > 1) {   // best_j_vec, ascending (:588), with node_has_unique of each
                if (be.ties(be.ctx, &flat.desc, tree_version, &q1.desc, (uint32_t)tj.size(), tj.data(), th.data(), tc.data()) != 0) { fclose(stats); return be_fail("tie listing"); }
            } else { tj[0] = r.best_j; th[0] = (uint8_t)r.best_has_unique; }
            fprintf(stderr, "Current tree size (#nodes): %zu\tSample name: %s\tParsimony score: %d\tNumber of parsimony-optimal placements: %zu\n", total_nodes, ms.name.c_str(), best, num_best);
            fprintf(stats, "%s\t%d\t%zu\t", ms.name.c_str(), best, num_best);
            if (num_best > 1) {
                if (num_best > opt.max_uncertainty) fprintf(stderr, "WARNING: Number of parsimony-optimal placements exceeds maximum allowed value (%u). Ignoring sample %s.\n", opt.max_uncertainty, ms.name.c_str());
                else if ((uint32_t)best <= opt.max_parsimony) fprintf(stderr, "WARNING: Multiple parsimony-optimal placements found. Placement done without high confidence.\n");
            }
            if ((uint32_t)best > opt.max_parsimony) fprintf(stderr, "WARNING: Parsimony score of the most parsimonious placement exceeds the maximum allowed value (%u). Ignoring sample %s.\n", opt.max_parsimony, ms.name.c_str());
            // a copy of the tree as it is now, for the 2nd, 3rd ... placement (:548-555)
            std::unique_ptr<Tree> curr_tree;
            if (num_best > 1 && num_trees < opt.max_trees) {
                curr_tree.reset(new Tree());
                std::string err;
                if (!copy_tree(*T, *curr_tree, err)) { fprintf(stderr, "ERROR: %s\n", err.c_str()); fclose(stats); return 1; }
            }
            if (num_best <= opt.max_uncertainty && (uint32_t)best <= opt.max_parsimony) {   // :583
                if (num_best > 1 && trees.size() <= opt.max_trees && num_best + trees.size() > opt.max_trees) {   // :592-597
                    if (num_best + trees.size() > (size_t)opt.max_trees + 1)
                        fprintf(stderr, "%zu parsimony-optimal placements found but total trees has already exceed the max possible value (%i)!\n", num_best, (int)opt.max_trees);
                    num_best = 1 + opt.max_trees - trees.size();
                }
                // node_has_unique[k] for k < num_best (:647 indexes the per-NODE flags with the loop counter): the flag
                // of breadth-first node k if that node ever matched or beat the running optimum of the in-order scan
                // (pass 1, :465-497) or is itself optimal (pass 2), else false (:379)
                std::vector<uint8_t> quirk(num_best, 0);
                if (num_best > 1) {
                    int run = (int)(ms.mutations.size() + T->root->mutations.size() + 1);   // :374
                    NodeVecs nv;
                    for (size_t k = 0; k < num_best && k < total_nodes; k++) {
                        node_vecs(flat.bfs[k], ms.mutations, nv);
                        if (nv.eligible && nv.set_difference <= run) { quirk[k] = nv.has_unique ? 1 : 0; run = std::min(run, nv.set_difference); }
                    }
                }
                std::vector<Node *> bfs = flat.bfs;
                for (size_t k = 0; k < num_best; k++) {
                    Node *best_node = flat.bfs[r.best_j];
                    bool best_has_unique = r.best_has_unique != 0;
                    if (num_best > 1) {
                        if (k == 0) fprintf(stderr, "Creating %zu additional tree(s) for %zu parsimony-optimal placements.\n", num_best - 1, num_best);
                        if (k > 0) {
                            owned.emplace_back(new Tree());
                            std::string err;
                            if (!copy_tree(*curr_tree, *owned.back(), err)) { fprintf(stderr, "ERROR: %s\n", err.c_str()); fclose(stats); return 1; }
                            trees.push_back(owned.back().get());
                            T = trees.back();
                            bfs = T->bfs();
                        }
                        best_node = bfs[tj[k]];
                        best_has_unique = quirk[k] != 0;
                    }
                    NodeVecs vec;
                    node_vecs(best_node, ms.mutations, vec);
                    if (!opt.no_add && !T->get_node(ms.name)) {
                        std::vector<Node *> touched;
                        insert_sample(*T, best_node, best_node->is_leaf() || best_has_unique, ms.name, vec.excess, touched);
                    }
                    if (!vec.imputed.empty()) {                                 // :767-781
                        fprintf(stderr, "Imputed mutations:\t");
                        for (size_t i = 0; i < vec.imputed.size(); i++) {
                            const char *sep = i + 1 < vec.imputed.size() ? ";" : "";
                            fprintf(stderr, "%i:%c%s", vec.imputed[i].position, nuc_char(vec.imputed[i].mut_nuc), sep);
                            fprintf(stats, "%i:%c%s", vec.imputed[i].position, nuc_char(vec.imputed[i].mut_nuc), sep);
                        }
                        fprintf(stderr, "\n");
                    }
                }
            }
            fputc('\n', stats);
            fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
        }
    }
    fclose(stats);
    return write_outputs(opt, trees, missing, low_confidence);
}

// The loaded tree as arrays, handed to the backend on a thread of its own while the VCF is read (driver.hpp)
struct Prebuilt {
    FlatTree flat;
    std::thread th;
    int rc = 0;
    double secs = 0;
    std::unique_ptr<FinalText> final_text;   // -n: the text of the final tree (= the tree as loaded), produced under everything else
};
Prebuilt *prebuild_start(const Options &opt, const Tree &T, const Backend &be) {
    // (-c changes the tree before anything is placed; -M works on copies of it)
    if (!be.prepare || !T.root || opt.collapse_tree || opt.max_trees > 1 || getenv("USHER_AMD_NO_PREBUILD")) return nullptr;
    Prebuilt *p = new Prebuilt();
    if (opt.no_add && !opt.print_scores && !opt.collapse_output_tree) {
        p->final_text.reset(new FinalText());
        p->final_text->start(T, opt.write_uncondensed);
    }
    const Backend b = be;
    p->th = std::thread([p, &T, b]() {
        const auto t0 = std::chrono::steady_clock::now();
        p->flat.build(T);
        p->rc = b.prepare(b.ctx, &p->flat.desc, 1);   // (version 1: run_usher's first)
        p->secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    });
    return p;
}
void prebuild_drop(Prebuilt *p) {
    if (!p) return;
    if (p->th.joinable()) p->th.join();
    delete p;
}

int run_usher(const Options &opt, Tree &T, std::vector<MissingSample> &missing, const Backend &be, Prebuilt *pre) {
    struct PreGuard { Prebuilt *&p; ~PreGuard() { prebuild_drop(p); p = nullptr; } } pre_guard{pre};
    // ---- option validation, usher_common.cpp:14-77
    if (opt.subtrees_size == 1) { fprintf(stderr, "ERROR: print-subtrees-size should be larger than 1\n"); return 1; }
    if ((int)opt.sort1 + (int)opt.sort2 + (int)opt.sort3 > 1) {
        fprintf(stderr, "ERROR: Can't use two or more of sort-before-placement-1, sort-before-placement-2 and sort-before-placement-3 simultaneously. Please specify only one.\n");
        return 1;
    }
    if (opt.sort1 || opt.sort2 || opt.sort3) {
        fprintf(stderr, "WARNING: Using experimental option %s\n", opt.sort1 ? "--sort-before-placement-1 (-s)" : opt.sort2 ? "--sort-before-placement-2 (-S)" : "--sort-before-placement-3 (-A)");
    } else if (opt.reverse_sort) {
        fprintf(stderr, "ERROR: Can't use reverse-sort without sorting options (sort-before-placement-1 or sort-before-placement-2 or sort-before-placement-3)\n");
        return 1;
    }
    if (opt.print_scores) {
        if (opt.max_trees > 1) { fprintf(stderr, "ERROR: cannot use --multiple-placements (-M) and --print_parsimony_scores (-p) options simulaneously.\n"); return 1; }
        if (opt.sort1 || opt.sort2 || opt.sort3 || opt.collapse_tree || opt.collapse_output_tree || opt.write_uncondensed || opt.subtrees_size > 0 || !opt.save_mat.empty())
            fprintf(stderr, "WARNING: --print-parsimony-scores-per-node is set. Will terminate without modifying the original tree.\n");
    }
    if (opt.max_trees == 0) { fprintf(stderr, "ERROR: Number of trees specified by --multiple-placements (-M) should be >= 1\n"); return 1; }
    if (opt.no_add && (opt.subtrees_size > 0 || opt.subtrees_single)) { fprintf(stderr, "ERROR: Sorry, cannot output subtrees when -n/--no-add is specified.\n"); return 1; }

    if (opt.retain_branch_len) fprintf(stderr, "Output newick files will retain branch lengths from the input tree (unspecified at branches modified during the placement).\n\n");
    else fprintf(stderr, "Output newick files will have branch lengths equal to the number of mutations of that branch.\n\n");

    std::string outdir = opt.outdir;
    struct stat sb;
    if (stat(outdir.c_str(), &sb) != 0) {
        fprintf(stderr, "Creating output directory.\n\n");
        if (mkdir(outdir.c_str(), 0777) != 0) { fprintf(stderr, "ERROR: cannot create %s\n", outdir.c_str()); return 1; }
    }
    Timer timer;
    if (opt.collapse_tree) {                                                    // :120-148
        timer.start();
        fprintf(stderr, "Collapsing input tree.\n");
        T.collapse_tree();
        fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
        fprintf(stderr, "Condensing identical sequences. \n");
        T.condense_leaves();
        const std::string fn = outdir + "/condensed-tree.nh";
        fprintf(stderr, "Writing condensed input tree to file %s\n", fn.c_str());
        write_text(fn, newick(T, T.root, true, true) + "\n");
        fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
    }
    fprintf(stderr, "Found %zu missing samples.\n\n", missing.size());
    std::vector<std::string> low_confidence;
    auto be_fail = [&](const char *what) {
        fprintf(stderr, "ERROR: %s failed: %s\n", what, be.last_error ? be.last_error(be.ctx) : "?");
        return 1;
    };

    if (opt.sort3) {                                                            // :150-159
        std::stable_sort(missing.begin(), missing.end(), [](const MissingSample &a, const MissingSample &b) { return a.num_ambiguous < b.num_ambiguous; });
        if (opt.reverse_sort) std::reverse(missing.begin(), missing.end());
    }

    uint64_t tree_version = 1;
    FlatTree flat;
    std::unique_ptr<FinalText> final_text = pre ? std::move(pre->final_text) : nullptr;   // (started right behind the load, or below)
    // The first flattening of a run: taken over from the thread that built it under the VCF read -- when the tree is still the one
    // that was loaded (no insertion yet) -- else built here.  false: the backend could not take the tree.
    auto build_flat = [&]() -> bool {
        if (pre && tree_version == 1) {
            if (pre->th.joinable()) pre->th.join();
            const int rc = pre->rc;
            if (getenv("USHER_AMD_PROFILE")) fprintf(stderr, "[usher-amd profile] tree -> arrays -> device on its own thread %.3f s (under the VCF read)\n", pre->secs);
            flat = std::move(pre->flat);
            flat.desc.parent = flat.parent.data(); flat.desc.mut_off = flat.mut_off.data(); flat.desc.mut_pos = flat.pos.data();
            flat.desc.mut_ref = flat.ref.data(); flat.desc.mut_par = flat.par.data(); flat.desc.mut_nuc = flat.nuc.data();
            delete pre; pre = nullptr;
            return rc == 0;
        }
        // (a flattening stamps the nodes with its index and epoch: never two at a time -- the thread's is over, and dropped, first)
        if (pre) { prebuild_drop(pre); pre = nullptr; }
        flat.build(T);
        return true;
    };
    if (!missing.empty()) {
        std::vector<size_t> indexes(missing.size());
        std::iota(indexes.begin(), indexes.end(), 0);
        // The reference sorts sample rows only in the -s/-S pre-pass (:203).  Samples whose rows are not strictly
        // increasing by position (and every sample, if the tree stores alleles that are not one base) are searched on
        // the host with the literal routine -- `odd`; the backend gets an ordered stand-in whose answer is ignored.
        if ((opt.sort1 || opt.sort2) && missing.size() > 1 && !opt.print_scores)
            for (auto &ms : missing) std::sort(ms.mutations.begin(), ms.mutations.end(), by_pos);   // :203
        const bool host_all = !tree_alleles_are_bases(T);
        std::vector<uint8_t> odd(missing.size(), 0);
        size_t n_odd = 0;
        for (size_t i = 0; i < missing.size(); i++) { odd[i] = host_all || !rows_in_order(missing[i].mutations); n_odd += odd[i]; }
        if (n_odd) fprintf(stderr, "NOTE: %zu sample(s) outside the preconditions of the GPU search (rows out of order / duplicated, or ambiguous tree alleles) are searched on the host.\n\n", n_odd);
        auto rows_of = [&](size_t i) -> std::vector<Mutation> { return odd[i] ? ordered_rows(missing[i].mutations) : missing[i].mutations; };
        HostSearch hs;

        const bool static_tree = opt.print_scores || opt.no_add;
        std::vector<ugp_result> batch_res;
        std::vector<int32_t> batch_scores;
        FlatQueries allq;
        for (size_t i = 0; i < missing.size(); i++) allq.add(rows_of(i));
        allq.finish();

        if (opt.print_scores) {                                                 // :176-185
            timer.start();
            const std::string fn = outdir + "/current-tree.nh";
            fprintf(stderr, "Writing current tree with internal nodes labelled to file %s \n", fn.c_str());
            write_text(fn, newick(T, T.root, true, true) + "\n");
            fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
        } else if ((opt.sort1 || opt.sort2) && missing.size() > 1) {            // :187-301
            timer.start();
            fprintf(stderr, "Computing parsimony scores and number of parsimony-optimal placements for new samples and using them to sort the samples.\n");
            if (!build_flat()) return be_fail("flattening");
            std::vector<ugp_result> r(missing.size());
            if (!host_all && be.place(be.ctx, &flat.desc, tree_version, &allq.desc, r.data()) != 0) return be_fail("placement");
            for (size_t i = 0; i < missing.size(); i++) if (odd[i]) {
                host_search(flat.bfs, missing[i].mutations, false, hs);
                r[i].best_set_difference = hs.best; r[i].num_best = (uint32_t)hs.num_best;
            }
            auto key1 = [&](size_t i) { return std::make_pair((int64_t)r[i].best_set_difference, (int64_t)r[i].num_best); };
            auto key2 = [&](size_t i) { return std::make_pair((int64_t)r[i].num_best, (int64_t)r[i].best_set_difference); };
            if (opt.sort1) std::stable_sort(indexes.begin(), indexes.end(), [&](size_t a, size_t b) { return key1(a) < key1(b); });
            else std::stable_sort(indexes.begin(), indexes.end(), [&](size_t a, size_t b) { return key2(a) < key2(b); });
            if (opt.reverse_sort) std::reverse(indexes.begin(), indexes.end());
            fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
        }
        if (!opt.print_scores) fprintf(stderr, "Adding missing samples to the tree.\n");
        if (opt.max_trees > 1) return run_multi(opt, T, missing, indexes, be, tree_version);

        if (static_tree) {   // the tree never changes: one batch call serves every sample
            if (!flat.epoch && !build_flat()) return be_fail("flattening");
            // -n: what will be written as the final tree is this tree -- its text is produced under the placement and the statistics
            if (!final_text && opt.no_add && !opt.print_scores && !opt.collapse_output_tree && !getenv("USHER_AMD_NO_PREBUILD")) {
                final_text.reset(new FinalText());
                final_text->start(T, opt.write_uncondensed);
            }
            batch_res.resize(missing.size());
            if (!host_all && be.place(be.ctx, &flat.desc, tree_version, &allq.desc, batch_res.data()) != 0) return be_fail("placement");
        }
        // -p: the samples x nodes score matrix is produced in slabs of at most ~1 GiB
        size_t slab_base = 0, slab_len = 0;
        std::vector<int32_t> host_scores;
        auto scores_row = [&](size_t s) -> const int32_t * {
            const size_t n = flat.bfs.size();
            if (odd[s]) {
                HostSearch h;
                host_search(flat.bfs, missing[s].mutations, true, h);
                host_scores.swap(h.scores);
                return host_scores.data();
            }
            if (s >= slab_base + slab_len || s < slab_base) {
                slab_base = s;
                slab_len = std::min(missing.size() - s, std::max<size_t>(1, ((size_t)1 << 28) / std::max<size_t>(n, 1)));
                FlatQueries part;
                for (size_t k = s; k < s + slab_len; k++) part.add(rows_of(k));
                part.finish();
                batch_scores.resize(slab_len * n);
                if (be.scores(be.ctx, &flat.desc, tree_version, &part.desc, batch_scores.data()) != 0) return nullptr;
            }
            return batch_scores.data() + (s - slab_base) * n;
        };

        // Add-mode: the reference re-searches the whole tree for every sample because the previous
        // insertion changed it (usher_common.cpp:342).  Here the next samples are placed in one batch on the
        // tree as it is (`flat`), together with their tie lists, and each answer is re-derived on the tree as
        // it has become: an insertion leaves the root-path mutation set of every other node unchanged, so
        // only the nodes created or rewritten since (`touched`) can have a different cost, eligibility or
        // has_unique; they are evaluated on the host with the literal per-node routine and merged with the
        // untouched part of the batched tie set, and the winner is picked with the current leaf counts and
        // breadth-first order.  The tree is flattened and placed again only when `touched` grows past
        // max_touched, when a tie list was too long to keep, or when every batched optimum was rewritten
        // and no touched node is at least as good.
        // USHER_AMD_PROFILE=1: where the add mode spends its time (printed once at the end)
        struct Prof { double build = 0, place = 0, ties = 0, touched = 0, vecs = 0, insert = 0, book = 0, loop = 0; size_t flats = 0, batches = 0, evals = 0; } prof;
        const double t_loop0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        auto now_s = []() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        std::vector<ugp_result> spec_res;
        size_t spec_base = 0, spec_len = 0, spec_next = 64;   // batch length adapts to how long answers survive
        bool have_spec = false;
        uint64_t flat_version = tree_version;
        std::vector<Node *> touched;
        std::unordered_set<const Node *> touched_set;
        std::unordered_map<const Node *, size_t> added_leaves;   // leaves gained below a node since `flat` was built
        std::vector<std::vector<std::pair<uint32_t, uint8_t>>> spec_ties;   // per batched sample with 1 < num_best <= kTieCap
        const uint32_t kTieCap = 256;
        // |touched| at which the tree is flattened again.  Re-deriving an answer costs c ~ 0.05 us per touched node
        // (closed form; ~1 us with the literal routine), a flatten + batch placement costs t_redo and is amortised
        // over ~|touched|/3 insertions, so the total is least near sqrt(3 t_redo / c): ~250 for a 1k-node tree,
        // ~14,000 for a 10M-node one (t_redo ~ 3 s).
        size_t max_touched = 64;
        const bool fixed_cap = getenv("USHER_AMD_MAX_TOUCHED") != nullptr;
        if (fixed_cap) max_touched = (size_t)atoll(getenv("USHER_AMD_MAX_TOUCHED"));
        NodeVecs probe;
        TouchedCache tcache;
        DenseSample dense;
        const bool literal_touched = getenv("USHER_AMD_LITERAL_TOUCHED") != nullptr;   // cross-check switch: literal routine instead of the closed form
        auto leaves_now = [&](const Node *n) -> size_t {
            size_t v = 0;
            if (flat.has(n)) v = flat.leaves[n->flat_index];
            auto a = added_leaves.find(n);
            return a == added_leaves.end() ? v : v + a->second;
        };
        struct Tie { Node *n; bool hu; };
        std::vector<Tie> tie_now;   // the tie set of the current sample on the tree as it is now (when it had to be rebuilt)
        // ---- Add mode with the edits on the device (Backend::update, touched_*; include/usher_amd.h "add mode").  The tree is
        // flattened ONCE.  Every node created or rewritten since becomes a record on the device; rewritten flattened nodes leave
        // the candidate set of the device's searches.  Samples are taken in batches (one search of the flattened tree + one
        // scoring of all live records per batch) and in rounds within a batch: the records of a round's insertions are uploaded
        // and merged into the batch's results when the round ends, the nodes touched within the round are evaluated here.  The
        // answer for a sample is the merge of three exact parts -- flattened nodes that are still what they were (device),
        // records (device), this round's nodes (host) -- or, when a part has lost the holders of its minimum to later edits
        // and that could matter, the part is asked again for this one sample.
        struct DevAdd {
            bool on = false, flat_done = false;
            size_t batch = 4096, round = 64;
            size_t base = 0, len = 0;       // the open batch: indexes [base, base + len)
            std::vector<ugp_result> res;
            std::vector<std::vector<std::pair<uint32_t, uint8_t>>> ties;
            size_t rbase = 0, rlen = 0;     // record results fetched for indexes [rbase, rbase + rlen)
            std::vector<int32_t> t_best;
            std::vector<uint32_t> t_cnt, t_ids;
            std::vector<uint8_t> t_hu;
            std::vector<Node *> rec_node;   // record id -> node
            std::vector<uint8_t> rec_dead;
            std::unordered_map<const Node *, uint32_t> rec_of;   // live record of a node
            std::vector<Node *> pending;    // touched since the last upload: evaluated on the host until then
            std::unordered_set<const Node *> pending_set;
            std::vector<uint32_t> pending_retired;
            std::unordered_set<const Node *> since_batch;   // flattened nodes rewritten since the open batch was searched
            size_t n_flush = 0, n_replace = 0, n_rescore = 0, n_host_all = 0, n_batches = 0;
            double t_flush = 0, t_fetch = 0, t_batch = 0;
        } dev;
        uint32_t kTCap = 64;   // list entries per sample the device keeps (USHER_AMD_TCAP: the tests shorten it to reach the truncated-list path)
        if (const char *e = getenv("USHER_AMD_TCAP")) kTCap = (uint32_t)std::max(1, std::min(64, atoi(e)));
        dev.on = !static_tree && be.update && be.touched_open && be.touched_score && be.touched_rescore && be.touched_fetch && be.ties && !fixed_cap &&
                 !literal_touched && !getenv("USHER_AMD_HOST_ADDMODE");
        if (const char *e = getenv("USHER_AMD_BATCH")) dev.batch = (size_t)std::max(1, atoi(e));
        if (const char *e = getenv("USHER_AMD_ROUND")) dev.round = (size_t)std::max(1, atoi(e));
        uint64_t flat_version_dev = 0;
        // records of the nodes touched since the last upload -> device; merged into the open batch's results from index ii on
        auto dev_flush = [&](size_t ii) -> bool {
            if (dev.pending.empty() && dev.pending_retired.empty()) return true;
            const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            std::vector<uint32_t> flat_j, n_path;
            std::vector<uint8_t> flags, al, pv, rf;
            std::vector<uint64_t> off{0};
            std::vector<int32_t> pos;
            for (Node *x : dev.pending) {
                const TouchedInfo &ti = tcache.info.at(x);
                flat_j.push_back(flat.has(x) ? x->flat_index : UINT32_MAX);
                flags.push_back((uint8_t)((x->is_leaf() ? UGP_T_LEAF : 0u) | (ti.masked ? UGP_T_MASKED : 0u)));
                n_path.push_back((uint32_t)ti.path.size());
                for (const auto &e : ti.path) { pos.push_back(e.pos); al.push_back(e.allele); pv.push_back(0); rf.push_back(e.ref); }
                for (const auto &o : ti.own) { pos.push_back(o.pos); al.push_back(o.mut); pv.push_back(o.prev); rf.push_back(o.ref); }
                off.push_back(pos.size());
            }
            ugp_touched t{};
            t.n = dev.pending.size(); t.flat_j = flat_j.data(); t.flags = flags.data(); t.n_path = n_path.data(); t.ent_off = off.data();
            t.pos = pos.data(); t.allele = al.data(); t.prev = pv.data(); t.ref = rf.data();
            uint32_t first = 0;
            if (be.update(be.ctx, &t, dev.pending_retired.data(), dev.pending_retired.size(), &first) != 0) return false;
            if (first != dev.rec_node.size()) return false;   // (ids count up across calls: ours and the device's must agree)
            for (Node *x : dev.pending) { dev.rec_of[x] = (uint32_t)dev.rec_node.size(); dev.rec_node.push_back(x); dev.rec_dead.push_back(0); }
            dev.pending.clear(); dev.pending_set.clear(); dev.pending_retired.clear();
            if (ii < dev.base + dev.len && be.touched_score(be.ctx, first, ii - dev.base) != 0) return false;
            dev.rlen = 0;   // (what was fetched is out of date)
            dev.n_flush++;
            dev.t_flush += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
            return true;
        };
        auto dev_fetch = [&](size_t ii) -> bool {
            const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            dev.rbase = ii; dev.rlen = std::min(dev.round, dev.base + dev.len - ii);
            dev.t_best.resize(dev.rlen); dev.t_cnt.resize(dev.rlen); dev.t_ids.resize(dev.rlen * kTCap); dev.t_hu.resize(dev.rlen * kTCap);
            const bool ok = be.touched_fetch(be.ctx, ii - dev.base, dev.rlen, kTCap, dev.t_best.data(), dev.t_cnt.data(), dev.t_ids.data(), dev.t_hu.data()) == 0;
            dev.t_fetch += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
            return ok;
        };
        const uint32_t kTieCapDev = 256;
        auto dev_batch = [&](size_t ii) -> bool {   // the next batch: searched on the flattened tree (rewritten nodes excluded), scored against every live record
            const double t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            if (!dev.flat_done) {
                if (!build_flat()) return false;
                prof.build += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0; prof.flats++;
                flat_version_dev = tree_version;
                dev.flat_done = true;
                // (samples searched on the host in front of the first batch -- rows out of order -- have been inserted already: their
                // nodes are part of this flattening, not edits of it)
                added_leaves.clear(); tcache.info.clear();
                dev.pending.clear(); dev.pending_set.clear(); dev.pending_retired.clear(); dev.since_batch.clear();
            }
            dev.base = ii; dev.len = 0;           // (no batch is open while the pending records go up)
            if (dev.flat_done && dev.n_batches && !dev_flush(ii)) return false;
            dev.len = std::min(dev.batch, indexes.size() - ii);
            FlatQueries rest;
            for (size_t k = ii; k < ii + dev.len; k++) rest.add(rows_of(indexes[k]));
            rest.finish();
            dev.res.assign(dev.len, ugp_result{});
            const double tp = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
            if (be.place(be.ctx, &flat.desc, flat_version_dev, &rest.desc, dev.res.data()) != 0) return false;
            prof.place += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - tp; prof.batches++;
            if (dev.n_batches == 0 && !dev_flush(ii + dev.len)) return false;   // (first batch: the handle exists only now; nothing is pending yet)
            dev.ties.assign(dev.len, {});
            {
                FlatQueries tq;
                std::vector<size_t> who;
                for (size_t k = 0; k < dev.len; k++)
                    if (!odd[indexes[ii + k]] && dev.res[k].num_best > 1 && dev.res[k].num_best <= kTieCapDev) { who.push_back(k); tq.add(missing[indexes[ii + k]].mutations); }
                if (!who.empty()) {
                    tq.finish();
                    std::vector<uint32_t> tj(who.size() * (size_t)kTieCapDev), tc(who.size());
                    std::vector<uint8_t> th(who.size() * (size_t)kTieCapDev);
                    const double tt = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
                    if (be.ties(be.ctx, &flat.desc, flat_version_dev, &tq.desc, kTieCapDev, tj.data(), th.data(), tc.data()) != 0) return false;
                    prof.ties += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - tt;
                    for (size_t w = 0; w < who.size(); w++) {
                        if (tc[w] != dev.res[who[w]].num_best) continue;   // (cannot happen; the list is then asked for again when the sample's turn comes)
                        for (uint32_t k = 0; k < tc[w]; k++) dev.ties[who[w]].push_back({tj[w * kTieCapDev + k], th[w * kTieCapDev + k]});
                    }
                }
            }
            if (be.touched_open(be.ctx, &rest.desc) != 0) return false;
            dev.since_batch.clear();
            dev.rlen = 0;
            dev.n_batches++;
            dev.t_batch += std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t0;
            return true;
        };
        FILE *stats = fopen((outdir + "/placement_stats.tsv").c_str(), "w");
        if (!stats) { fprintf(stderr, "ERROR: cannot write to %s\n", outdir.c_str()); return 1; }
        FILE *scores_file = nullptr;
        for (size_t ii = 0; ii < indexes.size(); ii++) {                        // :310
            timer.start();
            const size_t s = indexes[ii];
            MissingSample &ms = missing[s];
            if (T.get_node(ms.name)) { fprintf(stderr, "WARNING: Sample %s already in the tree! Ignoring.\n\n", ms.name.c_str()); continue; }
            const size_t total_nodes = T.all_nodes.size();                      // == bfs.size() of the current tree (:343)
            if (opt.print_scores && s == 0) {                                   // :331-340
                const std::string fn = outdir + "/parsimony-scores.tsv";
                fprintf(stderr, "\nNow computing branch parsimony scores for adding the missing samples at each of the %zu nodes in the existing tree without modifying the tree.\n", total_nodes);
                fprintf(stderr, "The branch parsimony scores will be written to file %s\n\n", fn.c_str());
                scores_file = fopen(fn.c_str(), "w");
                if (scores_file) fprintf(scores_file, "#Sample\tTree node\tParsimony score\tOptimal (y/n)\tParsimony-increasing mutations (for optimal nodes)\n");
            }
            ugp_result r;
            bool patched = false, patched_hu = false;
            int patched_best = 0;
            Node *patched_node = nullptr;
            if (odd[s]) {   // searched here, on the tree as it is now
                if (static_tree) host_search(flat.bfs, ms.mutations, false, hs);
                else host_search(T.bfs(), ms.mutations, false, hs);
                tie_now.clear();
                for (auto &t : hs.ties) tie_now.push_back({t.first, t.second});
                patched = true; patched_best = hs.best; patched_node = hs.best_node; patched_hu = hs.best_has_unique;
                r = ugp_result{};
            } else if (static_tree) r = batch_res[s];
            else if (dev.on) {
                if (ii >= dev.base + dev.len && !dev_batch(ii)) { fclose(stats); return be_fail("placement (add mode)"); }
                if (dev.rlen == 0 || ii >= dev.rbase + dev.rlen) {
                    if (!dev_flush(ii) || !dev_fetch(ii)) { fclose(stats); return be_fail("record scoring (add mode)"); }
                }
                const size_t k = ii - dev.base;
                ugp_result fr = dev.res[k];
                std::vector<std::pair<uint32_t, uint8_t>> own_list;
                const std::vector<std::pair<uint32_t, uint8_t>> *flst = &dev.ties[k];
                auto list_for = [&](ugp_result &x) -> bool {   // the tie list of a flattened-tree answer that came without one
                    FlatQueries q1;
                    q1.add(rows_of(s));
                    q1.finish();
                    // The device lists the ties of the search as it stands NOW -- nodes rewritten since the batch was searched have left
                    // its candidate set, and when they held the old minimum the list belongs to a higher cost.  The answer the list is
                    // merged under is therefore taken again first, at the same exclusion state as the list.
                    if (be.place(be.ctx, &flat.desc, flat_version_dev, &q1.desc, &x) != 0) return false;
                    if (x.num_best <= 1) { own_list.clear(); flst = &own_list; return true; }
                    const uint32_t cap = x.num_best;
                    std::vector<uint32_t> tj(cap), tc(1);
                    std::vector<uint8_t> th(cap);
                    if (be.ties(be.ctx, &flat.desc, flat_version_dev, &q1.desc, cap, tj.data(), th.data(), tc.data()) != 0) return false;
                    own_list.clear();
                    for (uint32_t i = 0; i < std::min(tc[0], cap); i++) own_list.push_back({tj[i], th[i]});
                    flst = &own_list;
                    return true;
                };
                if (!dev.pending.empty()) dense.set(ms.mutations);
                bool dense_set = !dev.pending.empty();
                bool host_all = false;
                int best_now = 0;
                for (int attempt = 0;; attempt++) {
                    const size_t kk = ii - dev.rbase;
                    // flattened nodes that are still what they were (rewritten ones are excluded on the device as of the search; those
                    // rewritten since are dropped here)
                    std::vector<Tie> F, Dv, H;
                    if (fr.num_best > 1 && flst->empty() && !list_for(fr)) { fclose(stats); return be_fail("tie listing"); }
                    if (fr.num_best == 1) { Node *bn = flat.bfs[fr.best_j]; if (!dev.since_batch.count(bn)) F.push_back({bn, fr.best_has_unique != 0}); }
                    else for (const auto &e : *flst) if (!dev.since_batch.count(flat.bfs[e.first])) F.push_back({flat.bfs[e.first], e.second != 0});
                    const int f_best = fr.best_set_difference;
                    const bool f_exact = !F.empty();            // else: the remaining flattened nodes cost more than f_best
                    // records on the device
                    int d_best = host_all ? INT32_MAX : dev.t_best[kk];
                    int d_state = 0;                            // 0 none, 1 exact, 2 truncated list, 3 every holder of the minimum was rewritten since
                    if (d_best != INT32_MAX) {
                        const uint32_t cnt = dev.t_cnt[kk];
                        for (uint32_t i = 0; i < std::min(cnt, kTCap); i++) {
                            const uint32_t id = dev.t_ids[kk * kTCap + i];
                            if (!dev.rec_dead[id]) Dv.push_back({dev.rec_node[id], dev.t_hu[kk * kTCap + i] != 0});
                        }
                        d_state = cnt > kTCap ? 2 : (Dv.empty() ? 3 : 1);
                    }
                    // this round's nodes (and, after a truncated list, every live record) on the host
                    int h_best = INT32_MAX;
                    const double te = now_s();
                    auto host_eval = [&](Node *tn) {
                        bool el, hu; int cost;
                        if (tn->is_root()) { node_vecs(tn, ms.mutations, probe); el = probe.eligible; hu = probe.has_unique; cost = probe.set_difference; }
                        else tcache.eval(tn, dense, el, cost, hu);
                        if (!el) return;
                        if (cost < h_best) { h_best = cost; H.clear(); }
                        if (cost == h_best) H.push_back({tn, hu});
                    };
                    if ((host_all || !dev.pending.empty()) && !dense_set) { dense.set(ms.mutations); dense_set = true; }
                    for (Node *tn : dev.pending) host_eval(tn);
                    prof.evals += dev.pending.size();
                    if (host_all) { for (size_t id = 0; id < dev.rec_node.size(); id++) if (!dev.rec_dead[id]) host_eval(dev.rec_node[id]); prof.evals += dev.rec_node.size(); }
                    prof.touched += now_s() - te;
                    int km = h_best;                            // the minimum over the parts that are exact
                    if (f_exact) km = std::min(km, f_best);
                    if (d_state == 1) km = std::min(km, d_best);
                    const bool need_f = !f_exact && km > f_best;                                       // an unlisted flattened node may tie or win
                    const bool need_d = (d_state == 2 && km >= d_best) || (d_state == 3 && km > d_best);   // an unlisted record may tie or win
                    if ((!need_f && !need_d) || attempt >= 3) {
                        if (need_f || need_d) { fclose(stats); fprintf(stderr, "ERROR: add mode: the answer for sample %s could not be re-derived\n", ms.name.c_str()); return 1; }
                        best_now = km;
                        tie_now.clear();
                        if (f_exact && f_best == km) tie_now.insert(tie_now.end(), F.begin(), F.end());
                        if (d_state == 1 && d_best == km) tie_now.insert(tie_now.end(), Dv.begin(), Dv.end());
                        if (h_best == km) tie_now.insert(tie_now.end(), H.begin(), H.end());
                        break;
                    }
                    // ask again, for this one sample: everything touched so far goes to the device first (records, exclusions)
                    if (!dev_flush(ii)) { fclose(stats); return be_fail("record upload (add mode)"); }
                    if (need_f) {
                        FlatQueries q1;
                        q1.add(rows_of(s));
                        q1.finish();
                        if (be.place(be.ctx, &flat.desc, flat_version_dev, &q1.desc, &fr) != 0) { fclose(stats); return be_fail("placement (add mode)"); }
                        own_list.clear(); flst = &own_list;
                        dev.n_replace++;
                    }
                    if (need_d) {
                        if (d_state == 2) { host_all = true; dev.n_host_all++; }
                        else { if (be.touched_rescore(be.ctx, ii - dev.base) != 0) { fclose(stats); return be_fail("record scoring (add mode)"); } dev.n_rescore++; }
                    }
                    if (!dev_fetch(ii)) { fclose(stats); return be_fail("record scoring (add mode)"); }
                }
                if (dense_set) dense.clear(ms.mutations);
                if (tie_now.empty()) { fclose(stats); fprintf(stderr, "ERROR: add mode: no candidate for sample %s\n", ms.name.c_str()); return 1; }
                const Tie *w = &tie_now[0];
                size_t w_leaves = leaves_now(w->n);
                for (size_t t = 1; t < tie_now.size(); t++) {
                    const size_t l = leaves_now(tie_now[t].n);
                    if (l > w_leaves || (l == w_leaves && bfs_before(w->n, tie_now[t].n))) { w = &tie_now[t]; w_leaves = l; }
                }
                patched = true; patched_best = best_now; patched_node = w->n; patched_hu = w->hu;
                r = ugp_result{};
            } else {
                // need_flat: flatten the current tree again; need_batch: place the next batch of samples (on the
                // tree as flattened last -- not necessarily the current one)
                bool need_flat = !have_spec || touched.size() > max_touched;
                bool need_batch = need_flat || ii >= spec_base + spec_len;
                for (int attempt = 0; attempt < 2; attempt++) {
                    if (need_batch) {
                        const auto t_redo0 = std::chrono::steady_clock::now();
                        if (need_flat) {
                            const double tb = now_s();
                            if (!build_flat()) { fclose(stats); return be_fail("flattening"); }
                            prof.build += now_s() - tb; prof.flats++;
                            flat_version = tree_version;
                            touched.clear();
                            touched_set.clear();
                            tcache.info.clear();
                            added_leaves.clear();
                        }
                        if (have_spec) spec_next = (attempt == 0 && ii >= spec_base + spec_len) ? std::min<size_t>(2 * spec_len, 4096)   // consumed whole: grow
                                                                                                  : std::max<size_t>(spec_len / 2, 16);      // cut short: shrink
                        spec_len = std::min(spec_next, indexes.size() - ii);
                        FlatQueries rest;
                        for (size_t k = ii; k < ii + spec_len; k++) rest.add(rows_of(indexes[k]));
                        rest.finish();
                        spec_res.assign(spec_len, ugp_result{});
                        const double tp = now_s();
                        if (be.place(be.ctx, &flat.desc, flat_version, &rest.desc, spec_res.data()) != 0) { fclose(stats); return be_fail("placement"); }
                        prof.place += now_s() - tp; prof.batches++;
                        // tie lists of the batch (needed to re-derive a tied answer after later insertions)
                        spec_ties.assign(spec_len, {});
                        if (!opt.no_add && be.ties) {
                            FlatQueries tq;
                            std::vector<size_t> who;
                            for (size_t k = touched.empty() ? 1 : 0; k < spec_len; k++)   // (on a fresh tree sample 0 is consumed as it is)
                                if (!odd[indexes[ii + k]] && spec_res[k].num_best > 1 && spec_res[k].num_best <= kTieCap) { who.push_back(k); tq.add(missing[indexes[ii + k]].mutations); }
                            if (!who.empty()) {
                                tq.finish();
                                std::vector<uint32_t> tj(who.size() * (size_t)kTieCap), tc(who.size());
                                std::vector<uint8_t> th(who.size() * (size_t)kTieCap);
                                const double tt = now_s();
                                if (be.ties(be.ctx, &flat.desc, flat_version, &tq.desc, kTieCap, tj.data(), th.data(), tc.data()) != 0) { fclose(stats); return be_fail("tie listing"); }
                                prof.ties += now_s() - tt;
                                for (size_t w = 0; w < who.size(); w++) {
                                    if (tc[w] != spec_res[who[w]].num_best) continue;   // (cannot happen; without a full list the sample is searched again)
                                    for (uint32_t k = 0; k < tc[w]; k++) spec_ties[who[w]].push_back({tj[w * kTieCap + k], th[w * kTieCap + k]});
                                }
                            }
                        }
                        if (!fixed_cap && need_flat) {
                            const double t_redo = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_redo0).count();
                            max_touched = (size_t)std::min(65536.0, std::max(16.0, std::sqrt(3.0 * t_redo / (literal_touched ? 1e-6 : 5e-8))));
                        }
                        spec_base = ii;
                        have_spec = true;
                    }
                    bool redo = false;
                    tie_now.clear();
                    patched = false;
                    r = spec_res[ii - spec_base];
                    if (!touched.empty()) {
                        // Rebuild the answer on the current tree from the batched one: the nodes that were
                        // optimal then and have not been touched keep their cost; touched and new nodes are
                        // evaluated here; the winner follows the reference's rule (more leaves, then the later
                        // breadth-first index, usher_mapper.cpp:476-497) with today's leaf counts and order.
                        const auto &bt = spec_ties[ii - spec_base];
                        if (r.num_best > 1 && bt.empty()) redo = true;   // tie list not kept (too long)
                        int base_best = r.best_set_difference;
                        if (!redo) {
                            if (r.num_best == 1) {
                                Node *bn = flat.bfs[r.best_j];
                                if (!touched_set.count(bn)) tie_now.push_back({bn, r.best_has_unique != 0});
                            } else {
                                for (const auto &e : bt)
                                    if (!touched_set.count(flat.bfs[e.first])) tie_now.push_back({flat.bfs[e.first], e.second != 0});
                            }
                            int m_best = INT32_MAX;
                            std::vector<Tie> m_ties;
                            const double te = now_s();
                            prof.evals += touched.size();
                            if (!literal_touched) dense.set(ms.mutations);
                            for (Node *tn : touched) {
                                bool el, hu; int cost;
                                if (literal_touched || tn->is_root()) { node_vecs(tn, ms.mutations, probe); el = probe.eligible; hu = probe.has_unique; cost = probe.set_difference; }
                                else tcache.eval(tn, dense, el, cost, hu);
                                if (!el) continue;
                                if (cost < m_best) { m_best = cost; m_ties.clear(); }
                                if (cost == m_best) m_ties.push_back({tn, hu});
                            }
                            if (!literal_touched) dense.clear(ms.mutations);
                            prof.touched += now_s() - te;
                            if (tie_now.empty()) {
                                // every node that was optimal has been rewritten: the untouched nodes are only known
                                // to cost more than base_best, which decides the matter only if a touched node does not
                                if (m_best <= base_best) tie_now = m_ties; else redo = true;
                            } else if (m_best < base_best) tie_now = m_ties;
                            else if (m_best == base_best) tie_now.insert(tie_now.end(), m_ties.begin(), m_ties.end());
                            if (!redo) {
                                const Tie *w = &tie_now[0];
                                size_t w_leaves = leaves_now(w->n);
                                for (size_t k = 1; k < tie_now.size(); k++) {
                                    const size_t l = leaves_now(tie_now[k].n);
                                    if (l > w_leaves || (l == w_leaves && bfs_before(w->n, tie_now[k].n))) { w = &tie_now[k]; w_leaves = l; }
                                }
                                patched = true;
                                patched_best = std::min(base_best, m_best);
                                patched_node = w->n;
                                patched_hu = w->hu;
                            }
                        }
                    }
                    if (!redo) break;
                    need_flat = need_batch = true;   // this answer cannot be re-derived: search the current tree
                }
            }
            const int best = patched ? patched_best : r.best_set_difference;
            const size_t num_best = patched ? tie_now.size() : r.num_best;
            Node *best_node = patched ? patched_node : flat.bfs[r.best_j];
            const bool best_has_unique = patched ? patched_hu : r.best_has_unique != 0;

            if (!opt.print_scores) {                                            // :451-469
                fprintf(stderr, "Current tree size (#nodes): %zu\tSample name: %s\tParsimony score: %d\tNumber of parsimony-optimal placements: %zu\n", total_nodes, ms.name.c_str(), best, num_best);
                fprintf(stats, "%s\t%d\t%zu\t", ms.name.c_str(), best, num_best);
                if (num_best > 1) {
                    low_confidence.push_back(ms.name);
                    if (num_best > opt.max_uncertainty) fprintf(stderr, "WARNING: Number of parsimony-optimal placements exceeds maximum allowed value (%u). Ignoring sample %s.\n", opt.max_uncertainty, ms.name.c_str());
                    else if ((uint32_t)best <= opt.max_parsimony) fprintf(stderr, "WARNING: Multiple parsimony-optimal placements found. Placement done without high confidence.\n");
                }
                if ((uint32_t)best > opt.max_parsimony) fprintf(stderr, "WARNING: Parsimony score of the most parsimonious placement exceeds the maximum allowed value (%u). Ignoring sample %s.\n", opt.max_parsimony, ms.name.c_str());
            } else {
                fprintf(stderr, "Missing sample: %s\t Best parsimony score: %d\tNumber of parsimony-optimal placements: %zu\n", ms.name.c_str(), best, num_best);
            }

            NodeVecs vec;
            if (opt.print_scores) {                                             // :557-578
                const int32_t *sc = scores_row(s);
                if (!sc) { fclose(stats); return be_fail("per-node scoring"); }
                for (size_t k = 0; k < total_nodes && scores_file; k++) {
                    const bool optimal = sc[k] == best;
                    fprintf(scores_file, "%s\t%s\t%d\t\t%c\t", ms.name.c_str(), flat.bfs[k]->id.c_str(), sc[k], optimal ? 'y' : 'n');
                    if (optimal) {
                        if (sc[k] == 0) fprintf(scores_file, "*");
                        node_vecs(flat.bfs[k], ms.mutations, vec);
                        for (int i = 0; i < sc[k] && (size_t)i < vec.excess.size(); i++)   // first `score` entries of the excess vector (:565-572)
                            fprintf(scores_file, "%s%s", vec.excess[i].str().c_str(), i + 1 < sc[k] ? "," : "");
                    } else fprintf(scores_file, "N/A");
                    fprintf(scores_file, "\n");
                }
            } else if (num_best <= opt.max_uncertainty && (uint32_t)best <= opt.max_parsimony) {   // :583
                const size_t n_ann = T.num_annotations();
                if (n_ann > 0) {                                                // clade assignment, :601-619
                    FlatQueries q1;
                    q1.add(ms.mutations);
                    q1.finish();
                    const uint32_t cap = (uint32_t)std::min<size_t>(num_best, 1u << 20);
                    std::vector<uint32_t> tj(cap), tc(1);
                    std::vector<uint8_t> th(cap);
                    size_t nt = 1;
                    std::vector<Tie> tl;
                    if (patched) tl = tie_now;   // rebuilt on the current tree above
                    else if (num_best > 1) {      // `flat` is current here
                        if (be.ties(be.ctx, &flat.desc, static_tree ? tree_version : flat_version, &q1.desc, cap, tj.data(), th.data(), tc.data()) != 0) { fclose(stats); return be_fail("tie listing"); }
                        nt = std::min<size_t>(tc[0], cap);
                        for (size_t k = 0; k < nt; k++) tl.push_back({flat.bfs[tj[k]], th[k] != 0});
                    } else tl.push_back({best_node, best_has_unique});
                    nt = tl.size();
                    ms.clade_assignments.assign(n_ann, {});
                    ms.best_clade_assignment.assign(n_ann, "");
                    for (size_t c = 0; c < n_ann; c++) {
                        for (size_t k = 0; k < nt; k++) {
                            Node *nd = tl[k].n;
                            const bool include_self = !nd->is_leaf() && !tl[k].hu;
                            std::string ca = T.clade_assignment(nd, c, include_self);
                            if (nd == best_node) ms.best_clade_assignment[c] = ca;
                            ms.clade_assignments[c].push_back(std::move(ca));
                        }
                        std::sort(ms.clade_assignments[c].begin(), ms.clade_assignments[c].end());
                    }
                }
                const double tv0 = now_s();
                node_vecs(best_node, ms.mutations, vec);                        // pass 2 for the winner, :426-449
                prof.vecs += now_s() - tv0;
                if (!opt.no_add) {
                    const size_t t0 = touched.size();
                    const size_t best_leaves = leaves_now(best_node);
                    const double ti0 = now_s();
                    insert_sample(T, best_node, best_node->is_leaf() || best_has_unique, ms.name, vec.excess, touched);
                    prof.insert += now_s() - ti0;
                    tree_version++;
                    const double tb0 = now_s();
                    struct Book { double &acc, t0; std::function<double()> now; ~Book() { acc += now() - t0; } } book{prof.book, tb0, now_s};
                    if (!static_tree) {   // bookkeeping for re-deriving later batched answers on the changed tree
                        Node *leaf = nullptr;
                        size_t keep = t0;   // `touched` lists every node once
                        for (size_t k = t0; k < touched.size(); k++) {
                            if (touched[k]->is_leaf() && touched[k] != best_node) leaf = touched[k];
                            if (!touched[k]->is_root()) tcache.build(touched[k]);   // (again if it was touched before: its branch changed)
                            if (dev.on) {
                                Node *x = touched[k];
                                if (x->is_root()) {   // (the root itself rewritten: left to the host-side scheme, which flattens again)
                                    dev.on = false; have_spec = false;
                                    continue;
                                }
                                if (!dev.pending_set.count(x)) {
                                    auto it = dev.rec_of.find(x);
                                    if (it != dev.rec_of.end()) { dev.rec_dead[it->second] = 1; dev.pending_retired.push_back(it->second); dev.rec_of.erase(it); }
                                    dev.pending.push_back(x); dev.pending_set.insert(x);
                                }
                                if (flat.has(x)) dev.since_batch.insert(x);
                                continue;
                            }
                            if (touched_set.insert(touched[k]).second) touched[keep++] = touched[k];
                        }
                        touched.resize(keep);
                        if (!dev.on && dev.flat_done && !have_spec) { touched.clear(); touched_set.clear(); }
                        if (leaf) {
                            if (leaf->parent != best_node) added_leaves[leaf->parent] = best_leaves;   // the new internal node starts with best_node's leaves
                            added_leaves[leaf] = 1;
                            for (const Node *a = leaf->parent; a; a = a->parent) added_leaves[a] += 1;
                        }
                    }
                }
                if (!vec.imputed.empty()) {                                     // :767-781
                    fprintf(stderr, "Imputed mutations:\t");
                    for (size_t i = 0; i < vec.imputed.size(); i++) {
                        const char *sep = i + 1 < vec.imputed.size() ? ";" : "";
                        fprintf(stderr, "%i:%c%s", vec.imputed[i].position, nuc_char(vec.imputed[i].mut_nuc), sep);
                        fprintf(stats, "%i:%c%s", vec.imputed[i].position, nuc_char(vec.imputed[i].mut_nuc), sep);
                    }
                    fprintf(stderr, "\n");
                }
            }
            fputc('\n', stats);
            fprintf(stderr, "Completed in %ld msec \n\n", timer.stop());
        }
        fclose(stats);
        if (scores_file) fclose(scores_file);
        if (getenv("USHER_AMD_PROFILE"))
            fprintf(stderr, "[usher-amd profile] tree -> arrays %.3f s (%zu times), batched placement %.3f s (%zu batches), tie lists %.3f s, "
                            "touched-node evaluation %.3f s (%zu node evaluations)\n", prof.build, prof.flats, prof.place, prof.batches, prof.ties,
                    prof.touched, prof.evals);
        prof.loop = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - t_loop0;
        if (getenv("USHER_AMD_PROFILE"))
            fprintf(stderr, "[usher-amd profile] per-sample loop %.3f s in all: vectors of the winning node (pass 2) %.3f s, tree edits %.3f s, bookkeeping of the edits %.3f s\n",
                    prof.loop, prof.vecs, prof.insert, prof.book);
        if (getenv("USHER_AMD_PROFILE") && dev.flat_done)
            fprintf(stderr, "[usher-amd profile] add mode on the device: %zu batches %.3f s, %zu record uploads %.3f s, result fetches %.3f s, %zu records; "
                            "asked again: flattened tree %zu, records %zu, all records on the host %zu\n", dev.n_batches, dev.t_batch, dev.n_flush, dev.t_flush,
                    dev.t_fetch, dev.rec_node.size(), dev.n_replace, dev.n_rescore, dev.n_host_all);
    }
    if (opt.print_scores) return 0;                                             // :800-805
    std::vector<Tree *> trees{&T};
    const auto t_out = std::chrono::steady_clock::now();
    const int rc = write_outputs(opt, trees, missing, low_confidence, (final_text && opt.no_add && !opt.collapse_output_tree) ? final_text.get() : nullptr);
    if (getenv("USHER_AMD_PROFILE")) fprintf(stderr, "[usher-amd profile] output files %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t_out).count());
    return rc;
}

}  // namespace uh
