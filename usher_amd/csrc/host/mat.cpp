// mat.cpp -- host-side MAT model, newick / parsimony.proto / VCF readers and
// writers.  Written from scratch; each function cites the reference behaviour
// it reproduces (src/mutation_annotated_tree.cpp unless noted).
#include "mat.hpp"

#include <zlib.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cerrno>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <mutex>
#include <new>
#include <sstream>
#include <thread>

#include "par.hpp"

namespace uh {

namespace {
struct Lap {   // USHER_AMD_PROFILE=1: where a loader's time goes
    const bool on = getenv("USHER_AMD_PROFILE") != nullptr;
    const char *what;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    explicit Lap(const char *w) : what(w) {}
    void operator()(const char *step) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        fprintf(stderr, "[usher-amd profile] %s: %s %.3f s\n", what, step, std::chrono::duration<double>(n - t).count());
        t = n;
    }
};
}  // namespace

// ------------------------------------------------------------- nucleotides

int8_t nuc_id(char c) {   // :19-74
    switch (c) {
        case 'a': case 'A': return 1;
        case 'c': case 'C': return 2;
        case 'g': case 'G': return 4;
        case 't': case 'T': return 8;
        case 'R': return 5;
        case 'Y': return 10;
        case 'S': return 6;
        case 'W': return 9;
        case 'K': return 12;
        case 'M': return 3;
        case 'B': return 14;
        case 'D': return 13;
        case 'H': return 11;
        default: return 15;   // 'V' included: the reference's case falls through to N (:65-70)
    }
}

char nuc_char(int8_t id) {   // :88-139
    static const char *t = "NACMGRSVTWYHKDBN";
    return (id >= 1 && id <= 14) ? t[id] : 'N';
}

int8_t nuc_index(int8_t id) {   // :142-162
    switch (id) { case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3; default: return -1; }
}

std::string Mutation::str() const {   // hpp:79-85
    if (masked()) return "MASKED";
    return std::string(1, nuc_char(par_nuc)) + std::to_string(position) + std::string(1, nuc_char(mut_nuc));
}

// -------------------------------------------------------------------- node

bool Node::add_mutation(const Mutation &mut) {   // :720-752
    auto it = std::lower_bound(mutations.begin(), mutations.end(), mut,
                               [](const Mutation &a, const Mutation &b) { return a.position < b.position; });
    if (it != mutations.end() && it->position == mut.position) {
        if (it->par_nuc != mut.mut_nuc) {
            it->mut_nuc = mut.mut_nuc;           // update to the new allele
        } else {
            if (it->mut_nuc != mut.par_nuc) return false;   // "consecutive mutations at same position disagree"
            const int32_t p = it->position;       // reversal: drop every mutation at this position
            mutations.erase(std::remove_if(mutations.begin(), mutations.end(),
                                           [p](const Mutation &m) { return m.position == p; }),
                            mutations.end());
        }
    } else {
        mutations.insert(it, mut);
    }
    return true;
}

// -------------------------------------------------------------------- tree

// ---------------------------------------------------------- name index, node blocks

uint64_t NodeIndex::hash(const std::string &s) {   // FNV-1a, then a finalizer (names are short: "node_1234567", "EPI_ISL_...")
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    h ^= h >> 32; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29;
    return h;
}

size_t NodeIndex::base_find(const std::string &id, uint64_t h) const {
    auto it = std::lower_bound(base_.begin(), base_.end(), h, [](const std::pair<uint64_t, Node *> &e, uint64_t v) { return e.first < v; });
    for (; it != base_.end() && it->first == h; ++it)
        if (it->second && it->second->id == id) return (size_t)(it - base_.begin());
    return SIZE_MAX;
}

Node *NodeIndex::find(const std::string &id) const {
    if (!extra_.empty()) { auto it = extra_.find(id); if (it != extra_.end()) return it->second; }
    if (base_.empty()) return nullptr;
    const size_t i = base_find(id, hash(id));
    return i == SIZE_MAX ? nullptr : base_[i].second;
}

bool NodeIndex::insert(const std::string &id, Node *n) {
    if (find(id)) return false;
    extra_.emplace(id, n);
    live_++;
    return true;
}

void NodeIndex::set(const std::string &id, Node *n) {
    auto it = extra_.find(id);
    if (it != extra_.end()) { it->second = n; return; }
    if (!base_.empty()) { const size_t i = base_find(id, hash(id)); if (i != SIZE_MAX) { base_[i].second = n; return; } }
    extra_.emplace(id, n);
    live_++;
}

bool NodeIndex::erase(const std::string &id) {
    auto it = extra_.find(id);
    if (it != extra_.end()) { extra_.erase(it); live_--; return true; }
    if (base_.empty()) return false;
    const size_t i = base_find(id, hash(id));
    if (i == SIZE_MAX) return false;
    base_[i].second = nullptr;
    live_--;
    return true;
}

// Hashes on the host threads, a partition by the top hash bits (histogram, scan, scatter), every partition sorted by its
// own thread.  Equal names are equal hashes, hence neighbours after the sort.
bool NodeIndex::bulk_build(Node *const *nodes, size_t n, std::string *dup) {
    std::vector<std::pair<uint64_t, Node *>> tmp(n);
    parallel_for(n, [&](uint64_t b, uint64_t e, unsigned) { for (uint64_t i = b; i < e; i++) tmp[i] = {hash(nodes[i]->id), nodes[i]}; }, 1u << 14);
    constexpr unsigned BITS = 10, P = 1u << BITS;
    const unsigned T = host_threads();
    std::vector<uint64_t> hist((size_t)T * P, 0);
    parallel_for(n, [&](uint64_t b, uint64_t e, unsigned tid) { uint64_t *h = &hist[(size_t)tid * P]; for (uint64_t i = b; i < e; i++) h[tmp[i].first >> (64 - BITS)]++; }, 1u << 14);
    std::vector<uint64_t> start(P + 1, 0);
    {   // offsets: partition-major, thread-minor (a thread's entries of one partition stay together: deterministic)
        uint64_t run = 0;
        for (unsigned p = 0; p < P; p++) { start[p] = run; for (unsigned t = 0; t < T; t++) { const uint64_t c = hist[(size_t)t * P + p]; hist[(size_t)t * P + p] = run; run += c; } }
        start[P] = run;
    }
    base_.assign(n, {0, nullptr});
    parallel_for(n, [&](uint64_t b, uint64_t e, unsigned tid) { uint64_t *h = &hist[(size_t)tid * P]; for (uint64_t i = b; i < e; i++) base_[h[tmp[i].first >> (64 - BITS)]++] = tmp[i]; }, 1u << 14);
    std::atomic<size_t> bad{SIZE_MAX};
    parallel_for(P, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t p = b; p < e; p++) {
            auto lo = base_.begin() + start[p], hi = base_.begin() + start[p + 1];
            std::sort(lo, hi, [](const std::pair<uint64_t, Node *> &x, const std::pair<uint64_t, Node *> &y) { return x.first != y.first ? x.first < y.first : x.second < y.second; });
            for (auto it = lo; it != hi; ++it)
                for (auto jt = it + 1; jt != hi && jt->first == it->first; ++jt)
                    if (jt->second->id == it->second->id) { size_t cur = bad.load(); const size_t me = (size_t)(it - base_.begin()); while (me < cur && !bad.compare_exchange_weak(cur, me)) {} }
        }
    }, 1);
    live_ = n + extra_.size();
    if (bad.load() != SIZE_MAX) { if (dup) *dup = base_[bad.load()].second->id; return false; }
    return true;
}

Node *Tree::alloc_block(size_t n) {
    Block b;
    b.p = (Node *)::operator new[](std::max<size_t>(n, 1) * sizeof(Node), std::align_val_t(alignof(Node)));
    b.n = n;
    b.dead.assign(n, 1);   // (nothing constructed yet; the caller marks what it constructs)
    blocks.push_back(std::move(b));
    return blocks.back().p;
}

void Tree::free_node(Node *n) {
    if (!n->in_block) { delete n; return; }
    for (Block &b : blocks)
        if (n >= b.p && n < b.p + b.n) { b.dead[(size_t)(n - b.p)] = 1; n->~Node(); return; }
}

static void free_all_nodes(Tree &T) {
    T.all_nodes.for_each([](Node *n) { if (!n->in_block) delete n; });
    for (Tree::Block &b : T.blocks) {
        parallel_for(b.n, [&](uint64_t lo, uint64_t hi, unsigned) { for (uint64_t i = lo; i < hi; i++) if (!b.dead[i]) b.p[i].~Node(); }, 1u << 14);
        ::operator delete[](b.p, std::align_val_t(alignof(Node)));
    }
    T.blocks.clear();
    T.all_nodes.clear();
}

Tree::~Tree() { free_all_nodes(*this); }

uint32_t Tree::chrom_id(const std::string &c) {
    for (uint32_t i = 0; i < chroms.size(); i++) if (chroms[i] == c) return i;
    chroms.push_back(c);
    return (uint32_t)chroms.size() - 1;
}

Node *Tree::get_node(const std::string &id) const {
    return all_nodes.find(id);
}

Node *Tree::create_node(const std::string &id, Node *parent, float branch_length) {   // :881-910
    if (parent) {
        if (all_nodes.find(id)) return nullptr;   // "already in the tree"
        Node *n = new Node();
        n->id = id;
        n->parent = parent;
        n->branch_length = branch_length;
        n->level = parent->level + 1;
        const size_t na = num_annotations();
        if (na) n->clade_annotations.assign(na, "");
        parent->children.push_back(n);
        all_nodes.set(id, n);
        return n;
    }
    Node *n = new Node();
    n->id = id;
    n->parent = parent;
    n->branch_length = branch_length;
    n->level = parent ? parent->level + 1 : 1;
    if (!parent) {
        free_all_nodes(*this);
        root = n;
    } else {
        n->clade_annotations.assign(num_annotations(), "");
        parent->children.push_back(n);
    }
    all_nodes.set(id, n);
    return n;
}

std::vector<Node *> Tree::bfs() const {
    std::vector<Node *> out;
    if (!root) return out;
    out.reserve(all_nodes.size());
    out.push_back(root);
    for (size_t h = 0; h < out.size(); h++)
        for (Node *c : out[h]->children) out.push_back(c);
    return out;
}

std::vector<Node *> Tree::dfs(Node *from) const {
    std::vector<Node *> out;
    Node *start = from ? from : root;
    if (!start) return out;
    std::vector<Node *> st{start};
    while (!st.empty()) {
        Node *n = st.back();
        st.pop_back();
        out.push_back(n);
        for (size_t i = n->children.size(); i-- > 0;) st.push_back(n->children[i]);
    }
    return out;
}

std::vector<Node *> Tree::rsearch(Node *n, bool include_self) const {
    std::vector<Node *> out;
    if (!n) return out;
    if (include_self) out.push_back(n);
    for (Node *p = n->parent; p; p = p->parent) out.push_back(p);
    return out;
}

std::string Tree::clade_assignment(Node *n, size_t clade, bool include_self) const {
    for (Node *a : rsearch(n, include_self))
        if (a->clade_annotations.size() > clade && !a->clade_annotations[clade].empty()) return a->clade_annotations[clade];
    return "UNDEFINED";
}

size_t Tree::parsimony_score() const {
    size_t s = 0;
    all_nodes.for_each([&](Node *n) { s += n->mutations.size(); });
    return s;
}

void Tree::fix_levels(Node *from) {
    std::vector<Node *> st{from};
    while (!st.empty()) {
        Node *n = st.back();
        st.pop_back();
        n->level = n->parent ? n->parent->level + 1 : 1;
        for (Node *c : n->children) st.push_back(c);
    }
}

// move_node() for the case usher's placement uses: the destination was just
// created and has one mutation-free child, so no "same mutations" merge can
// trigger (:1150-1158): unlink from the old parent, append below dst.
void Tree::reattach(Node *src, Node *dst) {
    Node *old = src->parent;
    old->children.erase(std::find(old->children.begin(), old->children.end(), src));
    src->parent = dst;
    src->branch_length = -1.0f;
    dst->children.push_back(src);
    fix_levels(src);
}

void Tree::remove_leaf(Node *n) {
    Node *p = n->parent;
    if (p) p->children.erase(std::find(p->children.begin(), p->children.end(), n));
    all_nodes.erase(n->id);
    free_node(n);
}

std::vector<Node *> Tree::leaves(Node *from) const {   // get_leaves, :818-840
    std::vector<Node *> out, q;
    Node *start = from ? from : root;
    if (!start) return out;
    q.push_back(start);
    for (size_t h = 0; h < q.size(); h++) {
        if (q[h]->children.empty()) out.push_back(q[h]);
        for (Node *c : q[h]->children) q.push_back(c);
    }
    return out;
}

bool Tree::is_ancestor(const Node *anc, const Node *n) const {   // :920-929 (proper ancestor)
    for (const Node *p = n->parent; p; p = p->parent) if (p == anc) return true;
    return false;
}

size_t Tree::num_leaves(Node *n) const { return leaves(n).size(); }

// remove_node_helper, :960-1049.  Deletes `source` with its descendants; a parent left without children goes too,
// and with move_level a parent left with ONE child is spliced out (the child inherits its mutations / annotations).
void Tree::remove_node(Node *source, bool move_level) {
    Node *par = source->parent;
    if (par) {
        par->children.erase(std::find(par->children.begin(), par->children.end(), source));
        if (par->children.empty()) {
            if (par != root) remove_node(par, move_level);      // (the reference exits on an emptied tree)
        } else if (move_level && par->children.size() == 1) {
            Node *child = par->children[0];
            if (par->parent) {
                for (size_t k = 0; k < par->clade_annotations.size() && k < child->clade_annotations.size(); k++)
                    if (child->clade_annotations[k].empty()) child->clade_annotations[k] = par->clade_annotations[k];
                child->parent = par->parent;
                child->branch_length += par->branch_length;
                const std::vector<Mutation> own = child->mutations;
                child->mutations.clear();
                for (const Mutation &m : par->mutations) child->add_mutation(m);
                for (const Mutation &m : own) child->add_mutation(m);
                par->parent->children.push_back(child);
                par->parent->children.erase(std::find(par->parent->children.begin(), par->parent->children.end(), par));
                fix_levels(child);
                all_nodes.erase(par->id);
                free_node(par);
            }
        }
    }
    std::vector<Node *> q{source};
    for (size_t h = 0; h < q.size(); h++) for (Node *c : q[h]->children) q.push_back(c);
    for (Node *n : q) { all_nodes.erase(n->id); free_node(n); }
}

static bool same_mutations(const std::vector<Mutation> &a, const std::vector<Mutation> &b) {   // Mutation::operator==, hpp:56-62
    if (a.size() != b.size()) return false;
    for (size_t i = 0; i < a.size(); i++)
        if (a[i].position != b[i].position || a[i].is_missing != b[i].is_missing || a[i].chrom != b[i].chrom ||
            a[i].par_nuc != b[i].par_nuc || a[i].mut_nuc != b[i].mut_nuc) return false;
    return true;
}

// move_node, :1135-1223.  (Mutation lists are kept sorted by add_mutation, so find_child_with_muts' sorting is a no-op.)
void Tree::move_node(Node *source, Node *destination, bool move_level) {
    Node *curr_parent = source->parent;
    if (curr_parent == destination) return;   // (the reference exits with an error)
    auto link = [](Node *parent, Node *child) { child->parent = parent; child->branch_length = -1.0f; parent->children.push_back(child); };
    auto unlink = [&](Node *parent, Node *child) {   // remove_child, :1118-1131
        parent->children.erase(std::find(parent->children.begin(), parent->children.end(), child));
        if (parent->children.empty()) remove_node(parent, move_level);
    };
    Node *existing = nullptr;                      // a child of destination with the same (non-empty) mutations
    for (Node *c : destination->children) if (same_mutations(c->mutations, source->mutations)) { existing = c; break; }
    if (existing == curr_parent || source->mutations.empty()) existing = nullptr;
    std::vector<Node *> relevel;
    if (!existing) {
        link(destination, source);
        unlink(curr_parent, source);
        relevel.push_back(source);
    } else if (existing->is_leaf()) {
        if (source->is_leaf()) {                   // two leaves: a new internal node carries the shared mutations
            Node *mid = create_node(new_internal_node_id(), destination, -1.0f);
            for (const Mutation &m : source->mutations) mid->add_mutation(m);
            source->mutations.clear();
            existing->mutations.clear();
            link(mid, source);
            link(mid, existing);
            unlink(destination, existing);
            unlink(curr_parent, source);
            relevel.push_back(mid);
        } else {                                   // the existing leaf moves into source
            existing->mutations.clear();
            link(source, existing);
            link(destination, source);
            unlink(destination, existing);
            unlink(curr_parent, source);
            relevel.push_back(source);
        }
    } else if (source->is_leaf()) {                // source moves into the existing internal node
        source->mutations.clear();
        link(existing, source);
        unlink(curr_parent, source);
        relevel.push_back(source);
    } else {                                       // both internal: source's children move into the existing node (recursively)
        const std::vector<Node *> kids = source->children;
        for (Node *k : kids) move_node(k, existing, move_level);
    }
    for (Node *n : relevel) fix_levels(n);
}

static void collapse_r(Tree *T, Node *node) {   // collapse_tree_r, :1384-1420
    if (node->children.empty()) return;
    const std::vector<Node *> kids = node->children;
    for (Node *c : kids) collapse_r(T, c);
    Node *parent = node->parent;
    if (!parent) return;
    if (node->mutations.empty()) {
        const std::vector<Node *> now = node->children;
        for (Node *c : now) T->move_node(c, parent, false);
    } else if (node->children.size() == 1) {
        Node *child = node->children.front();
        for (const Mutation &m : child->mutations) node->add_mutation(m);
        child->mutations = node->mutations;
        T->move_node(child, parent, false);
    }
}

void Tree::collapse_tree() { if (root) collapse_r(this, root); }

void Tree::rotate_for_display() {   // :1426-1453
    const std::vector<Node *> order = dfs();
    std::unordered_map<Node *, int> nd;
    for (size_t i = order.size(); i-- > 0;) {
        int d = 1;
        for (Node *c : order[i]->children) d += nd[c];
        nd[order[i]] = d;
    }
    // (tbb::parallel_sort is not stable; children with equal counts keep their order here)
    for (Node *n : order) std::stable_sort(n->children.begin(), n->children.end(), [&](Node *a, Node *b) { return nd[a] > nd[b]; });
}

bool get_subtree(const Tree &src, const std::vector<std::string> &samples, Tree &dst, std::string &err) {   // :1575-1681
    std::unordered_set<const Node *> keep;
    std::vector<std::unordered_set<const Node *>> anc(samples.size());
    for (size_t k = 0; k < samples.size(); k++) {
        Node *n = src.get_node(samples[k]);
        if (!n) { err = "get_subtree: sample " + samples[k] + " is not in the tree"; return false; }
        keep.insert(n);
        for (Node *a : src.rsearch(n, true)) anc[k].insert(a);
    }
    for (size_t i = 0; i < samples.size(); i++)
        for (size_t j = i + 1; j < samples.size(); j++)
            for (Node *a : src.rsearch(src.get_node(samples[i]), true))
                if (anc[j].count(a)) { keep.insert(a); break; }
    const size_t n_ann = src.num_annotations();
    std::vector<Node *> last;   // stack of subtree nodes (source nodes) on the current path
    for (Node *n : src.dfs()) {
        if (!keep.count(n)) continue;
        while (!last.empty() && !src.is_ancestor(last.back(), n)) last.pop_back();
        Node *sp = last.empty() ? nullptr : last.back();
        Node *nn = dst.create_node(n->id, sp ? dst.get_node(sp->id) : nullptr, -1.0f);
        if (!nn) { err = "get_subtree: duplicate node " + n->id; return false; }
        nn->clade_annotations.assign(n_ann, "");
        std::vector<Node *> path = src.rsearch(n, true);           // n .. root
        std::reverse(path.begin(), path.end());                     // root .. n
        if (sp) path.erase(path.begin(), std::find(path.begin(), path.end(), sp) + 1);
        for (Node *c : path) {
            // (the root of the subtree takes only its own annotations, :1640-1644; inner nodes inherit along the collapsed path)
            if (sp || c == n)
                for (size_t k = 0; k < n_ann && k < c->clade_annotations.size(); k++)
                    if (!c->clade_annotations[k].empty()) nn->clade_annotations[k] = c->clade_annotations[k];
            for (const Mutation &m : c->mutations) nn->add_mutation(m);
        }
        last.push_back(n);
    }
    dst.curr_internal_node = src.curr_internal_node;
    dst.chroms = src.chroms;
    return true;
}

void Tree::condense_leaves() {   // :1287-1332
    if (!condensed_nodes.empty()) uncondense_leaves();
    std::vector<std::string> leaf_ids;
    for (Node *n : bfs()) if (n->is_leaf()) leaf_ids.push_back(n->id);
    for (const std::string &lid : leaf_ids) {
        Node *l1 = get_node(lid);
        if (!l1 || !l1->mutations.empty() || !l1->parent) continue;
        std::vector<Node *> poly;
        for (Node *l2 : l1->parent->children)
            if (l2->is_leaf() && l2->mutations.empty()) poly.push_back(l2);
        if (poly.size() > 1) {
            std::string name = "node_" + std::to_string(1 + condensed_nodes.size()) + "_condensed_" +
                               std::to_string(poly.size()) + "_leaves";
            create_node(name, l1->parent, l1->branch_length);
            std::vector<std::string> ids;
            for (Node *p : poly) ids.push_back(p->id);
            for (Node *p : poly) remove_leaf(p);
            add_condensed(name, ids);
        }
    }
}

void Tree::uncondense_leaves() {   // :1334-1382
    for (const std::string &cname : condensed_order) {
        auto cit = condensed_nodes.find(cname);
        if (cit == condensed_nodes.end()) continue;
        auto &cn = *cit;
        Node *n = get_node(cn.first);
        if (!n) continue;
        Node *par = n->parent ? n->parent : n;
        const size_t k = cn.second.size();
        if (k > 1 && !n->mutations.empty()) {
            all_nodes.erase(n->id);
            n->id = new_internal_node_id();
            all_nodes.set(n->id, n);
            for (size_t s = 0; s < k; s++) {
                Node *c = new Node();
                c->id = cn.second[s]; c->parent = n; c->branch_length = -1.0f; c->level = n->level + 1;
                c->clade_annotations.assign(num_annotations(), "");
                all_nodes.set(c->id, c);
                n->children.push_back(c);
            }
        } else if (k > 1) {
            all_nodes.erase(n->id);
            n->id = cn.second[0];
            all_nodes.set(n->id, n);
            for (size_t s = 1; s < k; s++) {
                Node *c = new Node();
                c->id = cn.second[s]; c->parent = par; c->branch_length = n->branch_length; c->level = par->level + 1;
                c->clade_annotations.assign(num_annotations(), "");
                all_nodes.set(c->id, c);
                par->children.push_back(c);
            }
        } else if (k == 1) {
            all_nodes.erase(n->id);
            n->id = cn.second[0];
            all_nodes.set(n->id, n);
        }
    }
    condensed_nodes.clear();
    condensed_order.clear();
    condensed_leaves.clear();
}

bool copy_tree(const Tree &src, Tree &dst, std::string &err) {   // get_tree_copy, :1493-1549
    if (!src.root) { err = "empty tree"; return false; }
    if (!tree_from_newick(newick(src, src.root, true, true), dst, err)) return false;
    const std::vector<Node *> d1 = src.dfs(), d2 = dst.dfs();
    if (d1.size() != d2.size()) { err = "tree copy: node count changed in the newick round trip"; return false; }
    dst.chroms = src.chroms;
    for (size_t k = 0; k < d1.size(); k++) {
        d2[k]->clade_annotations = d1[k]->clade_annotations;
        for (const Mutation &m : d1[k]->mutations) d2[k]->add_mutation(m);
    }
    for (const std::string &cname : src.condensed_order) {
        const auto &ids = src.condensed_nodes.at(cname);
        dst.add_condensed(cname, ids);
        for (const std::string &l : ids) dst.condensed_leaves.insert(l);
    }
    return true;
}

// ------------------------------------------------------------------ newick

static void split(const std::string &s, char delim, std::vector<std::string> &out) {   // string_split, :383-398
    size_t start = 0, end;
    while ((end = s.find(delim, start)) != std::string::npos) {
        out.push_back(s.substr(start, end - start));
        start = end + 1;
    }
    if (start < s.size()) out.push_back(s.substr(start));
}

// A branch length as std::stof reads it (leading number, anything behind it ignored) -- without its exceptions: the reference
// terminates on ":-" or ":e" (mutation_annotated_tree.cpp:447-456 calls std::stof unguarded); here the loaders report an error.
static bool parse_len(const char *text, float &out) {
    errno = 0;
    char *end = nullptr;
    const float v = strtof(text, &end);
    if (end == text || errno == ERANGE) return false;
    out = v;
    return true;
}

bool tree_from_newick(const std::string &nwk, Tree &T, std::string &err) {   // :415-508
    std::vector<std::string> parts;
    split(nwk, ',', parts);
    struct Item { std::string leaf; size_t open = 0, close = 0; };
    std::vector<Item> items;
    std::vector<std::vector<float>> blen(128);
    size_t level = 0;
    bool bad_len = false;
    auto to_len = [&](const std::string &b) { float v = -1.0f; if (!b.empty() && !parse_len(b.c_str(), v)) bad_len = true; return v; };
    for (const std::string &s : parts) {
        Item it;
        bool stop = false, branch_start = false;
        std::string branch;
        for (char c : s) {
            if (c == ':') { stop = true; branch.clear(); branch_start = true; }
            else if (c == '(') { it.open++; level++; if (blen.size() <= level) blen.resize(level * 2); }
            else if (c == ')') {
                stop = true; it.close++;
                blen[level].push_back(to_len(branch));
                if (level == 0) { err = "incorrect Newick format"; return false; }
                level--; branch_start = false;
            } else if (!stop) { it.leaf += c; branch_start = false; }
            else if (branch_start && (isdigit((unsigned char)c) || c == '.' || c == 'e' || c == 'E' || c == '-' || c == '+')) branch += c;
        }
        blen[level].push_back(to_len(branch));
        items.push_back(std::move(it));
    }
    if (level != 0 || bad_len) { err = "incorrect Newick format"; return false; }
    T.all_nodes.reserve(items.size() * 2 + 16);
    std::vector<size_t> head(blen.size(), 0);
    std::vector<Node *> stack;
    for (Item &it : items) {
        for (size_t j = 0; j < it.open; j++) {
            Node *n = T.create_node(T.new_internal_node_id(), stack.empty() ? nullptr : stack.back(), blen[level][head[level]++]);
            if (!n) { err = "duplicate node identifier in Newick"; return false; }
            level++;
            stack.push_back(n);
        }
        if (stack.empty()) { err = "incorrect Newick format"; return false; }
        if (!T.create_node(it.leaf, stack.back(), blen[level][head[level]++])) { err = "Error: " + it.leaf + " already in the tree!"; return false; }
        for (size_t j = 0; j < it.close; j++) { stack.pop_back(); level--; }
    }
    return true;
}

// The same tree as tree_from_newick for large inputs (a 10M-node parsimony.proto: 6.2 s -> 1 s of load time).  The
// per-item state machine is the one above, run per item on the host threads; what crosses items is the paren level
// (a prefix sum) and the node stack (one cheap sequential pass over the items' open / close counts).  A value pushed
// into blen[level] above belongs to the node whose text just ended -- the item's leaf for its first push, the
// internal node closed by the previous ')' for the others -- and nodes of one level end in the order they begin, so
// handing each popped node "its" push is the same assignment as the per-level queues.
bool tree_from_newick_bulk(const char *nwk, size_t len, Tree &T, std::string &err, std::vector<Node *> *order_out) {
    // item boundaries: commas (string_split drops a trailing empty piece, keeps empty pieces in the middle)
    Lap lap("newick");
    const unsigned TH = host_threads();
    std::vector<std::vector<size_t>> commas(TH);
    parallel_for(len, [&](uint64_t b, uint64_t e, unsigned tid) {
        auto &v = commas[tid];
        for (const char *p = nwk + b, *pe = nwk + e; (p = (const char *)memchr(p, ',', (size_t)(pe - p))) != nullptr; p++) v.push_back((size_t)(p - nwk));
    }, 1u << 16);
    std::vector<size_t> start{0};   // item i = [start[i], start[i + 1] - 1)
    for (auto &v : commas) for (size_t c : v) start.push_back(c + 1);
    if (start.back() < len) start.push_back(len + 1);   // (a last piece without a comma behind it)
    const size_t n_items = start.size() - 1;
    if (n_items == 0) return true;
    lap("item boundaries");
    struct Item { uint32_t open, close, leaf_b, leaf_e; };
    std::vector<Item> items(n_items);
    std::vector<uint64_t> push_off(n_items + 1, 0), node_off(n_items + 1, 0), open_off(n_items + 1, 0);
    std::vector<int64_t> level_at(n_items + 1, 0);
    // pass 1: counts, leaf-name extent (the characters before the first ':' or ')' that are not parens)
    parallel_for(n_items, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; i++) {
            Item it{0, 0, 0, 0};
            bool stop = false;
            const size_t sb = start[i], se = start[i + 1] - 1;
            uint32_t lb = (uint32_t)sb, le = (uint32_t)sb;
            bool any = false;
            for (size_t k = sb; k < se; k++) {
                const char c = nwk[k];
                if (c == ':') stop = true;
                else if (c == '(') it.open++;
                else if (c == ')') { stop = true; it.close++; }
                else if (!stop) { if (!any) { lb = (uint32_t)k; any = true; } le = (uint32_t)k + 1; }
            }
            it.leaf_b = lb; it.leaf_e = any ? le : lb;
            items[i] = it;
            push_off[i] = it.close + 1; node_off[i] = it.open + 1; open_off[i] = it.open;
        }
    }, 4096);
    // a leaf name interrupted by parens ("a(b") would be the concatenation of its pieces: left to the general routine
    std::atomic<bool> odd{false};
    parallel_for(n_items, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t i = b; i < e; i++)
            for (uint32_t k = items[i].leaf_b; k < items[i].leaf_e; k++) if (nwk[k] == '(') { odd = true; break; }
    }, 4096);
    if (odd) return tree_from_newick(std::string(nwk, len), T, err);
    lap("item counts");
    const uint64_t n_push = exclusive_scan(push_off.data(), n_items); push_off[n_items] = n_push;
    const uint64_t n_nodes = exclusive_scan(node_off.data(), n_items); node_off[n_items] = n_nodes;
    const uint64_t n_open = exclusive_scan(open_off.data(), n_items); open_off[n_items] = n_open;
    if (n_nodes >= (1ull << 32)) { err = "Newick tree too large"; return false; }
    // pass 2: the branch lengths an item pushes, in order (the state machine of tree_from_newick, quirks included:
    // `branch` survives a ')' that is not followed by a new ':')
    std::vector<float> pushes(n_push);
    std::atomic<int> len_err{0};   // 1: a branch length that is no number; 2: one longer than the buffer (left to the general routine)
    parallel_for(n_items, [&](uint64_t b, uint64_t e, unsigned) {
        char buf[64];
        for (uint64_t i = b; i < e; i++) {
            float *out = &pushes[push_off[i]];
            size_t bl = 0;
            bool has = false, branch_start = false;
            auto val = [&]() -> float {
                if (!has || bl == 0) return -1.0f;
                buf[bl] = 0;
                float v = -1.0f;
                if (!parse_len(buf, v)) len_err = 1;
                return v;
            };
            for (size_t k = start[i], se = start[i + 1] - 1; k < se; k++) {
                const char c = nwk[k];
                if (c == ':') { bl = 0; has = true; branch_start = true; }
                else if (c == '(') {}
                else if (c == ')') { *out++ = val(); branch_start = false; }
                else if (branch_start && (isdigit((unsigned char)c) || c == '.' || c == 'e' || c == 'E' || c == '-' || c == '+')) { if (bl < sizeof buf - 1) buf[bl++] = c; else if (!len_err) len_err = 2; }
                else if (has) { /* a character that does not belong to a number: tree_from_newick skips it */ }
            }
            *out++ = val();
        }
    }, 4096);
    if (len_err == 1) { err = "incorrect Newick format"; return false; }
    if (len_err == 2) return tree_from_newick(std::string(nwk, len), T, err);
    lap("branch lengths");
    // sequential: the node stack -> parent and branch length of every node, children counts
    std::vector<uint32_t> parent(n_nodes);
    std::vector<float> blen(n_nodes, -1.0f);
    std::vector<uint32_t> level(n_nodes), n_kids(n_nodes, 0);
    {
        std::vector<uint32_t> stack;
        for (size_t i = 0; i < n_items; i++) {
            const Item &it = items[i];
            uint32_t id = (uint32_t)node_off[i];
            for (uint32_t j = 0; j < it.open; j++, id++) {
                parent[id] = stack.empty() ? UINT32_MAX : stack.back();
                if (stack.empty() && id != 0) { err = "incorrect Newick format"; return false; }   // (a second root)
                level[id] = (uint32_t)stack.size() + 1;
                if (!stack.empty()) n_kids[stack.back()]++;
                stack.push_back(id);
            }
            if (stack.empty()) { err = "incorrect Newick format"; return false; }
            parent[id] = stack.back(); level[id] = (uint32_t)stack.size() + 1; n_kids[stack.back()]++;
            const float *pv = &pushes[push_off[i]];
            blen[id] = pv[0];
            if (it.close > stack.size()) { err = "incorrect Newick format"; return false; }
            for (uint32_t j = 0; j < it.close; j++) { blen[stack.back()] = pv[j + 1]; stack.pop_back(); }
        }
        if (!stack.empty()) { err = "incorrect Newick format"; return false; }
    }
    lap("node stack (sequential)");
    // construct the nodes in one block, on the host threads (first touch by the thread that fills them)
    if (T.root) { err = "tree_from_newick_bulk needs an empty tree"; return false; }
    Node *blk = T.alloc_block(n_nodes);
    Tree::Block &B = T.blocks.back();
    const size_t first_internal = T.curr_internal_node;
    parallel_for(n_items, [&](uint64_t b, uint64_t e, unsigned) {
        char name[40];
        for (uint64_t i = b; i < e; i++) {
            uint32_t id = (uint32_t)node_off[i];
            for (uint32_t j = 0; j < items[i].open; j++, id++) {
                Node *n = new (blk + id) Node();
                snprintf(name, sizeof name, "node_%zu", first_internal + (size_t)open_off[i] + j + 1);
                n->id = name;
            }
            Node *n = new (blk + id) Node();
            n->id.assign(nwk + items[i].leaf_b, items[i].leaf_e - items[i].leaf_b);
        }
    }, 4096);
    parallel_for(n_nodes, [&](uint64_t b, uint64_t e, unsigned) {
        for (uint64_t k = b; k < e; k++) {
            Node *n = blk + k;
            n->in_block = true;
            n->parent = parent[k] == UINT32_MAX ? nullptr : blk + parent[k];
            n->branch_length = blen[k];
            n->level = level[k];
            if (n_kids[k]) n->children.reserve(n_kids[k]);
            B.dead[k] = 0;
        }
    }, 1u << 14);
    lap("construct nodes");
    for (uint64_t k = 1; k < n_nodes; k++) blk[parent[k]].children.push_back(blk + k);   // creation order = stored child order
    lap("child lists (sequential)");
    T.curr_internal_node = first_internal + n_open;
    T.root = blk;
    std::vector<Node *> all(n_nodes);
    parallel_for(n_nodes, [&](uint64_t b, uint64_t e, unsigned) { for (uint64_t k = b; k < e; k++) all[k] = blk + k; }, 1u << 16);
    std::string dup;
    if (!T.all_nodes.bulk_build(all.data(), n_nodes, &dup)) {
        err = dup.rfind("node_", 0) == 0 ? "duplicate node identifier in Newick" : "Error: " + dup + " already in the tree!";
        return false;
    }
    lap("name index");
    if (order_out) order_out->swap(all);
    return true;
}

static void put_len(std::string &out, float v) {   // operator<<(float): %g
    char buf[32];
    snprintf(buf, sizeof buf, "%g", v);
    out += buf;
}

// The same text for large trees, written on the host threads: text(n) = name[:len] for a leaf, "(" text(c1) "," ... ")" [name] [:len]
// for an internal node (what the loop below emits, paren by paren).  Lengths bottom-up over the depth-first order (a subtree is a
// contiguous range of it), offsets top-down, then every node writes its own pieces.  Condensed leaves are left to the loop.
static std::string newick_threads(const Tree &T, Node *start, bool internal_ids, bool branch_len) {
    const std::vector<Node *> order = T.dfs(start);
    const size_t n = order.size();
    std::vector<uint32_t> sub(n), lab(n);      // subtree size (nodes), length of the node's own label text (name + ":len")
    std::vector<uint64_t> len(n), off(n);
    auto len_text = [](float v, char *buf) { return snprintf(buf, 32, "%g", v); };
    parallel_for(n, [&](uint64_t b, uint64_t e, unsigned) {
        char buf[32];
        for (uint64_t i = b; i < e; i++) {
            const Node *x = order[i];
            const bool leaf = x->is_leaf();
            uint32_t l = (leaf || internal_ids) ? (uint32_t)x->id.size() : 0u;
            if (branch_len) l += 1u + (uint32_t)len_text((float)x->mutations.size(), buf);
            lab[i] = l;
        }
    }, 1u << 14);
    for (size_t i = n; i-- > 0;) {             // children of i: i + 1, then one subtree after the other
        const Node *x = order[i];
        uint32_t s = 1;
        uint64_t l = lab[i];
        const size_t k = x->children.size();
        if (k) {
            l += 2 + (k - 1);                  // parens and commas
            size_t c = i + 1;
            for (size_t j = 0; j < k; j++) { s += sub[c]; l += len[c]; c += sub[c]; }
        }
        sub[i] = s; len[i] = l;
    }
    off[0] = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t k = order[i]->children.size();
        uint64_t o = off[i] + 1;
        size_t c = i + 1;
        for (size_t j = 0; j < k; j++) { off[c] = o; o += len[c] + 1; c += sub[c]; }
    }
    std::string out;
    out.resize(len[0] + 1);
    char *dst = &out[0];
    parallel_for(n, [&](uint64_t b, uint64_t e, unsigned) {
        char buf[32];
        for (uint64_t i = b; i < e; i++) {
            const Node *x = order[i];
            const bool leaf = x->is_leaf();
            char *p = dst + off[i];
            if (!leaf) {
                *p = '(';
                size_t c = i + 1;
                for (size_t j = 0; j + 1 < x->children.size(); j++) { c += sub[c]; dst[off[c] - 1] = ','; }
                p = dst + off[i] + len[i] - lab[i];
                p[-1] = ')';
            }
            if (leaf || internal_ids) { memcpy(p, x->id.data(), x->id.size()); p += x->id.size(); }
            if (branch_len) { *p++ = ':'; const int k = len_text((float)x->mutations.size(), buf); memcpy(p, buf, (size_t)k); }
        }
    }, 1u << 14);
    out[len[0]] = ';';
    return out;
}

std::string newick(const Tree &T, Node *from, bool internal_ids, bool branch_len, bool uncondense) {   // :215-346
    std::string out;
    Node *start = from ? from : T.root;
    if (!start) return ";";
    if ((!uncondense || T.condensed_nodes.empty()) && (T.all_nodes.size() >= (1u << 16) || getenv("USHER_AMD_GRAIN")) && !getenv("USHER_AMD_NEWICK_LOOP"))
        return newick_threads(T, start, internal_ids, branch_len);
    std::unordered_map<std::string, const std::vector<std::string> *> cmap;
    if (uncondense) for (auto &cn : T.condensed_nodes) cmap[cn.first] = &cn.second;
    const size_t level_offset = start->level - 1;
    size_t curr_level = 0;
    bool prev_open = true;
    std::vector<std::pair<std::string, float>> stack;
    auto leaf_text = [&](Node *n, bool comma) {
        if (comma) out += ',';
        auto it = cmap.find(n->id);
        if (uncondense && it != cmap.end()) {
            for (size_t i = 0; i < it->second->size(); i++) { if (i) out += ','; out += (*it->second)[i]; }
        } else out += n->id;
        if (branch_len) { out += ':'; put_len(out, (float)n->mutations.size()); }
    };
    auto close_one = [&]() {
        out += ')';
        if (internal_ids) out += stack.back().first;
        if (branch_len && stack.back().second >= 0) { out += ':'; put_len(out, stack.back().second); }
        stack.pop_back();
    };
    for (Node *n : T.dfs(start)) {
        const size_t level = n->level - level_offset;
        const float bl = (float)n->mutations.size();   // the ":230 band-aid": branch length = #mutations
        if (curr_level < level) {
            if (!prev_open) out += ',';
            size_t l = level - 1;
            if (curr_level > 1) l = level - curr_level;
            for (size_t i = 0; i < l; i++) { out += '('; prev_open = true; }
            if (n->is_leaf()) { leaf_text(n, false); prev_open = false; }
            else stack.emplace_back(n->id, bl);
        } else if (curr_level > level) {
            prev_open = false;
            for (size_t i = level; i < curr_level; i++) close_one();
            if (n->is_leaf()) leaf_text(n, true);
            else stack.emplace_back(n->id, bl);
        } else {
            prev_open = false;
            if (n->is_leaf()) leaf_text(n, true);
            else stack.emplace_back(n->id, bl);
        }
        curr_level = level;
    }
    while (!stack.empty()) close_one();
    out += ';';
    return out;
}

// ---------------------------------------------------------- parsimony.proto

namespace {

// ISIZE of a gzip file: the length of the (last member's) data modulo 2^32 -- a size hint, 0 when unreadable
size_t gz_isize(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return 0;
    uint8_t t[4] = {0, 0, 0, 0};
    size_t v = 0;
    if (fseek(f, -4, SEEK_END) == 0 && fread(t, 1, 4, f) == 4) v = (size_t)t[0] | ((size_t)t[1] << 8) | ((size_t)t[2] << 16) | ((size_t)t[3] << 24);
    fclose(f);
    return v;
}

// A gzip file inflated on a thread of its own into a buffer of the size its trailer names, while the caller reads what is there
// already (load_mat: the field scan and the newick parse run under the inflate of the rest of a .pb.gz -- the public MAT ships
// compressed, mutation_annotated_tree.cpp:522-547, and one zlib stream inflates on one thread at ~0.3 GB/s).  state: 0 running,
// 1 complete, -1 read error, -2 the data does not fit the hint (several members, or more than 4 GB): the caller starts over
// with read_file.
struct GzProgressive {
    std::string buf;
    std::atomic<size_t> avail{0};
    std::atomic<int> state{0};
    std::thread th;
    ~GzProgressive() { if (th.joinable()) th.join(); }
    bool start(const std::string &path) {
        const size_t hint = gz_isize(path);
        struct stat st;
        if (hint < (1u << 20) || stat(path.c_str(), &st) != 0 || (size_t)st.st_size > hint) return false;   // (small, or not plausible: the simple way)
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) return false;
        gzbuffer(f, 1u << 20);
        buf.resize(hint);
        th = std::thread([this, f]() {
            size_t used = 0;
            int n = 0;
            while (used < buf.size()) {
                n = gzread(f, &buf[used], (unsigned)std::min<size_t>(buf.size() - used, 1u << 22));
                if (n <= 0) break;
                used += (size_t)n;
                avail.store(used, std::memory_order_release);
            }
            int fin = -1;
            if (n < 0) fin = -1;
            else if (used < buf.size()) fin = -2;                       // shorter than the trailer says
            else { char c; fin = gzread(f, &c, 1) == 0 ? 1 : -2; }      // (exactly full: must be the end)
            gzclose(f);
            state.store(fin, std::memory_order_release);
        });
        return true;
    }
    // true when bytes [0, n) are there; false when the stream ended (or failed) before
    bool wait_for(size_t n) {
        for (unsigned spin = 0;; spin++) {
            if (avail.load(std::memory_order_acquire) >= n) return true;
            if (state.load(std::memory_order_acquire) != 0) return avail.load(std::memory_order_acquire) >= n;
            if (spin < 64) std::this_thread::yield(); else usleep(200);
        }
    }
    int finish() { if (th.joinable()) th.join(); return state.load(); }
};

bool read_file(const std::string &path, std::string &buf, std::string &err) {
    if (path.find(".gz") != std::string::npos) {   // :530 / :2086
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) { err = "Could not open " + path; return false; }
        gzbuffer(f, 1u << 20);
        // (inflated straight into the result, which starts at the size the gzip trailer names -- a multiple of 2^32 short for huge
        // files and for several members: it grows when that runs out)
        size_t used = 0;
        buf.resize(std::max<size_t>(gz_isize(path), 1u << 22));
        int n;
        for (;;) {
            if (used == buf.size()) buf.resize(buf.size() + buf.size() / 2);
            n = gzread(f, &buf[used], (unsigned)std::min<size_t>(buf.size() - used, 1u << 30));
            if (n <= 0) break;
            used += (size_t)n;
        }
        gzclose(f);
        buf.resize(used);
        return n == 0;
    }
    // one read into a buffer of the file's size (a stringstream copies a 1 GB VCF twice)
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { err = "Could not open " + path; return false; }
    struct stat st;
    size_t size = (fstat(fileno(f), &st) == 0 && st.st_size > 0) ? (size_t)st.st_size : 0;
    if (size) {
        buf.resize(size);
        // the pages of a large file are read by several threads at once (pread: page-cache copies are memory-bound)
        const int fd = fileno(f);
        std::atomic<bool> ok{true};
        parallel_for(size, [&](uint64_t b, uint64_t e, unsigned) {
            uint64_t off = b;
            while (off < e) { const ssize_t r = pread(fd, &buf[off], e - off, (off_t)off); if (r <= 0) { ok = false; return; } off += (uint64_t)r; }
        }, 1u << 26);
        fclose(f);
        if (!ok) { err = "Could not read " + path; return false; }
        return true;
    }
    char tmp[1 << 16];   // (not a regular file: a pipe)
    size_t n;
    while ((n = fread(tmp, 1, sizeof tmp, f)) > 0) buf.append(tmp, n);
    fclose(f);
    return true;
}

// A file's bytes without a copy (round 6): a large regular, uncompressed file is mapped -- behind it one zero page, so that what a
// std::string guarantees, a readable NUL at [size], holds here too -- and its pages are faulted in by whichever thread parses
// them.  (read_file's buffer of a 5 GB VCF -- 100 000 samples -- was zero-filled on one thread before the first byte arrived:
// 1.4 s of the add mode's run.)  Anything else (.gz, a pipe, a small file, a mapping that fails) is read into `own` as before.
struct FileView {
    const char *data = nullptr;
    size_t size = 0;
    std::string own;
    void *map = nullptr;
    size_t map_len = 0;
    ~FileView() { if (map) munmap(map, map_len); }
};
bool view_file(const std::string &path, FileView &v, std::string &err) {
    if (path.find(".gz") == std::string::npos && !getenv("USHER_AMD_NO_MMAP")) {
        const int fd = open(path.c_str(), O_RDONLY);
        struct stat st;
        off_t least = (off_t)(64u << 20);
        if (const char *e = getenv("USHER_AMD_MMAP_MIN")) least = (off_t)atoll(e);   // (tests map their small fixtures)
        if (fd >= 0 && fstat(fd, &st) == 0 && S_ISREG(st.st_mode) && st.st_size > 0 && st.st_size >= least) {
            const size_t page = (size_t)sysconf(_SC_PAGESIZE), size = (size_t)st.st_size, len = (size + page) / page * page + page;
            void *base = mmap(nullptr, len, PROT_READ, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
            if (base != MAP_FAILED) {
                if (mmap(base, size, PROT_READ, MAP_PRIVATE | MAP_FIXED, fd, 0) == base) {
                    (void)madvise(base, size, MADV_WILLNEED);
                    close(fd);
                    v.map = base; v.map_len = len; v.data = (const char *)base; v.size = size;
                    return true;
                }
                munmap(base, len);
            }
        }
        if (fd >= 0) close(fd);
    }
    if (!read_file(path, v.own, err)) return false;
    v.data = v.own.data(); v.size = v.own.size();
    return true;
}

struct Rd {
    const uint8_t *p, *e;
    bool ok = true;
    uint64_t varint() {
        uint64_t v = 0; int sh = 0;
        while (p < e) { uint8_t b = *p++; v |= (uint64_t)(b & 0x7F) << sh; if (!(b & 0x80)) return v; sh += 7; if (sh > 63) break; }
        ok = false; return 0;
    }
    bool field(uint32_t &fno, uint32_t &wt, uint64_t &val, Rd &sub) {
        if (p >= e || !ok) return false;
        uint64_t key = varint();
        fno = (uint32_t)(key >> 3); wt = (uint32_t)(key & 7);
        if (wt == 0) val = varint();
        else if (wt == 2) { uint64_t n = varint(); if ((uint64_t)(e - p) < n) { ok = false; return false; } sub.p = p; sub.e = p + n; sub.ok = true; p += n; }
        else if (wt == 1) { if (e - p < 8) { ok = false; return false; } p += 8; }
        else if (wt == 5) { if (e - p < 4) { ok = false; return false; } p += 4; }
        else { ok = false; return false; }
        return ok;
    }
};

void put_varint(std::string &o, uint64_t v) {
    while (v >= 0x80) { o += (char)((v & 0x7F) | 0x80); v >>= 7; }
    o += (char)v;
}
void put_tag(std::string &o, uint32_t fno, uint32_t wt) { put_varint(o, ((uint64_t)fno << 3) | wt); }
void put_bytes(std::string &o, uint32_t fno, const std::string &b) { put_tag(o, fno, 2); put_varint(o, b.size()); o += b; }
void put_int32(std::string &o, uint32_t fno, int32_t v) {   // proto3: zero is omitted, negatives are sign-extended to 64 bits
    if (v == 0) return;
    put_tag(o, fno, 0);
    put_varint(o, (uint64_t)(int64_t)v);
}

}  // namespace

bool load_mat(const std::string &path, Tree &T, std::string &err) {   // :522-612
    std::string buf;
    Lap lap("load_mat");
    // A large .pb.gz is inflated on a thread of its own while this one scans the fields that are there already and the newick
    // string -- the first field -- is parsed on the host threads (round 5); anything else is read in one piece first.
    GzProgressive gz;
    bool progressive = path.find(".gz") != std::string::npos && !getenv("USHER_AMD_NO_GZ_PIPELINE") && gz.start(path);
    for (int attempt = 0; attempt < 2; attempt++) {
    if (!progressive) {
        if (!read_file(path, buf, err)) { err = "ERROR: Could not load the mutation-annotated tree object from file: " + path + "!"; return false; }
        lap("read file");
    }
    const uint8_t *base = (const uint8_t *)(progressive ? gz.buf.data() : buf.data());
    const size_t total = progressive ? gz.buf.size() : buf.size();
    const char *nwk_p = "";
    size_t nwk_len = 0;
    // First pass: the newick string, and where each node's mutation list / metadata entry sits in the buffer
    // (they are decoded straight into the nodes once the tree exists; no intermediate copies).
    std::vector<Rd> mut_lists, meta_lists;
    std::vector<std::pair<std::string, std::vector<std::string>>> cond;
    std::vector<Node *> order;
    std::thread nwk_thread;
    bool nwk_ok = true, nwk_started = false;
    std::string nwk_err;
    auto parse_newick = [&]() {
        // Large trees: items tokenised and nodes constructed on the host threads (tree_from_newick_bulk); the nodes come back in
        // creation order, which is the depth-first order the mutation lists are stored in (:552-554).
        const bool bulk = nwk_len >= (1u << 16) || getenv("USHER_AMD_GRAIN");
        if (bulk) nwk_ok = tree_from_newick_bulk(nwk_p, nwk_len, T, nwk_err, &order);
        else { nwk_ok = tree_from_newick(std::string(nwk_p, nwk_len), T, nwk_err); if (nwk_ok) order = T.dfs(); }
    };
    bool scan_ok = true;
    size_t off = 0;
    for (;;) {
        // (a field header is at most two varints: 20 bytes; the bytes of a field's body are only touched where noted)
        size_t have = total;
        if (progressive) { gz.wait_for(std::min(total, off + 20)); have = std::min(total, gz.avail.load(std::memory_order_acquire)); if (gz.state.load() < 0) break; }
        if (off >= have) break;
        Rd hd{base + off, base + have};
        const uint64_t key = hd.varint();
        const uint32_t fno = (uint32_t)(key >> 3), wt = (uint32_t)(key & 7);
        if (!hd.ok) { scan_ok = false; break; }
        if (wt == 0) { hd.varint(); if (!hd.ok) { scan_ok = false; break; } off = (size_t)(hd.p - base); continue; }
        if (wt == 1 || wt == 5) { off = (size_t)(hd.p - base) + (wt == 1 ? 8 : 4); if (off > total) { scan_ok = false; break; } continue; }
        if (wt != 2) { scan_ok = false; break; }
        const uint64_t n = hd.varint();
        if (!hd.ok) { scan_ok = false; break; }
        const size_t body = (size_t)(hd.p - base);
        if (n > total - body) { scan_ok = false; break; }
        Rd sub{base + body, base + body + n};
        off = body + (size_t)n;
        if (fno == 1) {
            nwk_p = (const char *)sub.p; nwk_len = (size_t)n;
            if (progressive && !nwk_started) {   // the tree is built while the rest of the file is still being inflated
                if (!gz.wait_for(off)) break;
                nwk_started = true;
                nwk_thread = std::thread(parse_newick);
            }
        } else if (fno == 2) mut_lists.push_back(sub);
        else if (fno == 3) {
            if (progressive && !gz.wait_for(off)) break;
            cond.emplace_back();
            uint32_t f2, w2; uint64_t v2; Rd s2{nullptr, nullptr};
            while (sub.field(f2, w2, v2, s2)) {
                if (f2 == 1 && w2 == 2) cond.back().first.assign((const char *)s2.p, s2.e - s2.p);
                else if (f2 == 2 && w2 == 2) cond.back().second.emplace_back((const char *)s2.p, s2.e - s2.p);
            }
        } else if (fno == 4) meta_lists.push_back(sub);
    }
    if (progressive) {
        const int fin = gz.finish();
        if (nwk_thread.joinable()) nwk_thread.join();
        if (fin == -2) {   // the trailer's size was no guide (several members, more than 4 GB): the simple way, from the start, on an empty tree
            free_all_nodes(T);
            T.root = nullptr; T.curr_internal_node = 0;
            progressive = false;
            continue;
        }
        if (fin != 1) { err = "ERROR: Could not load the mutation-annotated tree object from file: " + path + "!"; return false; }
        lap("inflate + top-level fields (pipelined)");
    }
    if (!scan_ok || off != total) { if (nwk_thread.joinable()) nwk_thread.join(); err = "malformed protobuf"; return false; }
    if (!progressive) lap("top-level fields");
    if (!nwk_started) parse_newick();
    if (!nwk_ok) { err = nwk_err; return false; }
    lap("tree from newick");
    if (mut_lists.size() < order.size()) { err = "protobuf has fewer mutation lists than tree nodes"; return false; }
    const bool hasmeta = !meta_lists.empty();
    if (!hasmeta) fprintf(stderr, "WARNING: This pb does not include any metadata. Filling in default values\n");
    if (hasmeta && meta_lists.size() < order.size()) { err = "protobuf has fewer metadata entries than tree nodes"; return false; }
    // Mutation lists and metadata decoded per node on the host threads (the reference: tbb::parallel_for over the nodes,
    // :556-612).  Chromosome names: every thread numbers the names it meets itself; the tables are merged in thread order --
    // threads own contiguous ranges of the depth-first order, so the merged numbering is the order of first appearance, as a
    // sequential load would give -- and the few threads whose numbering differs renumber their nodes' mutations.
    const unsigned TH = host_threads();
    std::vector<std::vector<std::string>> th_chroms(TH);
    std::vector<std::pair<uint64_t, uint64_t>> th_range(TH, {0, 0});
    std::vector<std::string> th_err(TH);
    parallel_for(order.size(), [&](uint64_t nb, uint64_t ne, unsigned tid) {
        th_range[tid] = {nb, ne};
        auto &chroms = th_chroms[tid];
        auto local_id = [&](const std::string &c) -> uint32_t {
            for (uint32_t i = 0; i < chroms.size(); i++) if (chroms[i] == c) return i;
            chroms.push_back(c);
            return (uint32_t)chroms.size() - 1;
        };
        std::string chrom, last_chrom;
        uint32_t last_chrom_id = local_id("");
        for (uint64_t i = nb; i < ne; i++) {
            Node *n = order[i];
            if (hasmeta) {
                Rd ml = meta_lists[i];
                uint32_t f2, w2; uint64_t v2; Rd s2{nullptr, nullptr};
                while (ml.field(f2, w2, v2, s2))
                    if (f2 == 1 && w2 == 2) n->clade_annotations.emplace_back((const char *)s2.p, s2.e - s2.p);
            }
            Rd list = mut_lists[i];
            uint32_t f2, w2; uint64_t v2; Rd s2{nullptr, nullptr};
            {   // (one allocation per node: count the entries first)
                Rd cnt = list; size_t k = 0; Rd t2{nullptr, nullptr};
                while (cnt.field(f2, w2, v2, t2)) k += (f2 == 1 && w2 == 2);
                if (k) n->mutations.reserve(k);
            }
            while (list.field(f2, w2, v2, s2)) {
                if (f2 != 1 || w2 != 2) continue;
                int32_t pos = 0, ref = 0, par = 0;
                int8_t nuc = 0;                                     // get_nuc_id(vector), :77-85
                chrom.clear();
                uint32_t f3, w3; uint64_t v3 = 0; Rd s3{nullptr, nullptr};
                while (s2.field(f3, w3, v3, s3)) {
                    if (f3 == 1 && w3 == 0) pos = (int32_t)(int64_t)v3;
                    else if (f3 == 2 && w3 == 0) ref = (int32_t)(int64_t)v3;
                    else if (f3 == 3 && w3 == 0) par = (int32_t)(int64_t)v3;
                    else if (f3 == 4 && w3 == 0) nuc = (int8_t)(nuc + (1 << (int32_t)(int64_t)v3));
                    else if (f3 == 4 && w3 == 2) { while (s3.p < s3.e && s3.ok) nuc = (int8_t)(nuc + (1 << (int32_t)(int64_t)s3.varint())); }
                    else if (f3 == 5 && w3 == 2) chrom.assign((const char *)s3.p, s3.e - s3.p);
                }
                if (!s2.ok) { th_err[tid] = "malformed protobuf (mut)"; return; }
                if (chrom != last_chrom) { last_chrom = chrom; last_chrom_id = local_id(chrom); }
                Mutation m;
                m.chrom = last_chrom_id;
                m.position = pos;
                if (pos >= 0) {
                    m.ref_nuc = (int8_t)(1 << ref);
                    m.par_nuc = (int8_t)(1 << par);
                    m.mut_nuc = nuc;
                    if (m.mut_nuc != m.par_nuc && !n->add_mutation(m)) { th_err[tid] = "add_mutation: mutations at the same position disagree"; return; }
                } else {
                    m.ref_nuc = m.par_nuc = m.mut_nuc = 0;
                    n->add_mutation(m);   // note: two masked entries cancel through the reversal rule, as in the reference
                }
            }
            if (!list.ok) { th_err[tid] = "malformed protobuf (mutation_list)"; return; }
        }
    }, 4096);
    lap("mutation lists");
    for (const std::string &e : th_err) if (!e.empty()) { err = e; return false; }   // (the first failing range in tree order)
    for (unsigned t = 0; t < TH; t++) {
        if (th_chroms[t].empty()) continue;
        std::vector<uint32_t> map(th_chroms[t].size());
        bool same = true;
        for (size_t k = 0; k < map.size(); k++) { map[k] = T.chrom_id(th_chroms[t][k]); same = same && map[k] == k; }
        if (!same)
            for (uint64_t i = th_range[t].first; i < th_range[t].second; i++) for (Mutation &m : order[i]->mutations) m.chrom = map[m.chrom];
    }
    for (auto &c : cond) {
        for (auto &l : c.second) T.condensed_leaves.insert(l);
        T.add_condensed(c.first, c.second);
    }
    return true;
    }   // (second attempt: a .gz whose trailer was no guide, read in one piece)
    err = "ERROR: Could not load the mutation-annotated tree object from file: " + path + "!";
    return false;
}

bool save_mat(Tree &T, const std::string &path, std::string &err) {   // :614-681
    std::string out;
    put_bytes(out, 1, newick(T, T.root, false, true));
    auto order = T.dfs();
    for (Node *n : order) {
        std::string ml;
        for (const Mutation &m : n->mutations) {
            std::string mm;
            put_int32(mm, 1, m.position);
            if (m.masked()) {
                put_int32(mm, 2, -1);
                put_int32(mm, 3, -1);
            } else {
                put_int32(mm, 2, nuc_index(m.ref_nuc));
                put_int32(mm, 3, nuc_index(m.par_nuc));
                std::string packed;
                for (int b = 0; b < 4; b++) if (m.mut_nuc & (1 << b)) put_varint(packed, (uint64_t)b);
                if (!packed.empty()) put_bytes(mm, 4, packed);
            }
            if (!T.chroms[m.chrom].empty()) put_bytes(mm, 5, T.chroms[m.chrom]);
            put_bytes(ml, 1, mm);
        }
        put_bytes(out, 2, ml);
    }
    for (const std::string &cname : T.condensed_order) {
        auto cit = T.condensed_nodes.find(cname);
        if (cit == T.condensed_nodes.end()) continue;
        auto &cn = *cit;
        std::string c;
        if (!cn.first.empty()) put_bytes(c, 1, cn.first);
        for (auto &l : cn.second) put_bytes(c, 2, l);
        put_bytes(out, 3, c);
    }
    for (Node *n : order) {
        std::string md;
        for (auto &a : n->clade_annotations) put_bytes(md, 1, a);
        put_bytes(out, 4, md);
    }
    if (path.find(".gz") != std::string::npos) {
        gzFile f = gzopen(path.c_str(), "wb");
        if (!f) { err = "Could not write " + path; return false; }
        bool ok = gzwrite(f, out.data(), (unsigned)out.size()) == (int)out.size();
        gzclose(f);
        if (!ok) err = "short write to " + path;
        return ok;
    }
    std::ofstream o(path, std::ios::binary);
    if (!o) { err = "Could not write " + path; return false; }
    o.write(out.data(), (std::streamsize)out.size());
    return (bool)o;
}

// --------------------------------------------------------------------- VCF

static void split_ws(const std::string &s, std::vector<std::string> &w) {
    size_t i = 0, n = s.size();
    while (i < n) {
        while (i < n && isspace((unsigned char)s[i])) i++;
        size_t j = i;
        while (j < n && !isspace((unsigned char)s[j])) j++;
        if (j > i) w.push_back(s.substr(i, j - i));
        i = j;
    }
}

static bool read_lines(const std::string &path, std::vector<std::string> &lines, std::string &err) {
    std::string buf;
    if (!read_file(path, buf, err)) { err = "ERROR: Could not open the VCF file: " + path + "!"; return false; }
    size_t s = 0;
    while (s < buf.size()) {
        size_t e = buf.find('\n', s);
        if (e == std::string::npos) e = buf.size();
        lines.push_back(buf.substr(s, e - s));
        s = e + 1;
    }
    return true;
}

// Existing-MAT branch of read_vcf (:2180-2277; the reference reads the file through a tbb::flow pipeline, :2108-2179).  The file is
// read in one piece, its line starts are found on the host threads, the header is handled where it stands, and the data lines
// are parsed in contiguous blocks, one per thread, without copying a word: every thread leaves the cells of its lines as
// (sample, mutation) records in line order, and the samples' lists are the concatenation of the blocks in file order -- the
// order a sequential reader appends in.  Quirks kept: lines in front of the header are skipped unless they look like one; a
// blank line is skipped; a genotype that starts with a digit is read as its leading digits; anything else is a missing call;
// the first letter of the ALT allele decides ('N' and every unknown letter: missing).
bool read_vcf_missing(Tree &T, const std::string &path, std::vector<MissingSample> &out, std::string &err) {
    Lap lap("read_vcf");
    FileView fv;
    if (!view_file(path, fv, err)) { err = "ERROR: Could not open the VCF file: " + path + "!"; return false; }
    lap("read file");
    const char *base = fv.data;
    const size_t len = fv.size;
    const unsigned TH = host_threads();
    std::vector<std::vector<size_t>> nl(TH);
    parallel_for(len, [&](uint64_t b, uint64_t e, unsigned tid) {
        auto &v = nl[tid];
        for (const char *p = base + b, *pe = base + e; (p = (const char *)memchr(p, '\n', (size_t)(pe - p))) != nullptr; p++) v.push_back((size_t)(p - base));
    }, 1u << 20);
    std::vector<size_t> ls{0};   // line i = [ls[i], ls[i + 1] - 1)
    for (auto &v : nl) for (size_t x : v) ls.push_back(x + 1);
    if (ls.back() < len) ls.push_back(len + 1);
    const size_t n_lines = ls.size() - 1;
    lap("line starts");
    auto is_sp = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f' || c == '\n'; };   // isspace
    // next word of [p, e): begin / end, or false
    auto next_word = [&](const char *&p, const char *e, const char *&wb, const char *&we) {
        while (p < e && is_sp(*p)) p++;
        if (p >= e) return false;
        wb = p;
        while (p < e && !is_sp(*p)) p++;
        we = p;
        return true;
    };
    // the header: the first line with more than one word whose second word is POS
    size_t n_ids = 0, first_data = n_lines;
    std::vector<size_t> missing_idx;
    for (size_t li = 0; li < n_lines; li++) {
        const char *p = base + ls[li], *e = base + ls[li + 1] - 1, *wb, *we;
        if (!next_word(p, e, wb, we) || !next_word(p, e, wb, we)) continue;
        if ((size_t)(we - wb) != 3 || memcmp(wb, "POS", 3) != 0) continue;
        const char *q = base + ls[li];
        for (size_t j = 0; next_word(q, e, wb, we); j++) {
            if (j < 9) continue;
            n_ids++;
            const std::string name(wb, (size_t)(we - wb));
            if (!T.get_node(name) && !T.condensed_leaves.count(name)) {
                MissingSample ms; ms.name = name;
                out.push_back(std::move(ms));
                missing_idx.push_back(j);
            } else {
                fprintf(stderr, "WARNING: Ignoring sample %s as it is already in the tree.\n", name.c_str());
            }
        }
        first_data = li + 1;
        break;
    }
    lap("header");
    if (first_data >= n_lines) return true;
    // column j -> index into `out`, or -1
    std::vector<int32_t> col2k(9 + n_ids, -1);
    for (size_t k = 0; k < missing_idx.size(); k++) col2k[missing_idx[k]] = (int32_t)k;
    struct Rec { uint32_t k; Mutation m; };
    struct Block { std::vector<Rec> recs; std::vector<std::string> chroms; size_t err_line = SIZE_MAX; std::string err; uint64_t lb = 0, le = 0; };
    std::vector<Block> blocks(TH);
    parallel_for(n_lines - first_data, [&](uint64_t b0, uint64_t e0, unsigned tid) {
        Block &B = blocks[tid];
        B.lb = first_data + b0; B.le = first_data + e0;
        auto local_chrom = [&](const char *b, const char *e) -> uint32_t {
            const size_t n = (size_t)(e - b);
            for (uint32_t i = 0; i < B.chroms.size(); i++) if (B.chroms[i].size() == n && memcmp(B.chroms[i].data(), b, n) == 0) return i;
            B.chroms.emplace_back(b, n);
            return (uint32_t)B.chroms.size() - 1;
        };
        std::vector<std::pair<const char *, const char *>> alleles;
        for (uint64_t li = B.lb; li < B.le; li++) {
            const char *p = base + ls[li], *e = base + ls[li + 1] - 1, *wb, *we;
            const char *f[9][2];
            size_t nw = 0;
            while (nw < 9 && next_word(p, e, wb, we)) { f[nw][0] = wb; f[nw][1] = we; nw++; }
            if (nw == 0) continue;                                   // blank line
            // the cells; a line that is short or long is an error (:2226-2231)
            Mutation proto;
            int8_t ref = 0;
            if (nw == 9) {
                proto.chrom = local_chrom(f[0][0], f[0][1]);
                proto.position = (int32_t)strtol(std::string(f[1][0], f[1][1]).c_str(), nullptr, 10);
                ref = nuc_id(f[3][0][0]);
                proto.ref_nuc = ref; proto.par_nuc = ref;
                alleles.clear();
                for (const char *a = f[4][0]; a < f[4][1];) {          // string_split on ',': empty pieces kept, a trailing one dropped
                    const char *c = (const char *)memchr(a, ',', (size_t)(f[4][1] - a));
                    if (!c) { alleles.push_back({a, f[4][1]}); break; }
                    alleles.push_back({a, c});
                    a = c + 1;
                }
            }
            size_t col = nw;
            bool bad = false;
            for (;;) {
                // (the common cell: a tab and a lone 0 -- the reference allele, no record)
                while (p + 1 < e && p[0] == '\t' && p[1] == '0' && (p + 2 == e || p[2] == '\t')) { p += 2; col++; }
                if (!next_word(p, e, wb, we)) break;
                if (col >= 9 + n_ids) { col++; continue; }
                const int32_t k = col2k[col];
                col++;
                if (k < 0) continue;
                Mutation m = proto;
                if (*wb >= '0' && *wb <= '9') {
                    long allele_id = 0;                                  // std::stoi: the leading digits
                    for (const char *c = wb; c < we && *c >= '0' && *c <= '9'; c++) { allele_id = allele_id * 10 + (*c - '0'); if (allele_id > 1000000) break; }
                    if (allele_id <= 0) continue;
                    if ((size_t)allele_id > alleles.size()) { B.err_line = li; B.err = "ERROR! VCF genotype refers to a missing ALT allele."; bad = true; break; }
                    const auto &al = alleles[allele_id - 1];
                    const char a0 = al.first < al.second ? *al.first : '\0';
                    if (a0 == 'N') { m.is_missing = true; m.mut_nuc = 15; }
                    else { m.mut_nuc = nuc_id(a0); m.is_missing = (m.mut_nuc == 15); }
                } else {
                    m.is_missing = true; m.mut_nuc = 15;
                }
                B.recs.push_back({(uint32_t)k, m});
            }
            if (bad) return;
            if (col != 9 + n_ids) {
                B.err_line = li;
                B.err = "ERROR! Incorrect VCF format. Expected " + std::to_string(9 + n_ids) + " columns but got " + std::to_string(col) + ".";
                return;
            }
        }
    }, 16);
    lap("data lines");
    // the first error in file order wins -- but the records in front of it were already appended by a sequential reader; nobody
    // uses them after an error, so they are not reproduced
    {
        size_t el = SIZE_MAX; const Block *eb = nullptr;
        for (const Block &B : blocks) if (B.err_line < el) { el = B.err_line; eb = &B; }
        if (eb) { err = eb->err; return false; }
    }
    // chromosome numbers in order of first appearance; records into the samples' lists, block after block
    std::vector<size_t> per(out.size(), 0);
    for (Block &B : blocks) {
        std::vector<uint32_t> map(B.chroms.size());
        for (size_t k = 0; k < map.size(); k++) map[k] = T.chrom_id(B.chroms[k]);
        for (Rec &r : B.recs) { r.m.chrom = map[r.m.chrom]; per[r.k]++; }
    }
    for (size_t k = 0; k < out.size(); k++) out[k].mutations.reserve(out[k].mutations.size() + per[k]);
    parallel_for(out.size(), [&](uint64_t kb, uint64_t ke, unsigned) {   // (every thread fills its own samples: it scans all records when there are few threads' worth of them)
        for (const Block &B : blocks)
            for (const Rec &r : B.recs)
                if (r.k >= kb && r.k < ke) {
                    if (r.m.mut_nuc & (r.m.mut_nuc - 1)) out[r.k].num_ambiguous++;
                    out[r.k].mutations.push_back(r.m);
                }
    }, 64);
    lap("merge");
    return true;
}

bool read_vcf_build(Tree &T, const std::string &path, std::vector<MissingSample> &out, std::string &err, AssignFn assign, void *ctx) {   // :2052-2179
    std::vector<std::string> lines;
    if (!read_lines(path, lines, err)) return false;
    auto bfs = T.bfs();
    std::unordered_map<std::string, uint32_t> idx;
    SiteBatch batch;
    batch.parent.assign(bfs.size(), UINT32_MAX);
    for (uint32_t j = 0; j < bfs.size(); j++) idx[bfs[j]->id] = j;
    for (uint32_t j = 0; j < bfs.size(); j++)
        if (bfs[j]->parent) batch.parent[j] = idx[bfs[j]->parent->id];
    std::vector<int32_t> site_pos;
    std::vector<uint32_t> site_chrom;
    std::vector<std::pair<uint32_t, uint8_t>> line_cells;
    bool header_found = false;
    std::vector<std::string> ids;
    std::vector<int64_t> col_node;      // per VCF column: BFS index, or -1 - (index into out)
    for (const std::string &s : lines) {
        std::vector<std::string> words;
        split_ws(s, words);
        if (!header_found && words.size() > 1) {
            if (words[1] == "POS") {
                for (size_t j = 9; j < words.size(); j++) {
                    ids.push_back(words[j]);
                    auto it = idx.find(words[j]);
                    if (it == idx.end()) {
                        MissingSample ms; ms.name = words[j];
                        col_node.push_back(-1 - (int64_t)out.size());
                        out.push_back(std::move(ms));
                    } else col_node.push_back(it->second);
                }
                header_found = true;
            }
        } else if (header_found) {
            if (words.empty()) continue;
            if (words.size() != 9 + ids.size()) { err = "ERROR! Incorrect VCF format."; return false; }
            std::vector<std::string> alleles;
            split(words[4], ',', alleles);
            const int32_t pos = (int32_t)strtol(words[1].c_str(), nullptr, 10);
            const int8_t ref = nuc_id(words[3][0]);
            if (nuc_index(ref) < 0) { err = "ERROR! VCF REF base is not one of A,C,G,T."; return false; }
            const uint32_t chrom = T.chrom_id(words[0]);
            fprintf(stderr, "At variant site %i\n", pos);
            line_cells.clear();
            for (size_t j = 9; j < words.size(); j++) {
                int8_t nuc;
                if (isdigit((unsigned char)words[j][0])) {
                    const long a = strtol(words[j].c_str(), nullptr, 10);
                    if (a <= 0) continue;
                    if ((size_t)a > alleles.size()) { err = "ERROR! VCF genotype refers to a missing ALT allele."; return false; }
                    nuc = nuc_id(alleles[a - 1][0]);
                } else nuc = 15;
                const int64_t c = col_node[j - 9];
                if (c >= 0) line_cells.emplace_back((uint32_t)c, (uint8_t)nuc);
                else {   // sample to be placed later: keep its row (usher_mapper.cpp:65-82)
                    Mutation m;
                    m.chrom = chrom; m.position = pos; m.ref_nuc = ref; m.par_nuc = ref;
                    if (nuc == 15) { m.is_missing = true; m.mut_nuc = 15; } else m.mut_nuc = nuc;
                    out[(size_t)(-1 - c)].mutations.push_back(m);
                }
            }
            // cells in ascending node order (what the backend's fast path wants); a node named by several
            // columns keeps the cell of the last one, as the in-order loop of usher_mapper.cpp:47-62 does
            std::stable_sort(line_cells.begin(), line_cells.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
            for (size_t k = 0; k < line_cells.size(); k++) {
                if (k + 1 < line_cells.size() && line_cells[k + 1].first == line_cells[k].first) continue;
                batch.var_node.push_back(line_cells[k].first); batch.var_nuc.push_back(line_cells[k].second);
            }
            batch.ref.push_back((uint8_t)ref);
            batch.var_off.push_back(batch.var_node.size());
            site_pos.push_back(pos);
            site_chrom.push_back(chrom);
        }
    }
    if (!assign) { err = "ERROR: no Fitch-Sankoff backend (the GPU library is required to build a MAT from a tree)."; return false; }
    SiteMutations muts;
    if (!assign(ctx, batch, muts, err)) return false;
    for (size_t i = 0; i < muts.site.size(); i++) {                             // usher_mapper.cpp:143-156
        if (muts.site[i] >= site_pos.size() || muts.node[i] >= bfs.size()) { err = "ERROR: Fitch-Sankoff backend returned an out-of-range index."; return false; }
        Mutation m;
        m.chrom = site_chrom[muts.site[i]]; m.position = site_pos[muts.site[i]]; m.ref_nuc = (int8_t)batch.ref[muts.site[i]];
        m.par_nuc = (int8_t)muts.par_nuc[i]; m.mut_nuc = (int8_t)muts.mut_nuc[i];
        bfs[muts.node[i]]->add_mutation(m);
    }
    return true;
}

// ------------------------------------------------------- bench / test utilities
// A tree given as breadth-first arrays (the layout of ugp_tree_desc) written as parsimony.proto, and a query batch in CSR form
// written as a VCF: the inputs of an end-to-end run of the front end on the synthetic workload (bench.py, tools/bench_addmode.py).

bool write_pb_from_arrays(uint64_t n, const uint32_t *parent, const uint64_t *mut_off, const int32_t *pos, const uint8_t *ref, const uint8_t *par,
                          const uint8_t *nuc, const std::string &path, std::string &err) {
    if (!n) { err = "empty tree"; return false; }
    std::vector<uint64_t> first(n + 1, 0);
    for (uint64_t j = 1; j < n; j++) first[parent[j] + 1]++;
    for (uint64_t j = 0; j < n; j++) first[j + 1] += first[j];
    std::vector<uint32_t> kids(n ? n - 1 : 0);
    { std::vector<uint64_t> fill(first.begin(), first.end() - 1); for (uint64_t j = 1; j < n; j++) kids[fill[parent[j]]++] = (uint32_t)j; }
    std::string nwk;
    nwk.reserve(n * 14);
    std::vector<uint32_t> order;
    order.reserve(n);
    std::vector<std::pair<uint32_t, uint32_t>> st{{0u, 0u}};
    char tmp[48];
    auto len_of = [&](uint32_t j) { snprintf(tmp, sizeof tmp, ":%g", (float)(mut_off[j + 1] - mut_off[j])); return tmp; };
    while (!st.empty()) {
        auto &fr = st.back();
        const uint32_t j = fr.first;
        const uint64_t b = first[j], e = first[j + 1];
        if (fr.second == 0) {
            order.push_back(j);
            if (b == e) { snprintf(tmp, sizeof tmp, "L%u", j); nwk += tmp; nwk += len_of(j); st.pop_back(); continue; }
            nwk += '(';
        }
        if (b + fr.second < e) {
            if (fr.second) nwk += ',';
            const uint32_t c = kids[b + fr.second++];
            st.push_back({c, 0u});
        } else { nwk += ')'; nwk += len_of(j); st.pop_back(); }
    }
    nwk += ';';
    std::string out;
    out.reserve(nwk.size() + (mut_off[n] * 14) + n * 2 + 64);
    put_bytes(out, 1, nwk);
    std::string ml, mm, packed;
    for (uint32_t j : order) {
        ml.clear();
        for (uint64_t i = mut_off[j]; i < mut_off[j + 1]; i++) {
            mm.clear();
            put_int32(mm, 1, pos[i]);
            if (pos[i] < 0) { put_int32(mm, 2, -1); put_int32(mm, 3, -1); }
            else {
                put_int32(mm, 2, nuc_index((int8_t)ref[i]));
                put_int32(mm, 3, nuc_index((int8_t)par[i]));
                packed.clear();
                for (int b = 0; b < 4; b++) if (nuc[i] & (1 << b)) put_varint(packed, (uint64_t)b);
                if (!packed.empty()) put_bytes(mm, 4, packed);
            }
            put_bytes(ml, 1, mm);
        }
        put_bytes(out, 2, ml);
    }
    // one (empty) node_metadata per node, as save_mutation_annotated_tree always writes (:619-623); serialised behind the mutation
    // lists (field 4).  Round 6: the file is then exactly what save_mat() gives for the same tree.
    for (uint64_t j = 0; j < n; j++) put_bytes(out, 4, std::string());
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { err = "Could not write " + path; return false; }
    const bool ok = fwrite(out.data(), 1, out.size(), f) == out.size();
    fclose(f);
    if (!ok) err = "short write to " + path;
    return ok;
}

bool write_vcf_from_csr(uint64_t n_samples, const uint64_t *ent_off, const int32_t *pos, const uint8_t *ref, const uint8_t *nuc, const uint8_t *is_missing,
                        const std::string &prefix, const std::string &path, std::string &err) {
    struct Cell { int32_t pos; uint32_t sample; uint8_t ref, allele; };
    std::vector<Cell> cells;
    cells.reserve(ent_off[n_samples]);
    for (uint64_t s = 0; s < n_samples; s++)
        for (uint64_t i = ent_off[s]; i < ent_off[s + 1]; i++) cells.push_back({pos[i], (uint32_t)s, ref[i], (uint8_t)(is_missing[i] ? 15 : nuc[i])});
    std::stable_sort(cells.begin(), cells.end(), [](const Cell &a, const Cell &b) { return a.pos < b.pos; });
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { err = "Could not write " + path; return false; }
    std::string line = "##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT";
    for (uint64_t s = 0; s < n_samples; s++) { line += '\t'; line += prefix; line += std::to_string(s); }
    line += '\n';
    fwrite(line.data(), 1, line.size(), f);
    std::string gt;
    for (size_t b = 0; b < cells.size();) {
        size_t e = b;
        while (e < cells.size() && cells[e].pos == cells[b].pos) e++;
        const uint8_t r = cells[b].ref;
        std::vector<uint8_t> alts;
        for (size_t k = b; k < e; k++) if (cells[k].allele != 15 && cells[k].allele != r && std::find(alts.begin(), alts.end(), cells[k].allele) == alts.end()) alts.push_back(cells[k].allele);
        std::sort(alts.begin(), alts.end());
        if (alts.empty()) for (uint8_t a : {1, 2, 4, 8}) if (a != r) { alts.push_back(a); break; }
        line = "chr\t" + std::to_string(cells[b].pos) + "\t.\t" + std::string(1, nuc_char((int8_t)r)) + "\t";
        for (size_t k = 0; k < alts.size(); k++) { if (k) line += ','; line += nuc_char((int8_t)alts[k]); }
        line += "\t.\t.\t.\tGT";
        gt.assign(n_samples * 2, '\t');
        for (uint64_t s = 0; s < n_samples; s++) gt[2 * s + 1] = '0';
        bool wide = false;
        for (size_t k = b; k < e; k++) {
            char c = '.';
            if (cells[k].allele != 15) {
                const size_t idx = cells[k].allele == r ? 0 : (size_t)(std::find(alts.begin(), alts.end(), cells[k].allele) - alts.begin()) + 1;
                if (idx > 9) wide = true;
                c = (char)('0' + idx);
            }
            gt[2 * cells[k].sample + 1] = c;
        }
        if (wide) { fclose(f); err = "more than 9 ALT alleles at one position"; return false; }
        line += gt;
        line += '\n';
        fwrite(line.data(), 1, line.size(), f);
        b = e;
    }
    fclose(f);
    return true;
}

}  // namespace uh
