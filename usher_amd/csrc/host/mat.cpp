// mat.cpp -- host-side MAT model, newick / parsimony.proto / VCF readers and
// writers.  Written from scratch; each function cites the reference behaviour
// it reproduces (src/mutation_annotated_tree.cpp unless noted).
#include "mat.hpp"

#include <zlib.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>

namespace uh {

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

Tree::~Tree() {
    for (auto &kv : all_nodes) delete kv.second;
}

uint32_t Tree::chrom_id(const std::string &c) {
    for (uint32_t i = 0; i < chroms.size(); i++) if (chroms[i] == c) return i;
    chroms.push_back(c);
    return (uint32_t)chroms.size() - 1;
}

Node *Tree::get_node(const std::string &id) const {
    auto it = all_nodes.find(id);
    return it == all_nodes.end() ? nullptr : it->second;
}

Node *Tree::create_node(const std::string &id, Node *parent, float branch_length) {   // :881-910
    if (parent) {
        auto ins = all_nodes.emplace(id, nullptr);
        if (!ins.second) return nullptr;   // "already in the tree"
        Node *n = new Node();
        n->id = id;
        n->parent = parent;
        n->branch_length = branch_length;
        n->level = parent->level + 1;
        const size_t na = num_annotations();
        if (na) n->clade_annotations.assign(na, "");
        parent->children.push_back(n);
        ins.first->second = n;
        return n;
    }
    Node *n = new Node();
    n->id = id;
    n->parent = parent;
    n->branch_length = branch_length;
    n->level = parent ? parent->level + 1 : 1;
    if (!parent) {
        for (auto &kv : all_nodes) delete kv.second;
        all_nodes.clear();
        root = n;
    } else {
        n->clade_annotations.assign(num_annotations(), "");
        parent->children.push_back(n);
    }
    all_nodes[id] = n;
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
    for (auto &kv : all_nodes) s += kv.second->mutations.size();
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
    delete n;
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
                delete par;
            }
        }
    }
    std::vector<Node *> q{source};
    for (size_t h = 0; h < q.size(); h++) for (Node *c : q[h]->children) q.push_back(c);
    for (Node *n : q) { all_nodes.erase(n->id); delete n; }
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
            all_nodes[n->id] = n;
            for (size_t s = 0; s < k; s++) {
                Node *c = new Node();
                c->id = cn.second[s]; c->parent = n; c->branch_length = -1.0f; c->level = n->level + 1;
                c->clade_annotations.assign(num_annotations(), "");
                all_nodes[c->id] = c;
                n->children.push_back(c);
            }
        } else if (k > 1) {
            all_nodes.erase(n->id);
            n->id = cn.second[0];
            all_nodes[n->id] = n;
            for (size_t s = 1; s < k; s++) {
                Node *c = new Node();
                c->id = cn.second[s]; c->parent = par; c->branch_length = n->branch_length; c->level = par->level + 1;
                c->clade_annotations.assign(num_annotations(), "");
                all_nodes[c->id] = c;
                par->children.push_back(c);
            }
        } else if (k == 1) {
            all_nodes.erase(n->id);
            n->id = cn.second[0];
            all_nodes[n->id] = n;
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

bool tree_from_newick(const std::string &nwk, Tree &T, std::string &err) {   // :415-508
    std::vector<std::string> parts;
    split(nwk, ',', parts);
    struct Item { std::string leaf; size_t open = 0, close = 0; };
    std::vector<Item> items;
    std::vector<std::vector<float>> blen(128);
    size_t level = 0;
    auto to_len = [](const std::string &b) { return b.empty() ? -1.0f : std::stof(b); };
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
    if (level != 0) { err = "incorrect Newick format"; return false; }
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

static void put_len(std::string &out, float v) {   // operator<<(float): %g
    char buf[32];
    snprintf(buf, sizeof buf, "%g", v);
    out += buf;
}

std::string newick(const Tree &T, Node *from, bool internal_ids, bool branch_len, bool uncondense) {   // :215-346
    std::string out;
    Node *start = from ? from : T.root;
    if (!start) return ";";
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

bool read_file(const std::string &path, std::string &buf, std::string &err) {
    if (path.find(".gz") != std::string::npos) {   // :530 / :2086
        gzFile f = gzopen(path.c_str(), "rb");
        if (!f) { err = "Could not open " + path; return false; }
        char tmp[1 << 16];
        int n;
        while ((n = gzread(f, tmp, sizeof tmp)) > 0) buf.append(tmp, (size_t)n);
        gzclose(f);
        return n == 0;
    }
    std::ifstream in(path, std::ios::binary);
    if (!in) { err = "Could not open " + path; return false; }
    std::stringstream ss;
    ss << in.rdbuf();
    buf = ss.str();
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
    if (!read_file(path, buf, err)) { err = "ERROR: Could not load the mutation-annotated tree object from file: " + path + "!"; return false; }
    Rd top{(const uint8_t *)buf.data(), (const uint8_t *)buf.data() + buf.size()};
    std::string nwk;
    // First pass: the newick string, and where each node's mutation list / metadata entry sits in the buffer
    // (they are decoded straight into the nodes once the tree exists; no intermediate copies).
    std::vector<Rd> mut_lists, meta_lists;
    std::vector<std::pair<std::string, std::vector<std::string>>> cond;
    uint32_t fno, wt; uint64_t val; Rd sub{nullptr, nullptr};
    while (top.field(fno, wt, val, sub)) {
        if (fno == 1 && wt == 2) nwk.assign((const char *)sub.p, sub.e - sub.p);
        else if (fno == 2 && wt == 2) mut_lists.push_back(sub);
        else if (fno == 3 && wt == 2) {
            cond.emplace_back();
            uint32_t f2, w2; uint64_t v2; Rd s2{nullptr, nullptr};
            while (sub.field(f2, w2, v2, s2)) {
                if (f2 == 1 && w2 == 2) cond.back().first.assign((const char *)s2.p, s2.e - s2.p);
                else if (f2 == 2 && w2 == 2) cond.back().second.emplace_back((const char *)s2.p, s2.e - s2.p);
            }
        } else if (fno == 4 && wt == 2) meta_lists.push_back(sub);
    }
    if (!top.ok) { err = "malformed protobuf"; return false; }
    if (!tree_from_newick(nwk, T, err)) return false;
    auto order = T.dfs();
    if (mut_lists.size() < order.size()) { err = "protobuf has fewer mutation lists than tree nodes"; return false; }
    const bool hasmeta = !meta_lists.empty();
    if (!hasmeta) fprintf(stderr, "WARNING: This pb does not include any metadata. Filling in default values\n");
    if (hasmeta && meta_lists.size() < order.size()) { err = "protobuf has fewer metadata entries than tree nodes"; return false; }
    std::string chrom, last_chrom;
    uint32_t last_chrom_id = T.chrom_id("");
    for (size_t i = 0; i < order.size(); i++) {
        Node *n = order[i];
        if (hasmeta) {
            Rd ml = meta_lists[i];
            uint32_t f2, w2; uint64_t v2; Rd s2{nullptr, nullptr};
            while (ml.field(f2, w2, v2, s2))
                if (f2 == 1 && w2 == 2) n->clade_annotations.emplace_back((const char *)s2.p, s2.e - s2.p);
        }
        Rd list = mut_lists[i];
        uint32_t f2, w2; uint64_t v2; Rd s2{nullptr, nullptr};
        while (list.field(f2, w2, v2, s2)) {
            if (f2 != 1 || w2 != 2) continue;
            int32_t pos = 0, ref = 0, par = 0;
            int8_t nuc = 0;                                     // get_nuc_id(vector), :77-85
            chrom.clear();
            uint32_t f3, w3; uint64_t v3; Rd s3{nullptr, nullptr};
            while (s2.field(f3, w3, v3, s3)) {
                if (f3 == 1 && w3 == 0) pos = (int32_t)(int64_t)v3;
                else if (f3 == 2 && w3 == 0) ref = (int32_t)(int64_t)v3;
                else if (f3 == 3 && w3 == 0) par = (int32_t)(int64_t)v3;
                else if (f3 == 4 && w3 == 0) nuc = (int8_t)(nuc + (1 << (int32_t)(int64_t)v3));
                else if (f3 == 4 && w3 == 2) { while (s3.p < s3.e && s3.ok) nuc = (int8_t)(nuc + (1 << (int32_t)(int64_t)s3.varint())); }
                else if (f3 == 5 && w3 == 2) chrom.assign((const char *)s3.p, s3.e - s3.p);
            }
            if (!s2.ok) { err = "malformed protobuf (mut)"; return false; }
            if (chrom != last_chrom) { last_chrom = chrom; last_chrom_id = T.chrom_id(chrom); }
            Mutation m;
            m.chrom = last_chrom_id;
            m.position = pos;
            if (pos >= 0) {
                m.ref_nuc = (int8_t)(1 << ref);
                m.par_nuc = (int8_t)(1 << par);
                m.mut_nuc = nuc;
                if (m.mut_nuc != m.par_nuc && !n->add_mutation(m)) { err = "add_mutation: mutations at the same position disagree"; return false; }
            } else {
                m.ref_nuc = m.par_nuc = m.mut_nuc = 0;
                n->add_mutation(m);   // note: two masked entries cancel through the reversal rule, as in the reference
            }
        }
        if (!list.ok) { err = "malformed protobuf (mutation_list)"; return false; }
    }
    for (auto &c : cond) {
        for (auto &l : c.second) T.condensed_leaves.insert(l);
        T.add_condensed(c.first, c.second);
    }
    return true;
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

bool read_vcf_missing(Tree &T, const std::string &path, std::vector<MissingSample> &out, std::string &err) {   // :2180-2277
    std::vector<std::string> lines;
    if (!read_lines(path, lines, err)) return false;
    bool header_found = false;
    size_t n_ids = 0;
    std::vector<size_t> missing_idx;
    for (const std::string &s : lines) {
        std::vector<std::string> words;
        split_ws(s, words);
        if (!header_found && words.size() > 1) {
            if (words[1] == "POS") {
                for (size_t j = 9; j < words.size(); j++) {
                    n_ids++;
                    if (!T.get_node(words[j]) && !T.condensed_leaves.count(words[j])) {
                        MissingSample ms; ms.name = words[j];
                        out.push_back(std::move(ms));
                        missing_idx.push_back(j);
                    } else {
                        fprintf(stderr, "WARNING: Ignoring sample %s as it is already in the tree.\n", words[j].c_str());
                    }
                }
                header_found = true;
            }
        } else if (header_found) {
            if (words.empty()) continue;
            if (words.size() != 9 + n_ids) {
                err = "ERROR! Incorrect VCF format. Expected " + std::to_string(9 + n_ids) + " columns but got " + std::to_string(words.size()) + ".";
                return false;
            }
            std::vector<std::string> alleles;
            split(words[4], ',', alleles);
            const int32_t pos = (int32_t)strtol(words[1].c_str(), nullptr, 10);
            const int8_t ref = nuc_id(words[3][0]);
            const uint32_t chrom = T.chrom_id(words[0]);
            for (size_t k = 0; k < missing_idx.size(); k++) {
                const std::string &cell = words[missing_idx[k]];
                Mutation m;
                m.chrom = chrom; m.position = pos; m.ref_nuc = ref; m.par_nuc = ref;
                if (isdigit((unsigned char)cell[0])) {
                    const long allele_id = strtol(cell.c_str(), nullptr, 10);   // std::stoi: leading digits
                    if (allele_id <= 0) continue;
                    if ((size_t)allele_id > alleles.size()) { err = "ERROR! VCF genotype refers to a missing ALT allele."; return false; }
                    const std::string &allele = alleles[allele_id - 1];
                    if (allele[0] == 'N') { m.is_missing = true; m.mut_nuc = 15; }
                    else { m.mut_nuc = nuc_id(allele[0]); m.is_missing = (m.mut_nuc == 15); }
                } else {
                    m.is_missing = true; m.mut_nuc = 15;
                }
                if (m.mut_nuc & (m.mut_nuc - 1)) out[k].num_ambiguous++;
                out[k].mutations.push_back(m);
            }
        }
    }
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

}  // namespace uh
