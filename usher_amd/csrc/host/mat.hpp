// mat.hpp -- host-side mutation-annotated tree (MAT) model, readers and writers
// for the usher-compatible front end.  Written from scratch; behaviour follows
// the reference's src/mutation_annotated_tree.{hpp,cpp} (cited per function).
#pragma once
#include <cstdint>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace uh {

// one-hot allele codes A=1 C=2 G=4 T=8, IUPAC = unions, N = 15
int8_t nuc_id(char c);        // mutation_annotated_tree.cpp:19-74 (incl. the 'V' -> N fall-through)
char nuc_char(int8_t id);     // :88-139
int8_t nuc_index(int8_t id);  // :142-162  (A,C,G,T -> 0..3, else -1)

struct Mutation {             // mutation_annotated_tree.hpp:45-86
    int32_t position = 0;     // < 0 = masked
    int8_t ref_nuc = 0, par_nuc = 0, mut_nuc = 0;
    bool is_missing = false;
    uint32_t chrom = 0;       // index into Tree::chroms
    bool masked() const { return position < 0; }
    std::string str() const;  // "A123G" / "MASKED"
};

struct Node {                 // mutation_annotated_tree.hpp:88-111
    std::string id;
    Node *parent = nullptr;
    std::vector<Node *> children;
    std::vector<Mutation> mutations;      // sorted by position
    std::vector<std::string> clade_annotations;
    float branch_length = -1.0f;
    size_t level = 1;
    // scratch for the placement driver: breadth-first index in its last flattening of the tree (valid
    // while flat_epoch equals the flattening's epoch) -- avoids a 10M-entry hash map per flattening
    mutable uint32_t flat_index = 0, flat_epoch = 0;
    bool in_block = false;    // storage belongs to one of the tree's node blocks (bulk load), not to `new`
    bool is_leaf() const { return children.empty(); }
    bool is_root() const { return parent == nullptr; }
    // :720-752.  Returns false on the reference's "called out of order" error.
    bool add_mutation(const Mutation &m);
};

// name -> node (the reference's Tree::all_nodes, mutation_annotated_tree.hpp:121).  A 10M-node load cannot afford 10M
// sequential inserts into a std::unordered_map (1.5 s of the 6 s a load took): the nodes a bulk load creates are indexed
// by a table of (hash, node) sorted on the host threads; nodes created later go to an ordinary hash map in front of it.
class NodeIndex {
  public:
    Node *find(const std::string &id) const;
    bool insert(const std::string &id, Node *n);   // false (and no change) when the name is taken
    void set(const std::string &id, Node *n);      // insert or overwrite
    bool erase(const std::string &id);
    size_t size() const { return live_; }
    void clear() { base_.clear(); extra_.clear(); live_ = 0; }
    void reserve(size_t n) { if (base_.empty()) extra_.reserve(n); }
    // index `n` nodes at once (replaces the table; the map of later nodes is kept); false = two nodes share a name (*dup)
    bool bulk_build(Node *const *nodes, size_t n, std::string *dup);
    template <class F> void for_each(F f) const {
        for (const auto &e : base_) if (e.second) f(e.second);
        for (const auto &kv : extra_) f(kv.second);
    }
    static uint64_t hash(const std::string &s);
  private:
    std::vector<std::pair<uint64_t, Node *>> base_;   // sorted by hash; node == nullptr: erased
    std::unordered_map<std::string, Node *> extra_;
    size_t live_ = 0;
    size_t base_find(const std::string &id, uint64_t h) const;   // index into base_ or SIZE_MAX
};

struct Tree {                 // mutation_annotated_tree.hpp:113-161
    Node *root = nullptr;
    NodeIndex all_nodes;
    // nodes of a bulk load live in contiguous blocks (constructed and destroyed on the host threads); Node::in_block
    struct Block { Node *p = nullptr; size_t n = 0; std::vector<uint8_t> dead; };
    std::vector<Block> blocks;
    Node *alloc_block(size_t n);          // raw storage for n nodes (not yet constructed)
    void free_node(Node *n);              // delete, or destroy in place when the node lives in a block
    size_t curr_internal_node = 0;
    // name -> condensed leaf ids.  (The reference keeps a tbb::concurrent_unordered_map here, whose
    // iteration order is unspecified; a std::unordered_map fed in the same sequence is used instead.)
    std::unordered_map<std::string, std::vector<std::string>> condensed_nodes;
    std::vector<std::string> condensed_order;   // keys of condensed_nodes in the order they were entered (iteration order below)
    void add_condensed(const std::string &name, const std::vector<std::string> &ids) {
        if (condensed_nodes.emplace(name, ids).second) condensed_order.push_back(name);
    }
    std::unordered_set<std::string> condensed_leaves;
    std::vector<std::string> chroms{""};

    Tree() = default;
    Tree(const Tree &) = delete;
    Tree &operator=(const Tree &) = delete;
    ~Tree();

    std::string new_internal_node_id() { return "node_" + std::to_string(++curr_internal_node); }
    uint32_t chrom_id(const std::string &c);
    Node *get_node(const std::string &id) const;
    size_t num_annotations() const { return root ? root->clade_annotations.size() : 0; }
    Node *create_node(const std::string &id, Node *parent, float branch_length = -1.0f);   // :881-910
    std::vector<Node *> bfs() const;   // :1225-1251
    std::vector<Node *> dfs(Node *from = nullptr) const;   // :1253-1273
    std::vector<Node *> rsearch(Node *n, bool include_self) const;   // :931-948
    std::string clade_assignment(Node *n, size_t clade, bool include_self) const;   // :950-958
    size_t parsimony_score() const;   // :1275-1285
    // move `src` (a child elsewhere) below `dst` as its last child; plain case of :1135-1158
    void reattach(Node *src, Node *dst);
    void remove_leaf(Node *n);   // the part of remove_node() condense_leaves needs (:960-1049, move_level=false)
    void remove_node(Node *n, bool move_level);                // :960-1049 (the node and its descendants)
    void move_node(Node *src, Node *dst, bool move_level = true);   // :1135-1223, incl. the "same mutations" merges
    void collapse_tree();                                      // :1384-1424
    std::vector<Node *> leaves(Node *from = nullptr) const;    // get_leaves, :818-840 (breadth-first)
    bool is_ancestor(const Node *anc, const Node *n) const;    // :920-929
    size_t num_leaves(Node *n) const;                          // get_num_leaves, :866-879
    void rotate_for_display();                                 // :1426-1453 (children by descending subtree size)
    void condense_leaves();      // :1287-1332
    void uncondense_leaves();    // :1334-1382
    void fix_levels(Node *from);
};

// get_tree_copy(), :1493-1549: a newick round trip (internal nodes are re-numbered node_1.. in '(' order, :484),
// then clade annotations, mutations and the condensed-node table are copied node by node in depth-first order.
bool copy_tree(const Tree &src, Tree &dst, std::string &err);

// get_subtree(), :1575-1681: the subtree induced by `samples` (their leaves, every pairwise most recent common
// ancestor, mutations merged along the collapsed paths); internal node names are kept.
bool get_subtree(const Tree &src, const std::vector<std::string> &samples, Tree &dst, std::string &err);

// newick --------------------------------------------------------------------
bool tree_from_newick(const std::string &nwk, Tree &out, std::string &err);   // :415-508
// the same result for large inputs: items tokenised on the host threads, nodes constructed in one block, the name
// index built by a parallel sort; `order` (optional) receives the nodes in creation order = depth-first preorder
bool tree_from_newick_bulk(const char *nwk, size_t len, Tree &out, std::string &err, std::vector<Node *> *order = nullptr);
std::string newick(const Tree &t, Node *from, bool internal_ids, bool branch_len, bool uncondense = false);   // :215-346

// parsimony.proto -------------------------------------------------------------
bool load_mat(const std::string &path, Tree &out, std::string &err);   // :522-612 (.gz via zlib)
bool save_mat(Tree &t, const std::string &path, std::string &err);     // :614-681

// VCF ---------------------------------------------------------------------------
struct MissingSample {           // usher_graph.hpp:33-53
    std::string name;
    std::vector<Mutation> mutations;
    size_t num_ambiguous = 0;
    std::vector<std::string> best_clade_assignment;
    std::vector<std::vector<std::string>> clade_assignments;
};
// existing-MAT branch of read_vcf, :2180-2277
bool read_vcf_missing(Tree &t, const std::string &path, std::vector<MissingSample> &out, std::string &err);
// new-MAT branch, :2052-2179.  The Fitch-Sankoff assignment of every site (mapper_body,
// usher_mapper.cpp:6-161) is delegated to `assign` (the GPU library's ugp_fitch_sankoff); the
// resulting mutations are added in file order of the sites, nodes in breadth-first order (deterministic).
struct SiteBatch {               // what ugp_sites carries, owned
    std::vector<uint32_t> parent;            // breadth-first parent indices, root = UINT32_MAX
    std::vector<uint8_t> ref;
    std::vector<uint64_t> var_off{0};
    std::vector<uint32_t> var_node;
    std::vector<uint8_t> var_nuc;
};
struct SiteMutations { std::vector<uint32_t> site, node; std::vector<uint8_t> par_nuc, mut_nuc; };
typedef bool (*AssignFn)(void *ctx, const SiteBatch &in, SiteMutations &out, std::string &err);
bool read_vcf_build(Tree &t, const std::string &path, std::vector<MissingSample> &out, std::string &err, AssignFn assign, void *ctx);

// bench / test utilities: a tree given as breadth-first arrays written as parsimony.proto (leaves "L<j>"), a CSR query batch as VCF
bool write_pb_from_arrays(uint64_t n, const uint32_t *parent, const uint64_t *mut_off, const int32_t *pos, const uint8_t *ref, const uint8_t *par,
                          const uint8_t *nuc, const std::string &path, std::string &err);
bool write_vcf_from_csr(uint64_t n_samples, const uint64_t *ent_off, const int32_t *pos, const uint8_t *ref, const uint8_t *nuc, const uint8_t *is_missing,
                        const std::string &prefix, const std::string &path, std::string &err);

}  // namespace uh
