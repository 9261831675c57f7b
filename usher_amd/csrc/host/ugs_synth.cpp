// ugs_synth.cpp -- seeded synthetic MAT + query generator (bench / test input
// only; not on the product path).  Follows the recipe of SURVEY.md 8(d):
// genome L (29,903 for SARS-CoV-2), random-attachment tree up to a target node
// count (a leaf gets 2-3 children, an internal node one more), per-branch
// mutation count drawn from {0,0,1,1,1,2,3} over V variable sites so that
// homoplasy and back-mutation occur, alleles uniform over the three non-current
// bases.  Queries copy the genotype of a uniformly random node, add 0..3
// substitutions and optionally N runs and IUPAC cells (BASELINE config 5).
// Output is in the reference's BFS node order, the layout ugp_tree_desc takes.
//
// shape 1 ("sars2") imitates the public SARS-CoV-2 tree instead (BASELINE configs 3-5; the real
// public-latest.all.masked.pb.gz is not in the image): three new nodes in four attach as a SIBLING of a
// random existing node (7 times in 10 one of the newest fifth: the epidemic grows from its recent lineages) --
// preferential attachment by child count, which grows polytomies of hundreds to thousands of children -- and
// one in four below it; the tree stays shallow (depth in the tens, a ladder-like backbone); internal branches carry about one mutation, more than half of the leaves are
// identical to their parent (at most one such leaf per parent: identical samples are condensed), and the
// mutated site is drawn with a quadratic skew so that a few sites are hit thousands of times (homoplasy).
// "recent" queries copy one of the last 10 % of the nodes created and add a few private mutations.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <new>
#include <vector>

namespace {

struct Rng {   // splitmix64 / xoshiro256**
    uint64_t s[4];
    static uint64_t sm(uint64_t &x) {
        uint64_t z = (x += 0x9e3779b97f4a7c15ull);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
        return z ^ (z >> 31);
    }
    explicit Rng(uint64_t seed) { for (auto &v : s) v = sm(seed); }
    static uint64_t rotl(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
    uint64_t next() {
        uint64_t r = rotl(s[1] * 5, 7) * 9, t = s[1] << 17;
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]; s[2] ^= t; s[3] = rotl(s[3], 45);
        return r;
    }
    uint64_t below(uint64_t n) { return (uint64_t)(((__uint128_t)next() * n) >> 64); }
};

const uint8_t kOneHot[4] = {1, 2, 4, 8};

}  // namespace

struct ugs_tree {
    uint32_t genome_len = 0;
    std::vector<uint32_t> parent;     // BFS order, parent[0] = UINT32_MAX
    std::vector<uint64_t> mut_off;
    std::vector<int32_t> mut_pos;
    std::vector<uint8_t> mut_ref, mut_par, mut_nuc;
    std::vector<uint8_t> ref;         // [genome_len+1] allele index
    std::vector<uint32_t> sites;      // sorted variable positions
    std::vector<uint32_t> recent;     // BFS ids of the last 10 % of the nodes created (shape 1)
};

struct ugs_queries {
    std::vector<uint64_t> ent_off;
    std::vector<int32_t> pos;
    std::vector<uint8_t> ref, nuc, is_missing;
    std::vector<uint32_t> source_node;
};

extern "C" {

ugs_tree *ugs_tree_create2(uint64_t target_nodes, uint32_t genome_len, uint32_t n_sites, uint64_t seed, uint32_t shape);
ugs_tree *ugs_tree_create(uint64_t target_nodes, uint32_t genome_len, uint32_t n_sites, uint64_t seed) {
    return ugs_tree_create2(target_nodes, genome_len, n_sites, seed, 0);
}

ugs_tree *ugs_tree_create2(uint64_t target_nodes, uint32_t genome_len, uint32_t n_sites, uint64_t seed, uint32_t shape) {
    if (target_nodes < 1 || genome_len < 1 || n_sites < 1 || n_sites > genome_len || target_nodes >= (1ull << 31)) return nullptr;
    ugs_tree *t = new (std::nothrow) ugs_tree();
    if (!t) return nullptr;
    Rng rng(seed);
    t->genome_len = genome_len;
    t->ref.resize(genome_len + 1);
    for (auto &r : t->ref) r = (uint8_t)rng.below(4);
    {   // V distinct variable positions
        std::vector<uint32_t> all(genome_len);
        for (uint32_t i = 0; i < genome_len; i++) all[i] = i + 1;
        for (uint32_t i = 0; i < n_sites; i++) std::swap(all[i], all[i + rng.below(genome_len - i)]);
        t->sites.assign(all.begin(), all.begin() + n_sites);
        std::sort(t->sites.begin(), t->sites.end());
    }
    // ---- topology (creation order), then BFS renumbering
    std::vector<uint32_t> par;
    std::vector<uint32_t> nchild;
    par.reserve(target_nodes + 4); nchild.reserve(target_nodes + 4);
    par.push_back(UINT32_MAX); nchild.push_back(0);
    if (shape == 0) {
        while (par.size() < target_nodes) {
            uint32_t v = (uint32_t)rng.below(par.size());
            uint32_t k = nchild[v] ? 1u : (rng.below(3) < 2 ? 2u : 3u);
            for (uint32_t i = 0; i < k; i++) { par.push_back(v); nchild.push_back(0); nchild[v]++; }
        }
    } else {
        while (par.size() < target_nodes) {
            // the epidemic grows from its recent lineages: 7 times in 10 the anchor is one of the newest 20 % of the nodes
            const uint64_t sz = par.size(), win = std::max<uint64_t>(1, sz / 5);
            uint32_t v = rng.below(10) < 7 ? (uint32_t)(sz - 1 - rng.below(win)) : (uint32_t)rng.below(sz);
            uint32_t p = (v != 0 && rng.below(4) != 0) ? par[v] : v;   // sibling of v (3 in 4) or child of v
            // a leaf that gets its first child would become a one-child internal node: give it two
            const uint32_t k = nchild[p] ? 1u : 2u;
            for (uint32_t i = 0; i < k && par.size() < target_nodes + 1; i++) { par.push_back(p); nchild.push_back(0); nchild[p]++; }
        }
    }
    const uint64_t N = par.size();
    std::vector<uint32_t> coff(N + 1, 0), kids(N > 1 ? N - 1 : 0);
    for (uint64_t j = 1; j < N; j++) coff[par[j] + 1]++;
    for (uint64_t j = 0; j < N; j++) coff[j + 1] += coff[j];
    {
        std::vector<uint32_t> fill(coff.begin(), coff.end() - 1);
        for (uint64_t j = 1; j < N; j++) kids[fill[par[j]]++] = (uint32_t)j;
    }
    std::vector<uint32_t> order; order.reserve(N);
    order.push_back(0);
    for (uint64_t h = 0; h < order.size(); h++)
        for (uint32_t c = coff[order[h]]; c < coff[order[h] + 1]; c++) order.push_back(kids[c]);
    std::vector<uint32_t> newid(N);
    for (uint64_t j = 0; j < N; j++) newid[order[j]] = (uint32_t)j;
    t->parent.resize(N);
    for (uint64_t j = 0; j < N; j++) t->parent[newid[j]] = (par[j] == UINT32_MAX) ? UINT32_MAX : newid[par[j]];
    if (shape != 0) for (uint64_t j = N - std::max<uint64_t>(1, N / 10); j < N; j++) t->recent.push_back(newid[j]);
    // children CSR in BFS ids (contiguous by construction)
    std::vector<uint32_t> first(N + 1, 0);
    for (uint64_t j = 1; j < N; j++) first[t->parent[j] + 1]++;
    for (uint64_t j = 0; j < N; j++) first[j + 1] += first[j];   // children of j are ids [1+first[j], 1+first[j+1])
    // ---- mutations: DFS with a running state and an undo log
    struct Rec { uint32_t node; uint32_t site_idx; uint8_t prev, mut; };
    std::vector<Rec> recs; recs.reserve((size_t)(N * 1.2));
    std::vector<uint8_t> cur(n_sites);
    for (uint32_t s = 0; s < n_sites; s++) cur[s] = t->ref[t->sites[s]];
    struct Undo { uint32_t site_idx; uint8_t old; };
    std::vector<Undo> undo;
    struct Frame { uint32_t node, next, mark; };
    std::vector<Frame> st;
    const uint32_t counts[7] = {0, 0, 1, 1, 1, 2, 3};
    std::vector<uint8_t> has_twin;   // shape 1: the node already has a leaf child identical to it
    if (shape != 0) has_twin.assign(N, 0);
    auto mutate = [&](uint32_t node) {
        if (node == 0) return;
        uint32_t k = counts[rng.below(7)];
        if (shape != 0) {
            const bool leaf = first[node + 1] == first[node];
            const uint64_t u = rng.below(100);
            if (leaf) {
                k = u < 58 ? 0u : (u < 88 ? 1u : (u < 97 ? 2u : 3u));
                if (k == 0) { if (has_twin[t->parent[node]]) k = 1; else has_twin[t->parent[node]] = 1; }
            } else k = u < 6 ? 0u : (u < 76 ? 1u : (u < 94 ? 2u : 3u));
        }
        uint32_t chosen[3];
        for (uint32_t i = 0; i < k; i++) {
            for (;;) {
                uint32_t s = (uint32_t)rng.below(n_sites);
                if (shape != 0) {   // quadratic skew over a fixed pseudo-random order of the sites
                    const uint64_t a = rng.below(1u << 20), b = rng.below(1u << 20);
                    const uint64_t r = (a * b * (uint64_t)n_sites) >> 40;
                    s = (uint32_t)((r * 2654435761ull) % n_sites);
                }
                bool dup = false;
                for (uint32_t q = 0; q < i; q++) dup |= (chosen[q] == s);
                if (!dup) { chosen[i] = s; break; }
            }
        }
        std::sort(chosen, chosen + k);
        for (uint32_t i = 0; i < k; i++) {
            uint32_t s = chosen[i];
            uint8_t prev = cur[s];
            uint8_t m = (uint8_t)((prev + 1 + rng.below(3)) & 3);
            recs.push_back({node, s, prev, m});
            undo.push_back({s, prev});
            cur[s] = m;
        }
    };
    st.push_back({0, 0, 0});
    while (!st.empty()) {
        Frame &f = st.back();
        uint32_t b = 1 + first[f.node], e = 1 + first[f.node + 1];
        if (b + f.next < e) {
            uint32_t c = b + f.next++;
            uint32_t mark = (uint32_t)undo.size();
            mutate(c);
            st.push_back({c, 0, mark});
        } else {
            while (undo.size() > f.mark) { cur[undo.back().site_idx] = undo.back().old; undo.pop_back(); }
            st.pop_back();
        }
    }
    // counting sort of the records into BFS CSR (records of one node are already position-sorted)
    const uint64_t M = recs.size();
    t->mut_off.assign(N + 1, 0);
    for (const Rec &r : recs) t->mut_off[r.node + 1]++;
    for (uint64_t j = 0; j < N; j++) t->mut_off[j + 1] += t->mut_off[j];
    t->mut_pos.resize(M); t->mut_ref.resize(M); t->mut_par.resize(M); t->mut_nuc.resize(M);
    {
        std::vector<uint64_t> fill(t->mut_off.begin(), t->mut_off.end() - 1);
        for (const Rec &r : recs) {
            uint64_t i = fill[r.node]++;
            uint32_t p = t->sites[r.site_idx];
            t->mut_pos[i] = (int32_t)p;
            t->mut_ref[i] = kOneHot[t->ref[p]];
            t->mut_par[i] = kOneHot[r.prev];
            t->mut_nuc[i] = kOneHot[r.mut];
        }
    }
    return t;
}

void ugs_tree_destroy(ugs_tree *t) { delete t; }
uint64_t ugs_tree_nodes(const ugs_tree *t) { return t->parent.size(); }
uint64_t ugs_tree_muts(const ugs_tree *t) { return t->mut_pos.size(); }
const uint32_t *ugs_tree_parent(const ugs_tree *t) { return t->parent.data(); }
const uint64_t *ugs_tree_mut_off(const ugs_tree *t) { return t->mut_off.data(); }
const int32_t *ugs_tree_mut_pos(const ugs_tree *t) { return t->mut_pos.data(); }
const uint8_t *ugs_tree_mut_ref(const ugs_tree *t) { return t->mut_ref.data(); }
const uint8_t *ugs_tree_mut_par(const ugs_tree *t) { return t->mut_par.data(); }
const uint8_t *ugs_tree_mut_nuc(const ugs_tree *t) { return t->mut_nuc.data(); }

// n_lo..n_hi: number of N cells per query (as 1..8 contiguous runs); iupac_hi: 0..iupac_hi ambiguity cells.
ugs_queries *ugs_queries_create2(const ugs_tree *t, uint64_t n_queries, uint64_t seed, uint32_t max_subst,
                                 uint32_t n_lo, uint32_t n_hi, uint32_t iupac_hi, uint32_t recent);
ugs_queries *ugs_queries_create(const ugs_tree *t, uint64_t n_queries, uint64_t seed, uint32_t max_subst,
                                uint32_t n_lo, uint32_t n_hi, uint32_t iupac_hi) {
    return ugs_queries_create2(t, n_queries, seed, max_subst, n_lo, n_hi, iupac_hi, 0);
}

// Adopt existing tree arrays (another process generated them with the same seed / genome_len / n_sites): only
// the reference genome and the site list, the first draws of the generator, are regenerated.
ugs_tree *ugs_tree_adopt(uint64_t n_nodes, const uint32_t *parent, const uint64_t *mut_off, const int32_t *mut_pos, const uint8_t *mut_ref,
                         const uint8_t *mut_par, const uint8_t *mut_nuc, uint32_t genome_len, uint32_t n_sites, uint64_t seed, uint32_t shape) {
    ugs_tree *t = new (std::nothrow) ugs_tree();
    if (!t) return nullptr;
    Rng rng(seed);
    t->genome_len = genome_len;
    t->ref.resize(genome_len + 1);
    for (auto &r : t->ref) r = (uint8_t)rng.below(4);
    std::vector<uint32_t> all(genome_len);
    for (uint32_t i = 0; i < genome_len; i++) all[i] = i + 1;
    for (uint32_t i = 0; i < n_sites; i++) std::swap(all[i], all[i + rng.below(genome_len - i)]);
    t->sites.assign(all.begin(), all.begin() + n_sites);
    std::sort(t->sites.begin(), t->sites.end());
    const uint64_t M = mut_off[n_nodes];
    t->parent.assign(parent, parent + n_nodes);
    t->mut_off.assign(mut_off, mut_off + n_nodes + 1);
    t->mut_pos.assign(mut_pos, mut_pos + M); t->mut_ref.assign(mut_ref, mut_ref + M);
    t->mut_par.assign(mut_par, mut_par + M); t->mut_nuc.assign(mut_nuc, mut_nuc + M);
    if (shape != 0) {   // "recent" is not recoverable from the arrays: the deepest tenth of the BFS order stands in
        for (uint64_t j = n_nodes - std::max<uint64_t>(1, n_nodes / 10); j < n_nodes; j++) t->recent.push_back((uint32_t)j);
    }
    return t;
}

ugs_queries *ugs_queries_create2(const ugs_tree *t, uint64_t n_queries, uint64_t seed, uint32_t max_subst,
                                 uint32_t n_lo, uint32_t n_hi, uint32_t iupac_hi, uint32_t recent) {
    ugs_queries *q = new (std::nothrow) ugs_queries();
    if (!q) return nullptr;
    Rng rng(seed ^ 0x5eed5eed5eedull);
    const uint64_t N = t->parent.size();
    const uint32_t L = t->genome_len;
    q->ent_off.push_back(0);
    struct Row { int32_t pos; uint8_t nuc, missing; };
    std::vector<Row> rows;
    std::vector<int32_t> row_of_pos((size_t)L + 2, -1);
    for (uint64_t i = 0; i < n_queries; i++) {
        for (const Row &r : rows) row_of_pos[r.pos] = -1;
        rows.clear();
        uint32_t node = (uint32_t)rng.below(N);
        if ((recent & 1u) && !t->recent.empty()) node = t->recent[rng.below(t->recent.size())];
        q->source_node.push_back(node);
        // genotype of `node`: most recent mutation per position on the root path
        for (uint32_t v = node; v != UINT32_MAX; v = t->parent[v]) {
            for (uint64_t m = t->mut_off[v]; m < t->mut_off[v + 1]; m++) {
                int32_t p = t->mut_pos[m];
                if (row_of_pos[p] < 0) { row_of_pos[p] = (int32_t)rows.size(); rows.push_back({p, t->mut_nuc[m], 0}); }
            }
        }
        auto set_row = [&](int32_t p, uint8_t nuc, uint8_t missing) {
            if (row_of_pos[p] >= 0) { rows[row_of_pos[p]].nuc = nuc; rows[row_of_pos[p]].missing = missing; return; }
            row_of_pos[p] = (int32_t)rows.size();
            rows.push_back({p, nuc, missing});
        };
        // `recent` carries more switches (round 6, "far" queries: samples that are NOT near any node of the tree): bits 31:8 = the least
        // number of substitutions (the count is then uniform in [min, max_subst]); bit 2 = every 8th sample is the all-reference
        // sample -- no rows at all
        const uint32_t min_subst = std::min(recent >> 8, max_subst);
        uint32_t ns = max_subst ? min_subst + (uint32_t)rng.below(max_subst - min_subst + 1) : 0;
        if ((recent & 4u) && (i & 7u) == 7u) { rows.clear(); ns = 0; for (uint32_t v = node; v != UINT32_MAX; v = t->parent[v]) for (uint64_t m = t->mut_off[v]; m < t->mut_off[v + 1]; m++) row_of_pos[t->mut_pos[m]] = -1; }
        for (uint32_t k = 0; k < ns; k++) {
            int32_t p = (rng.below(10) < 3) ? (int32_t)(1 + rng.below(L)) : (int32_t)t->sites[rng.below(t->sites.size())];
            uint8_t curr = kOneHot[t->ref[p]];
            if (row_of_pos[p] >= 0) curr = rows[row_of_pos[p]].nuc;
            uint8_t idx = 0; while (kOneHot[idx] != curr && idx < 3) idx++;
            set_row(p, kOneHot[(idx + 1 + rng.below(3)) & 3], 0);
        }
        uint32_t nn = n_hi > n_lo ? n_lo + (uint32_t)rng.below(n_hi - n_lo + 1) : n_lo;
        if (nn) {
            uint32_t runs = 1 + (uint32_t)rng.below(8);
            for (uint32_t r = 0; r < runs; r++) {
                uint32_t len = nn / runs + (r < nn % runs ? 1 : 0);
                if (!len) continue;
                uint32_t start = 1 + (uint32_t)rng.below(L - std::min(len, L - 1));
                for (uint32_t d = 0; d < len && start + d <= L; d++) set_row((int32_t)(start + d), 15, 1);
            }
        }
        uint32_t ni = iupac_hi ? (uint32_t)rng.below(iupac_hi + 1) : 0;
        for (uint32_t k = 0; k < ni; k++) {
            int32_t p = (int32_t)t->sites[rng.below(t->sites.size())];
            uint8_t mask = (uint8_t)(1 + rng.below(14));
            if ((mask & (mask - 1)) == 0) mask |= kOneHot[rng.below(4)];
            if (recent & 2u) {   // an ambiguity code that holds the sample's own base, as a real mixed call does (default: any set)
                uint8_t curr = kOneHot[t->ref[p]];
                if (row_of_pos[p] >= 0) { if (rows[row_of_pos[p]].missing) continue; curr = rows[row_of_pos[p]].nuc; }
                mask |= curr;
            }
            if (mask == 15) set_row(p, 15, 1); else set_row(p, mask, 0);
        }
        // drop rows equal to the reference base (a VCF would not carry them), sort
        std::vector<Row> sorted_rows(rows);
        std::sort(sorted_rows.begin(), sorted_rows.end(), [](const Row &a, const Row &b) { return a.pos < b.pos; });
        for (const Row &r : sorted_rows) {
            uint8_t refhot = kOneHot[t->ref[r.pos]];
            if (!r.missing && r.nuc == refhot) continue;
            q->pos.push_back(r.pos); q->ref.push_back(refhot); q->nuc.push_back(r.nuc); q->is_missing.push_back(r.missing);
        }
        q->ent_off.push_back(q->pos.size());
    }
    return q;
}

void ugs_queries_destroy(ugs_queries *q) { delete q; }
uint64_t ugs_queries_count(const ugs_queries *q) { return q->ent_off.size() - 1; }
uint64_t ugs_queries_entries(const ugs_queries *q) { return q->pos.size(); }
const uint64_t *ugs_queries_ent_off(const ugs_queries *q) { return q->ent_off.data(); }
const int32_t *ugs_queries_pos(const ugs_queries *q) { return q->pos.data(); }
const uint8_t *ugs_queries_ref(const ugs_queries *q) { return q->ref.data(); }
const uint8_t *ugs_queries_nuc(const ugs_queries *q) { return q->nuc.data(); }
const uint8_t *ugs_queries_missing(const ugs_queries *q) { return q->is_missing.data(); }
const uint32_t *ugs_queries_source(const ugs_queries *q) { return q->source_node.data(); }

}  // extern "C"
