"""Small seeded random trees / queries for parity tests (test helper).

Trees are emitted directly as the BFS-order flat arrays the oracle and the
C-ABI take.  The generator deliberately produces the awkward cases: internal
nodes and leaves without mutations, root mutations, back-mutations and repeated
positions along a path, masked mutations (position < 0), queries with N cells,
IUPAC codes (with and without the reference base), rows at positions the tree
never mutates and rows equal to the reference base.
"""
from __future__ import annotations

import numpy as np

ONEHOT = [1, 2, 4, 8]


def random_tree(rng: np.random.Generator, n_leaves: int, genome_len: int = 2000, n_sites: int = 120,
                mut_counts=(0, 0, 1, 1, 1, 2, 3), p_masked: float = 0.0, root_muts: int = 0):
    ref = rng.choice(ONEHOT, size=genome_len + 1)
    sites = np.sort(rng.choice(np.arange(1, genome_len + 1), size=min(n_sites, genome_len), replace=False))
    # random attachment: a leaf gets 2-3 children, an internal node one more
    children = {0: []}
    nid = 1
    n_leaf = 1
    while n_leaf < n_leaves:
        v = int(rng.integers(0, nid))
        k = int(rng.choice([2, 2, 3])) if not children[v] else 1
        if not children[v]:
            n_leaf -= 1
        for _ in range(k):
            children[v].append(nid)
            children[nid] = []
            nid += 1
            n_leaf += 1
    # BFS renumbering
    order = [0]
    head = 0
    while head < len(order):
        order.extend(children[order[head]])
        head += 1
    new = {old: j for j, old in enumerate(order)}
    n = len(order)
    parent = np.full(n, -1, dtype=np.int64)
    for old in order:
        for c in children[old]:
            parent[new[c]] = new[old]
    # mutations by walking parents-first with the true state
    state = [None] * n
    mut_off = np.zeros(n + 1, dtype=np.int64)
    pos, rf, par, nuc = [], [], [], []
    for j in range(n):
        st = {} if parent[j] < 0 else dict(state[parent[j]])
        k = root_muts if parent[j] < 0 else int(rng.choice(mut_counts))
        chosen = np.sort(rng.choice(sites, size=min(k, len(sites)), replace=False)) if k else []
        muts = []
        if parent[j] >= 0 and rng.random() < p_masked:
            muts.append((-1, 0, 0, 0))
        for p in chosen:
            p = int(p)
            cur = st.get(p, int(ref[p]))
            new_allele = int(rng.choice([a for a in ONEHOT if a != cur]))
            muts.append((p, int(ref[p]), cur, new_allele))
            st[p] = new_allele
        state[j] = st
        for (p, r, pa, m) in muts:
            pos.append(p); rf.append(r); par.append(pa); nuc.append(m)
        mut_off[j + 1] = len(pos)
    arrays = {
        "n": n, "parent": parent, "mut_off": mut_off,
        "mut_pos": np.asarray(pos, dtype=np.int32), "mut_ref": np.asarray(rf, dtype=np.int8),
        "mut_par": np.asarray(par, dtype=np.int8), "mut_nuc": np.asarray(nuc, dtype=np.int8),
        "names": ["n%d" % j for j in range(n)],
    }
    return arrays, ref, sites, state


def random_query(rng: np.random.Generator, arrays, ref, sites, state, genome_len: int,
                 n_subst=(0, 1, 2, 3), n_ambig=(0, 0, 2, 5, 30), name="Q"):
    n = arrays["n"]
    src = int(rng.integers(0, n))
    geno = {p: a for p, a in state[src].items() if a != int(ref[p])}
    for _ in range(int(rng.choice(n_subst))):
        p = int(rng.integers(1, genome_len + 1)) if rng.random() < 0.3 else int(rng.choice(sites))
        cur = geno.get(p, int(ref[p]))
        a = int(rng.choice([x for x in ONEHOT if x != cur]))
        if a == int(ref[p]):
            geno.pop(p, None)
        else:
            geno[p] = a
    rows = {p: (a, 0) for p, a in geno.items()}
    for _ in range(int(rng.choice(n_ambig))):
        p = int(rng.choice(sites)) if rng.random() < 0.8 else int(rng.integers(1, genome_len + 1))
        kind = rng.random()
        if kind < 0.4:
            rows[p] = (15, 1)                       # N / '.'
        elif kind < 0.95:
            mask = int(rng.integers(1, 15))
            if mask & (mask - 1) == 0:
                mask |= int(rng.choice(ONEHOT))
            if mask == 15:
                rows[p] = (15, 1)
            else:
                rows[p] = (mask, 0)
        else:
            rows[p] = (int(ref[p]), 0)              # explicit row equal to the reference base
    ps = sorted(rows)
    return {
        "name": name,
        "pos": np.asarray(ps, dtype=np.int32),
        "ref": np.asarray([int(ref[p]) for p in ps], dtype=np.int8),
        "nuc": np.asarray([rows[p][0] for p in ps], dtype=np.int8),
        "is_missing": np.asarray([rows[p][1] for p in ps], dtype=np.int8),
    }


def make_case(seed: int, n_leaves: int, n_queries: int, genome_len: int = 2000, n_sites: int = 120, **kw):
    rng = np.random.default_rng(seed)
    qkw = {k: kw.pop(k) for k in ("n_subst", "n_ambig") if k in kw}
    arrays, ref, sites, state = random_tree(rng, n_leaves, genome_len, n_sites, **kw)
    queries = [random_query(rng, arrays, ref, sites, state, genome_len, name="Q%d" % i, **qkw) for i in range(n_queries)]
    return arrays, queries


def relabel_preorder(arrays):
    """The same tree with its nodes renumbered in depth-first preorder (children in their stored order): parents still
    precede their children -- all the C ABI asks of parent[] -- but the numbering is no breadth-first expansion, so the
    children of a node are not contiguous.  Returns (arrays, new_index_of_old)."""
    par = np.asarray(arrays["parent"]).astype(np.int64)
    n = len(par)
    kids = [[] for _ in range(n)]
    for j in range(1, n):
        kids[par[j]].append(j)
    stack, order = [0], []
    while stack:
        v = stack.pop()
        order.append(v)
        stack.extend(reversed(kids[v]))
    new = np.empty(n, np.int64)
    new[order] = np.arange(n)
    off = np.asarray(arrays["mut_off"]).astype(np.int64)
    cnt = np.diff(off)[order]
    noff = np.concatenate([[0], np.cumsum(cnt)])
    take = np.concatenate([np.arange(off[v], off[v + 1]) for v in order]) if off[-1] else np.zeros(0, np.int64)
    out = dict(arrays)
    out["parent"] = np.array([-1 if par[v] < 0 else new[par[v]] for v in order], dtype=np.asarray(arrays["parent"]).dtype)
    out["mut_off"] = noff.astype(np.asarray(arrays["mut_off"]).dtype)
    for k in ("mut_pos", "mut_ref", "mut_par", "mut_nuc"):
        if k in arrays:
            out[k] = np.asarray(arrays[k])[take]
    if "names" in arrays:
        out["names"] = [arrays["names"][v] for v in order]
    return out, new


def caterpillar_case(seed: int, depth: int, muts_per_node: int, n_queries: int, genome_len: int = 29000, n_sites: int = 3000):
    """A maximally deep tree: a chain of `depth` internal nodes, each with one leaf child and the next
    chain node (BFS order: chain node 2k-1, leaf 2k at level k).  Root paths carry depth*muts_per_node
    mutations, so D values, the hsub bound and the 16-bit packed counters are all pushed to their limits."""
    rng = np.random.default_rng(seed)
    ref = rng.choice(ONEHOT, size=genome_len + 1)
    sites = np.sort(rng.choice(np.arange(1, genome_len + 1), size=n_sites, replace=False))
    n = 2 * depth + 1
    parent = np.full(n, -1, dtype=np.int64)
    for k in range(1, depth + 1):
        parent[2 * k - 1] = 0 if k == 1 else 2 * k - 3
        parent[2 * k] = 0 if k == 1 else 2 * k - 3
    chain = {}                       # running genotype of the chain
    chain_muts = []                  # per level: list of (pos, new_allele)
    recs = [[] for _ in range(n)]
    for k in range(1, depth + 1):
        c, l = 2 * k - 1, 2 * k
        # the leaf hangs off chain node k-1: its mutations see the state before level k's chain mutations
        for p in np.sort(rng.choice(sites, size=1, replace=False)):
            p = int(p)
            cur = chain.get(p, int(ref[p]))
            recs[l].append((p, int(ref[p]), cur, int(rng.choice([a for a in ONEHOT if a != cur]))))
        lvl = []
        for p in np.sort(rng.choice(sites, size=muts_per_node, replace=False)):
            p = int(p)
            cur = chain.get(p, int(ref[p]))
            a = int(rng.choice([x for x in ONEHOT if x != cur]))
            recs[c].append((p, int(ref[p]), cur, a))
            chain[p] = a
            lvl.append((p, a))
        chain_muts.append(lvl)
    mut_off = np.zeros(n + 1, dtype=np.int64)
    pos, rf, par, nuc = [], [], [], []
    for j in range(n):
        for (p, r, pa, m) in recs[j]:
            pos.append(p); rf.append(r); par.append(pa); nuc.append(m)
        mut_off[j + 1] = len(pos)
    arrays = {"n": n, "parent": parent, "mut_off": mut_off, "mut_pos": np.asarray(pos, dtype=np.int32),
              "mut_ref": np.asarray(rf, dtype=np.int8), "mut_par": np.asarray(par, dtype=np.int8),
              "mut_nuc": np.asarray(nuc, dtype=np.int8), "names": ["n%d" % j for j in range(n)]}
    queries = []
    for i in range(n_queries):
        k = int(rng.integers(1, depth + 1))
        geno = {}
        for lvl in chain_muts[:k]:
            for p, a in lvl:
                geno[p] = a
        for _ in range(int(rng.integers(0, 3))):
            p = int(rng.choice(sites))
            geno[p] = int(rng.choice([x for x in ONEHOT if x != geno.get(p, int(ref[p]))]))
        rows = {p: (a, 0) for p, a in geno.items() if a != int(ref[p])}
        for _ in range(int(rng.integers(0, 4))):
            rows[int(rng.choice(sites))] = (15, 1)
        ps = sorted(rows)
        queries.append({"name": "Q%d" % i, "pos": np.asarray(ps, dtype=np.int32),
                        "ref": np.asarray([int(ref[p]) for p in ps], dtype=np.int8),
                        "nuc": np.asarray([rows[p][0] for p in ps], dtype=np.int8),
                        "is_missing": np.asarray([rows[p][1] for p in ps], dtype=np.int8)})
    return arrays, queries


def polytomy_case(seed: int, fanouts=(40, 60, 25), n_queries: int = 600, genome_len: int = 5000, n_sites: int = 400):
    """A bushy tree: every internal node of level L has fanouts[L] children (huge polytomies, as in trees of
    densely sampled outbreaks), 0-2 mutations per branch, many leaves identical to their parent."""
    rng = np.random.default_rng(seed)
    ref = rng.choice(ONEHOT, size=genome_len + 1)
    sites = np.sort(rng.choice(np.arange(1, genome_len + 1), size=n_sites, replace=False))
    parent = [-1]
    level = [0]
    frontier = [0]
    for L, k in enumerate(fanouts):   # BFS construction: children of a node are contiguous, parents non-decreasing
        nxt = []
        for p in frontier:
            kk = k if L == 0 else int(rng.integers(1, k + 1))
            for _ in range(kk):
                parent.append(p); level.append(L + 1); nxt.append(len(parent) - 1)
        frontier = nxt
    n = len(parent)
    parent = np.asarray(parent, dtype=np.int64)
    state = [None] * n
    mut_off = np.zeros(n + 1, dtype=np.int64)
    pos, rf, par, nuc = [], [], [], []
    for j in range(n):
        st = {} if parent[j] < 0 else dict(state[parent[j]])
        k = 0 if parent[j] < 0 else int(rng.choice([0, 0, 0, 1, 1, 2]))
        for p in (np.sort(rng.choice(sites, size=k, replace=False)) if k else []):
            p = int(p)
            cur = st.get(p, int(ref[p]))
            a = int(rng.choice([x for x in ONEHOT if x != cur]))
            pos.append(p); rf.append(int(ref[p])); par.append(cur); nuc.append(a)
            st[p] = a
        state[j] = st
        mut_off[j + 1] = len(pos)
    arrays = {"n": n, "parent": parent, "mut_off": mut_off, "mut_pos": np.asarray(pos, dtype=np.int32),
              "mut_ref": np.asarray(rf, dtype=np.int8), "mut_par": np.asarray(par, dtype=np.int8),
              "mut_nuc": np.asarray(nuc, dtype=np.int8), "names": ["n%d" % j for j in range(n)]}
    queries = [random_query(rng, arrays, ref, sites, state, genome_len, name="Q%d" % i, n_ambig=(0, 0, 1, 3)) for i in range(n_queries)]
    return arrays, queries
