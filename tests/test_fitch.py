"""Fitch-Sankoff site assignment (MAT construction): the set formulation the HIP kernels
use, and the kernels themselves through the C ABI, against the literal oracle restatement
of mapper_body::operator() (usher_mapper.cpp:6-161)."""
import numpy as np
import pytest

from oracle import capi


def random_bfs_tree(rng, n, kind):
    """BFS-ordered parent array (parent[0] = -1, non-decreasing)."""
    if kind == "chain":
        par = np.arange(-1, n - 1)
    elif kind == "star":
        par = np.concatenate([[-1], np.zeros(n - 1, np.int64)])
    else:
        # random attachment, then relabel in BFS order
        p0 = np.concatenate([[-1], [rng.integers(0, max(1, j if kind == "random" else min(j, 3))) for j in range(1, n)]]).astype(np.int64)
        children = [[] for _ in range(n)]
        for j in range(1, n):
            children[p0[j]].append(j)
        order, q = [], [0]
        while q:
            nxt = []
            for u in q:
                order.append(u)
                nxt.extend(children[u])
            q = nxt
        new = np.zeros(n, np.int64)
        new[order] = np.arange(n)
        par = np.full(n, -1, np.int64)
        for j in range(1, n):
            par[new[j]] = new[p0[j]]
    return par.astype(np.int64)


def random_sites(rng, parent, n_sites, p_var=0.3, p_internal=0.05, p_dense=0.9):
    n = len(parent)
    is_leaf = np.ones(n, bool)
    is_leaf[parent[1:]] = False
    leaves, internal = np.flatnonzero(is_leaf), np.flatnonzero(~is_leaf)
    ref = (1 << rng.integers(0, 4, n_sites)).astype(np.uint8)
    off, nodes, nucs = [0], [], []
    for s in range(n_sites):
        pick = leaves[rng.random(len(leaves)) < (p_var if s % 5 else p_dense)]
        extra = internal[rng.random(len(internal)) < p_internal] if s % 3 == 0 else np.zeros(0, np.int64)
        for nd in np.concatenate([pick, extra]):
            r = rng.random()
            if r < 0.7:
                nuc = 1 << rng.integers(0, 4)
            elif r < 0.85:
                nuc = int(rng.integers(1, 16))
            else:
                nuc = 15
            nodes.append(nd)
            nucs.append(nuc)
        off.append(len(nodes))
    return ref, np.array(off, np.uint64), np.array(nodes, np.uint32), np.array(nucs, np.uint8)


def oracle_mutations(parent, ref, off, nodes, nucs):
    out = []
    for s in range(len(ref)):
        a, b = int(off[s]), int(off[s + 1])
        state, mpar, mnuc = capi.fitch_site(parent, int(ref[s]), nodes[a:b].astype(np.int64), nucs[a:b].astype(np.int8))
        for j in np.flatnonzero(mnuc):
            out.append((s, int(j), int(mpar[j]), int(mnuc[j])))
    return out


def set_model(parent, ref, off, nodes, nucs):
    """The formulation of usher_amd/csrc/ugp_fitch.hip in plain python: Fitch sets + child counts."""
    n = len(parent)
    nch = np.zeros(n, np.int64)
    for j in range(1, n):
        nch[parent[j]] += 1
    out = []
    for s in range(len(ref)):
        F = np.where(nch > 0, 15, int(ref[s])).astype(np.int64)
        for v in range(int(off[s]), int(off[s + 1])):
            F[nodes[v]] = nucs[v]
        cnt = np.zeros((n, 4), np.int64)
        for j in range(n - 1, -1, -1):
            if nch[j]:
                allowed = [b for b in range(4) if F[j] >> b & 1]
                m = min(cnt[j, b] for b in allowed)
                F[j] = sum(1 << b for b in allowed if cnt[j, b] == m)
            if j:
                for b in range(4):
                    cnt[parent[j], b] += 0 if F[j] >> b & 1 else 1
        st = np.zeros(n, np.int64)
        for j in range(n):
            sp = int(ref[s]) if j == 0 else st[parent[j]]
            st[j] = sp if F[j] & sp else F[j] & -F[j]
            if st[j] != sp:
                out.append((s, j, sp, int(st[j])))
    return out


CASES = [("random", 300, 40, 1), ("bushy", 500, 24, 2), ("chain", 40, 16, 3), ("star", 70, 16, 4), ("random", 2, 9, 5), ("random", 1, 5, 6)]


@pytest.mark.parametrize("kind,n,n_sites,seed", CASES)
def test_set_formulation_equals_oracle(kind, n, n_sites, seed):
    rng = np.random.default_rng(seed)
    parent = random_bfs_tree(rng, n, kind)
    ref, off, nodes, nucs = random_sites(rng, parent, n_sites)
    assert set_model(parent, ref, off, nodes, nucs) == oracle_mutations(parent, ref, off, nodes, nucs)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,n_sites,seed", CASES + [("random", 3000, 700, 7), ("bushy", 5000, 1100, 8), ("star", 1500, 40, 9)])
def test_gpu_fitch_sankoff_equals_oracle(kind, n, n_sites, seed, monkeypatch):
    from usher_amd.fitch import fitch_sankoff
    rng = np.random.default_rng(seed)
    parent = random_bfs_tree(rng, n, kind)
    ref, off, nodes, nucs = random_sites(rng, parent, n_sites, p_var=0.1 if n > 1000 else 0.3)
    want = oracle_mutations(parent, ref, off, nodes, nucs)
    # one pass; several passes of 24 sites; and the exact listing pass used when the first buffer is too small
    for budget, cap in ((None, None), (str(n * 4 * 3), None), (None, "3")):
        monkeypatch.delenv("UGP_FITCH_BYTES", raising=False)
        monkeypatch.delenv("UGP_FITCH_EMIT_CAP", raising=False)
        if budget:
            monkeypatch.setenv("UGP_FITCH_BYTES", budget)
        if cap:
            monkeypatch.setenv("UGP_FITCH_EMIT_CAP", cap)
        site, node, mpar, mnuc = fitch_sankoff(parent, ref, off, nodes, nucs)
        got = list(zip(site.tolist(), node.tolist(), mpar.tolist(), mnuc.tolist()))
        assert got == want


@pytest.mark.gpu
@pytest.mark.parametrize("kind,n,n_sites,p_var,p_internal,seed", [("random", 3000, 1100, 0.001, 0.0, 21), ("bushy", 4000, 1600, 0.0006, 0.002, 22),
                                                                  ("star", 2000, 600, 0.002, 0.0, 23), ("chain", 300, 1030, 0.002, 0.01, 24),
                                                                  ("random", 2500, 520, 0.0, 0.0, 25), ("random", 6, 33000, 0.02, 0.3, 26),
                                                                  ("bushy", 900, 1500, 0.003, 0.0, 27)])
def test_gpu_fitch_rows_that_are_never_stored(kind, n, n_sites, p_var, p_internal, seed, monkeypatch):
    """Few cells per 512-site tile: most (leaf, tile) items have none and get no row on the device (ugp_fitch.hip, round 6) -- with and
    without cells on internal nodes, tiles that end inside a word row, a tree without any cell; one pass, passes of 520 sites, and the
    exact listing pass."""
    from usher_amd.fitch import fitch_sankoff
    rng = np.random.default_rng(seed)
    parent = random_bfs_tree(rng, n, kind)
    ref, off, nodes, nucs = random_sites(rng, parent, n_sites, p_var=p_var, p_internal=p_internal, p_dense=2 * p_var)
    want = oracle_mutations(parent, ref, off, nodes, nucs)
    # one pass; passes of 512 sites; the exact listing pass; the cells uploaded in pieces of whole tiles (as large inputs are)
    for budget, cap, pieces in ((None, None, None), (str(n * 4 * 65), None, None), (None, "3", None), (None, None, "4"), (None, None, "2")):
        monkeypatch.delenv("UGP_FITCH_BYTES", raising=False)
        monkeypatch.delenv("UGP_FITCH_EMIT_CAP", raising=False)
        monkeypatch.delenv("UGP_FITCH_PIECES", raising=False)
        if budget:
            monkeypatch.setenv("UGP_FITCH_BYTES", budget)
        if cap:
            monkeypatch.setenv("UGP_FITCH_EMIT_CAP", cap)
        if pieces:
            monkeypatch.setenv("UGP_FITCH_PIECES", pieces)
        site, node, mpar, mnuc = fitch_sankoff(parent, ref, off, nodes, nucs)
        got = list(zip(site.tolist(), node.tolist(), mpar.tolist(), mnuc.tolist()))
        assert got == want


@pytest.mark.gpu
def test_gpu_fitch_duplicate_cells_and_bad_input():
    from usher_amd.fitch import fitch_sankoff
    parent = np.array([-1, 0, 0, 1, 1], np.int64)
    # node 3 named twice at site 0: the later cell wins (usher_mapper.cpp:47-62 runs in order)
    ref = np.array([1, 2], np.uint8)
    off = np.array([0, 3, 4], np.uint64)
    nodes = np.array([3, 4, 3, 2], np.uint32)
    nucs = np.array([2, 4, 4, 8], np.uint8)
    site, node, mpar, mnuc = fitch_sankoff(parent, ref, off, nodes, nucs)
    dedup_nodes, dedup_nucs = np.array([3, 4, 2], np.uint32), np.array([4, 4, 8], np.uint8)
    want = oracle_mutations(parent, ref, np.array([0, 2, 3], np.uint64), dedup_nodes, dedup_nucs)
    assert list(zip(site.tolist(), node.tolist(), mpar.tolist(), mnuc.tolist())) == want
    with pytest.raises(RuntimeError, match="breadth-first"):
        fitch_sankoff(np.array([-1, 0, 1, 0], np.int64), ref, off, nodes, nucs)
    with pytest.raises(RuntimeError, match="A,C,G,T"):
        fitch_sankoff(parent, np.array([3, 2], np.uint8), off, nodes, nucs)
