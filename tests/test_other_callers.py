"""The other callers of mapper2_body -- matUtils uncertainty / annotate / merge and ripples -- through the extended C ABI
(ugp_place_batch_ex, ugp_tied_nodes_ex, ugp_node_order, ugp_subtree_mask) against the oracle's restatement of how each
of them drives the routine (orc_place_sample_list): their own node vector and index j, a node left out, a level-capped
subtree, a per-node distance, per-node scores."""
import numpy as np
import pytest

from oracle import capi
from tests import synth
from usher_amd import Placer, QueryBatch

pytestmark = pytest.mark.gpu


def _node_sample(arrays, ref_at, j, name):
    """The mutation set of node j as those callers build it: most recent mutation per position on root -> j
    (uncertainty.cpp:143-166), here sorted by position (the ABI's precondition)."""
    rows = {}
    a = j
    while a >= 0:
        for i in range(int(arrays["mut_off"][a]), int(arrays["mut_off"][a + 1])):
            p = int(arrays["mut_pos"][i])
            if p >= 0 and p not in rows:
                rows[p] = (int(arrays["mut_ref"][i]), int(arrays["mut_nuc"][i]))
        a = int(arrays["parent"][a])
    ps = sorted(rows)
    return {"name": name, "pos": np.asarray(ps, np.int32), "ref": np.asarray([rows[p][0] for p in ps], np.int8),
            "nuc": np.asarray([rows[p][1] for p in ps], np.int8), "is_missing": np.zeros(len(ps), np.int8)}


def _same(res, i, w):
    assert (int(res["best_set_difference"][i]), int(res["num_best"][i]), int(res["best_j"][i]), bool(res["best_has_unique"][i])) == \
           (w["best"], w["num_best"], w["best_j"], w["has_unique"]), (i, res[i], w)


@pytest.mark.parametrize("seed", [1, 2])
def test_uncertainty_style_depth_first_indices_and_self_exclusion(seed):
    """matUtils uncertainty (uncertainty.cpp:132-255): the sample is a tree node's own mutation set, every node but that
    node is scored, indices are positions in the depth-first expansion; equally parsimonious placements are listed."""
    arrays, _ = synth.make_case(300 + seed, n_leaves=500, n_queries=1, n_sites=90, p_masked=0.02 if seed == 2 else 0.0)
    n = arrays["n"]
    pl = Placer(arrays, chunk_nodes=40)
    dfs = pl.node_order("dfs").astype(np.int64)              # dfs position -> BFS index
    assert sorted(dfs.tolist()) == list(range(n)) and dfs[0] == 0
    pos_of = np.empty(n, np.int64); pos_of[dfs] = np.arange(n)
    rng = np.random.default_rng(seed)
    leaves = np.setdiff1d(np.arange(n), arrays["parent"][1:])
    picks = rng.choice(leaves, 70, replace=False)
    samples = [_node_sample(arrays, None, int(j), "n%d" % j) for j in picks]
    batch = QueryBatch(samples)
    skip = pos_of[picks].astype(np.uint32)
    res = pl.place_ex(batch, order="dfs", skip_node=skip)
    ties, ties_hu, tc = pl.tied_nodes_ex(batch, 512, order="dfs", skip_node=skip)
    ot = capi.OracleTree(arrays)
    root_muts = int(arrays["mut_off"][1])
    for i, s in enumerate(samples):
        keep = dfs != picks[i]
        w = ot.place_list(s, dfs[keep], jidx=np.arange(n)[keep], init_best=len(s["pos"]) + root_muts + 1)
        _same(res, i, w)
        assert int(tc[i]) == w["num_best"] and ties[i].tolist() == w["ties"].tolist() and ties_hu[i].tolist() == w["ties_has_unique"].tolist()
    pl.close()


def test_ripples_style_mask_distance_and_scores():
    """ripples (ripples/main.cpp:333-377): nodes with fewer descendant leaves than a threshold are not scored, every
    scored node reports its parsimony score, a per-node distance breaks ties before the leaf count."""
    arrays, queries = synth.make_case(411, n_leaves=600, n_queries=40, n_sites=100, n_ambig=(0, 0, 2))
    n = arrays["n"]
    ot = capi.OracleTree(arrays)
    leaves_below = np.array([ot.num_leaves(j) for j in range(n)])
    mask = (leaves_below >= 3).astype(np.uint8)
    mask[0] = 1
    rng = np.random.default_rng(5)
    pl = Placer(arrays, chunk_nodes=64)
    batch = QueryBatch(queries)
    nodes = np.flatnonzero(mask)
    for dist in (None, rng.integers(0, 3, n).astype(np.uint32)):
        res, scores = pl.place_ex(batch, order="bfs", node_mask=mask, distance=dist, want_scores=True)
        for i, s in enumerate(queries):
            w = ot.place_list(s, nodes, jidx=nodes, distance=None if dist is None else dist[nodes], compute_scores=True)
            _same(res, i, w)
            assert scores[i][nodes].tolist() == w["scores"].tolist()
            assert not scores[i][mask == 0].any()
    pl.close()


def test_merge_style_level_capped_subtree():
    """matUtils merge (merge.cpp:236-280): the search runs over the breadth-first expansion of the subtree of a
    consistent node, cut max_levels below it; the whole tree's breadth-first order restricted to that subtree lists
    the nodes in the same relative order, so positions translate by ranking."""
    arrays, queries = synth.make_case(512, n_leaves=700, n_queries=30, n_sites=100)
    n = arrays["n"]
    pl = Placer(arrays, chunk_nodes=50)
    ot = capi.OracleTree(arrays)
    internal = np.unique(arrays["parent"][1:])
    rng = np.random.default_rng(2)
    for root in (0, int(rng.choice(internal[internal > 0])), int(rng.choice(internal[internal > 0]))):
        for max_levels in (2, 5, 1000):
            mask = pl.subtree_mask(root, max_levels)
            nodes = np.flatnonzero(mask)
            assert nodes[0] == root
            res = pl.place_ex(QueryBatch(queries), order="bfs", node_mask=mask)
            for i, s in enumerate(queries):
                # (merge starts from |S| + |root-of-subtree mutations| + 1, merge.cpp:245, and keeps (that, 1, 0) when no admitted
                # node is at least as good; the library reports the search itself -- a caller compares with its own bound)
                w = ot.place_list(s, nodes, jidx=nodes)
                if int(res["num_best"][i]) == 0:             # nothing eligible in the admitted set
                    assert w["best"] == 10 ** 9 and int(res["best_set_difference"][i]) == 2 ** 31 - 1
                else:
                    _same(res, i, w)
    pl.close()


def test_depth_first_order_of_a_tree_that_is_not_numbered_breadth_first():
    """The boundary only promises parent[j] < j.  A tree numbered in preorder IS its own depth-first expansion, so
    ugp_node_order(DFS) is the identity there, DFS-order calls equal BFS-order calls, and both equal the oracle
    (the children of a node are not contiguous in such a numbering)."""
    arrays0, queries = synth.make_case(613, n_leaves=420, n_queries=50, n_sites=90, p_masked=0.02)
    arrays, new = synth.relabel_preorder(arrays0)
    n = arrays["n"]
    assert (np.diff(arrays["parent"][1:]) < 0).any()
    pl = Placer(arrays, chunk_nodes=32)
    dfs = pl.node_order("dfs").astype(np.int64)
    assert dfs.tolist() == list(range(n))
    batch = QueryBatch(queries)
    ot = capi.OracleTree(arrays)
    res = pl.place(batch)
    rng = np.random.default_rng(3)
    skip = rng.integers(0, n, len(queries)).astype(np.uint32)
    res_b = pl.place_ex(batch, order="bfs", skip_node=skip)
    res_d = pl.place_ex(batch, order="dfs", skip_node=skip)
    ties_d, hu_d, tc_d = pl.tied_nodes_ex(batch, 256, order="dfs", skip_node=skip)
    allj = np.arange(n)
    for i, s in enumerate(queries):
        _same(res, i, ot.place(s))
        keep = allj != skip[i]
        w = ot.place_list(s, allj[keep], jidx=allj[keep])
        _same(res_b, i, w)
        _same(res_d, i, w)
        assert int(tc_d[i]) == w["num_best"] and ties_d[i].tolist() == w["ties"].tolist() and hu_d[i].tolist() == w["ties_has_unique"].tolist()
    # level-capped subtree masks, both orders, against a plain walk over parent[]
    par = arrays["parent"]
    depth = np.zeros(n, np.int64)
    for j in range(1, n):
        depth[j] = depth[par[j]] + 1
    root = int(np.flatnonzero(np.bincount(par[1:], minlength=n) >= 2)[5])
    inside = np.zeros(n, bool)
    inside[root] = True
    for j in range(root + 1, n):
        inside[j] = inside[par[j]]
    for lv in (1, 3, 1000):
        want = inside & (depth - depth[root] <= lv)
        assert pl.subtree_mask(root, lv, order="bfs").astype(bool).tolist() == want.tolist()
        assert pl.subtree_mask(root, lv, order="dfs").astype(bool).tolist() == want.tolist()
    pl.close()


def test_extended_searches_on_the_packed_pruned_path(monkeypatch):
    """Round 4: an extended search whose options the packed path can express runs there -- a node order / a distance is a tie rank
    of phase 2, a node mask is a temporary exclusion (the "no candidate" bit, set in the tree and in the coarse tree of the locality
    pre-pass for the call, cleared behind it).  annotate-style (depth-first indices), ripples-style without the score matrix (mask by
    descendant leaves + distance) and merge-style from the root (level-capped subtree), each with its tie lists: packed == the
    one-sample-per-lane kernel (UGP_EX_SLOW=1) == the oracle's restatement of the call sites; a plain search afterwards sees every
    node again."""
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    monkeypatch.delenv("UGP_EX_SLOW", raising=False)
    arrays, queries = synth.make_case(777, n_leaves=2500, n_queries=700, n_sites=140, n_ambig=(0, 0, 2, 5), p_masked=0.01)
    n = arrays["n"]
    ot = capi.OracleTree(arrays)
    batch = QueryBatch(queries)
    fast = Placer(arrays, chunk_nodes=48)
    monkeypatch.setenv("UGP_EX_SLOW", "1")
    slow = Placer(arrays, chunk_nodes=48)
    monkeypatch.delenv("UGP_EX_SLOW")
    plain = fast.place(batch).copy()
    dfs = fast.node_order("dfs").astype(np.int64)
    leaves_below = np.array([ot.num_leaves(j) for j in range(n)])
    rng = np.random.default_rng(9)
    dist = rng.integers(0, 3, n).astype(np.uint32)
    m_leaves = (leaves_below >= 3).astype(np.uint8); m_leaves[0] = 1
    m_levels = fast.subtree_mask(0, 6)
    cases = [("annotate", dict(order="dfs"), dfs, np.arange(n), None),
             ("ripples", dict(order="bfs", node_mask=m_leaves, distance=dist), np.flatnonzero(m_leaves), np.flatnonzero(m_leaves), dist),
             ("merge", dict(order="bfs", node_mask=m_levels), np.flatnonzero(m_levels), np.flatnonzero(m_levels), None)]
    for name, kw, nodes, jidx, d in cases:
        a = fast.place_ex(batch, **kw)
        assert fast.timing()["packed_path"] == 1, name
        b = slow.place_ex(batch, **kw)
        assert slow.timing()["packed_path"] == 0, name
        assert (a.view(np.int32) == b.view(np.int32)).all(), name
        ta, ha, ca = fast.tied_nodes_ex(batch, 256, **kw)
        tb, hb, cb = slow.tied_nodes_ex(batch, 256, **kw)
        assert (ca == cb).all() and all(x.tolist() == y.tolist() for x, y in zip(ta, tb)) and all(x.tolist() == y.tolist() for x, y in zip(ha, hb)), name
        for i in range(0, len(queries), 7):
            w = ot.place_list(queries[i], nodes, jidx=jidx, distance=None if d is None else d[nodes])
            _same(a, i, w)
            assert int(ca[i]) == w["num_best"] and ta[i].tolist() == w["ties"].tolist() and ha[i].tolist() == w["ties_has_unique"].tolist(), (name, i)
        assert (fast.place(batch).view(np.int32) == plain.view(np.int32)).all(), name        # the mask is gone again
    fast.close(); slow.close()


@pytest.mark.parametrize("seed", [5, 6])
def test_self_exclusion_on_the_packed_pruned_path(seed, monkeypatch):
    """Round 4: a search that leaves one node out per sample (matUtils uncertainty: a sample is never mapped onto its own node,
    uncertainty.cpp:216) runs on the packed, pruned path too: bounds that never rely on the excluded node (seeds that skip it, no
    tightening from chunk minima), then the one chunk minimum it may have set is recomputed without it, and phase 2 leaves it out of
    the ties.  Samples = the mutation sets of tree nodes -- leaves, internal nodes, nodes of the coarse tree of the locality pre-pass --
    with their own node left out, some with nothing left out, plus ordinary queries with a random node left out; depth-first
    indices, and once with a node mask and a distance on top.  packed == one sample per lane (UGP_EX_SLOW=1) == the oracle."""
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    monkeypatch.delenv("UGP_EX_SLOW", raising=False)
    arrays, queries = synth.make_case(880 + seed, n_leaves=2600, n_queries=260, n_sites=150, n_ambig=(0, 0, 2), p_masked=0.01 if seed == 6 else 0.0)
    n = arrays["n"]
    ot = capi.OracleTree(arrays)
    fast = Placer(arrays, chunk_nodes=48)
    monkeypatch.setenv("UGP_EX_SLOW", "1")
    slow = Placer(arrays, chunk_nodes=48)
    monkeypatch.delenv("UGP_EX_SLOW")
    rng = np.random.default_rng(seed)
    dfs = fast.node_order("dfs").astype(np.int64)
    pos_of = np.empty(n, np.int64); pos_of[dfs] = np.arange(n)
    size = np.ones(n, np.int64)
    for j in range(n - 1, 0, -1):
        size[arrays["parent"][j]] += size[j]
    big = np.argsort(-size)[1:40]                                   # the top of the tree: nodes of the coarse tree
    picks = np.concatenate([rng.choice(np.arange(1, n), 420, replace=False), big])
    samples = [_node_sample(arrays, None, int(j), "n%d" % j) for j in picks] + list(queries)
    skip_bfs = np.concatenate([picks, rng.integers(0, n, len(queries))]).astype(np.int64)
    none = rng.random(len(samples)) < 0.1
    batch = QueryBatch(samples)
    assert len(samples) > 512
    leaves_below = np.array([ot.num_leaves(j) for j in range(n)])
    m_leaves = (leaves_below >= 2).astype(np.uint8); m_leaves[0] = 1
    dist = rng.integers(0, 2, n).astype(np.uint32)
    root_muts = int(arrays["mut_off"][1])
    for name, kw, order_nodes in (("dfs", dict(order="dfs"), dfs), ("bfs+mask+distance", dict(order="bfs", node_mask=m_leaves, distance=dist), np.arange(n))):
        idx_of = pos_of if name == "dfs" else np.arange(n)
        skip = np.where(none, 0xFFFFFFFF, idx_of[skip_bfs]).astype(np.uint32)
        a = fast.place_ex(batch, skip_node=skip, **kw)
        assert fast.timing()["packed_path"] == 1, name
        b = slow.place_ex(batch, skip_node=skip, **kw)
        assert slow.timing()["packed_path"] == 0, name
        bad = np.flatnonzero((a.view(np.int32) != b.view(np.int32)).reshape(len(samples), -1).any(axis=1))
        assert len(bad) == 0, (name, bad[:10], a[bad[:3]], b[bad[:3]])
        ta, ha, ca = fast.tied_nodes_ex(batch, 128, skip_node=skip, **kw)
        tb, hb, cb = slow.tied_nodes_ex(batch, 128, skip_node=skip, **kw)
        assert (ca == cb).all() and all(x.tolist() == y.tolist() for x, y in zip(ta, tb)) and all(x.tolist() == y.tolist() for x, y in zip(ha, hb)), name
        for i in list(range(0, len(samples), 9)) + list(range(420, 459)):
            keep = np.ones(n, bool)
            if not none[i]:
                keep[idx_of[skip_bfs[i]]] = False
            if "node_mask" in kw:
                keep &= m_leaves[order_nodes].astype(bool)
            nodes = order_nodes[keep]
            w = ot.place_list(samples[i], nodes, jidx=np.arange(n)[keep], distance=dist[nodes] if "distance" in kw else None,
                              init_best=len(samples[i]["pos"]) + root_muts + 1)
            _same(a, i, w)
            assert int(ca[i]) == w["num_best"] and ta[i].tolist() == w["ties"][:128].tolist(), (name, i)
    fast.close(); slow.close()


def test_self_exclusion_on_a_preorder_numbered_tree_through_the_sorted_path(monkeypatch):
    """a tree that is not numbered breadth-first has no seed descent, and the bounds of a search that leaves
    one node out per sample are then the coarse pass's alone -- which must not be taken from the excluded node itself.  More than
    512 samples (the sorted, seeded, packed path), every sample the mutation set of a tree node with that node left out: the coarse
    winner IS the excluded node for the nodes of the coarse tree.  packed == one sample per lane == the oracle."""
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    monkeypatch.delenv("UGP_EX_SLOW", raising=False)
    arrays0, queries = synth.make_case(990, n_leaves=2600, n_queries=120, n_sites=150, n_ambig=(0, 0, 2))
    arrays, new = synth.relabel_preorder(arrays0)
    n = arrays["n"]
    assert (np.diff(arrays["parent"][1:]) < 0).any()
    ot = capi.OracleTree(arrays)
    fast = Placer(arrays, chunk_nodes=48)
    monkeypatch.setenv("UGP_EX_SLOW", "1")
    slow = Placer(arrays, chunk_nodes=48)
    monkeypatch.delenv("UGP_EX_SLOW")
    rng = np.random.default_rng(17)
    size = np.ones(n, np.int64)
    for j in range(n - 1, 0, -1):
        size[arrays["parent"][j]] += size[j]
    big = np.argsort(-size)[1:80]                                   # the top of the tree: nodes of the coarse tree
    picks = np.concatenate([rng.choice(np.arange(1, n), 500, replace=False), big])
    samples = [_node_sample(arrays, None, int(j), "n%d" % j) for j in picks] + list(queries)
    skip = np.concatenate([picks, rng.integers(0, n, len(queries))]).astype(np.uint32)
    batch = QueryBatch(samples)
    assert len(samples) > 512
    root_muts = int(arrays["mut_off"][1])
    a = fast.place_ex(batch, order="bfs", skip_node=skip)
    assert fast.timing()["packed_path"] == 1
    b = slow.place_ex(batch, order="bfs", skip_node=skip)
    assert slow.timing()["packed_path"] == 0
    bad = np.flatnonzero((a.view(np.int32) != b.view(np.int32)).reshape(len(samples), -1).any(axis=1))
    assert len(bad) == 0, (bad[:10], a[bad[:3]], b[bad[:3]])
    ta, ha, ca = fast.tied_nodes_ex(batch, 128, order="bfs", skip_node=skip)
    tb, hb, cb = slow.tied_nodes_ex(batch, 128, order="bfs", skip_node=skip)
    assert (ca == cb).all() and all(x.tolist() == y.tolist() for x, y in zip(ta, tb))
    allj = np.arange(n)
    for i in list(range(0, len(samples), 11)) + list(range(500, 579, 2)):
        keep = allj != skip[i]
        w = ot.place_list(samples[i], allj[keep], jidx=allj[keep], init_best=len(samples[i]["pos"]) + root_muts + 1)
        _same(a, i, w)
    fast.close(); slow.close()


def test_prepared_options_equal_the_one_shot_calls(monkeypatch):
    """Round 5: ugp_ex_prepare does the node-level part of an extended search once (the caller's order, its mask, its distance ranks --
    ripples keeps one node vector and one distance array for a whole run, ripples/main.cpp:303-377); ugp_place_batch_prepared then
    equals ugp_place_batch_ex call by call: ripples-style (mask + distance) with and without the score matrix -- written into a DEVICE
    buffer by the level-by-level kernel --, merge-style, annotate- and uncertainty-style (depth-first indices, a node left out per
    sample), on the packed path and on the one-sample-per-lane kernel."""
    import torch
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    monkeypatch.delenv("UGP_EX_SLOW", raising=False)
    arrays, queries = synth.make_case(1234, n_leaves=2400, n_queries=640, n_sites=140, n_ambig=(0, 0, 2, 5), p_masked=0.01)
    n = arrays["n"]
    ot = capi.OracleTree(arrays)
    batch = QueryBatch(queries)
    leaves_below = np.array([ot.num_leaves(j) for j in range(n)])
    rng = np.random.default_rng(4)
    dist = rng.integers(0, 3, n).astype(np.uint32)
    m_leaves = (leaves_below >= 3).astype(np.uint8); m_leaves[0] = 1
    for slow in (False, True):
        if slow:
            monkeypatch.setenv("UGP_EX_SLOW", "1")
        pl = Placer(arrays, chunk_nodes=48)
        dfs = pl.node_order("dfs").astype(np.int64)
        pos_of = np.empty(n, np.int64); pos_of[dfs] = np.arange(n)
        skip_b = rng.integers(0, n, len(queries)).astype(np.uint32)
        m_levels = pl.subtree_mask(0, 6)
        for name, kw, skip in (("ripples", dict(order="bfs", node_mask=m_leaves, distance=dist), None),
                               ("merge", dict(order="bfs", node_mask=m_levels), None),
                               ("annotate", dict(order="dfs"), None),
                               ("uncertainty", dict(order="dfs"), pos_of[skip_b].astype(np.uint32)),
                               ("ripples+skip", dict(order="bfs", node_mask=m_leaves, distance=dist), skip_b)):
            ex = pl.prepare_ex(**kw)
            want = pl.place_ex(batch, skip_node=skip, **kw)
            for _ in range(2):                                     # (the handle serves any number of calls)
                got = pl.place_prepared(batch, ex, skip_node=skip)
                assert pl.timing()["packed_path"] == (0 if slow else 1), name
                assert (got.view(np.int32) == want.view(np.int32)).all(), (name, slow)
            if name.startswith("ripples"):                         # the score matrix in a device buffer
                few = batch.slice(0, 24)
                wr, ws = pl.place_ex(few, skip_node=None if skip is None else skip[:24], want_scores=True, **kw)
                d = torch.full((24, n), -7, dtype=torch.int32, device="cuda")
                gr = pl.place_prepared(few, ex, skip_node=None if skip is None else skip[:24], d_scores=d.data_ptr())
                torch.cuda.synchronize()
                assert (gr.view(np.int32) == wr.view(np.int32)).all(), name
                assert (d.cpu().numpy() == ws).all(), (name, slow)
            pl.free_ex(ex)
        plain = pl.place(batch)
        for i in range(0, len(queries), 40):                       # nothing is left masked behind the calls
            _same(plain, i, ot.place(queries[i]))
        pl.close()
        monkeypatch.delenv("UGP_EX_SLOW", raising=False)
