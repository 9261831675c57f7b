"""Phase 2 without a walk for single-node minima (round 6; ugp_kernels.hpp Phase2Uniq): k_best8 keeps the second-smallest cost of every
chunk, k_descend names the node whose cost it reports, and a sample whose global minimum is attained in one chunk, by one node, at
the descent's cost is answered from that node -- pass 2 of the reference visits tied nodes only (usher_common.cpp:416-449).  The
answers must be the oracle's and the walk's (UGP_NO_UNIQ=1) for trees with many ties, few ties, masked mutations, excluded nodes."""
import numpy as np
import pytest

from oracle import capi
from tests import synth
from usher_amd import Placer, QueryBatch

pytestmark = pytest.mark.gpu


def _check(arrays, queries, monkeypatch, every=1):
    batch = QueryBatch(queries)
    monkeypatch.delenv("UGP_NO_UNIQ", raising=False)
    a = Placer(arrays)
    ra = a.place(batch)
    assert a.timing()["packed_path"] == 1
    a.close()
    monkeypatch.setenv("UGP_NO_UNIQ", "1")
    b = Placer(arrays)
    rb = b.place(batch)
    b.close()
    monkeypatch.delenv("UGP_NO_UNIQ", raising=False)
    assert (ra["num_best"] >= 1).all()
    assert (ra.view(np.int32) == rb.view(np.int32)).all()
    ot = capi.OracleTree(arrays)
    for i in range(0, len(queries), every):
        w = ot.place(queries[i])
        got = (int(ra["best_set_difference"][i]), int(ra["num_best"][i]), int(ra["best_j"][i]), bool(ra["best_has_unique"][i]))
        assert got == (w["best"], w["num_best"], w["best_j"], w["has_unique"]), (i, got, w)
    return ra


@pytest.mark.parametrize("seed,n_leaves,n_sites,kw", [
    (601, 3000, 60, dict(n_ambig=(0, 0, 2))),                      # few sites: most samples tie, many ways
    (602, 6000, 900, dict(n_ambig=(0, 0, 0, 3))),                  # many sites: most minima are attained by one node
    (603, 4000, 300, dict(p_masked=0.05, n_ambig=(0, 2, 6))),      # masked mutations: has_unique of the single node
    (604, 5000, 400, dict(root_muts=4, mut_counts=(0, 0, 0, 1, 1, 2))),   # many empty branches: internal nodes without mutations tie with their parents
])
def test_single_node_minima_without_a_walk(seed, n_leaves, n_sites, kw, monkeypatch):
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")   # (the shortcut belongs to the sorted main walk: needs the locality pre-pass and its descent)
    arrays, queries = synth.make_case(seed, n_leaves=n_leaves, n_queries=1300, n_sites=n_sites, **kw)
    res = _check(arrays, queries, monkeypatch, every=3)
    frac_single = float((res["num_best"] == 1).mean())
    assert 0.02 < frac_single < 0.999


def test_samples_that_are_nodes_of_the_tree(monkeypatch):
    """Queries with no substitutions: the genotype of a node -- cost 0 at that node, ties with every mutation-free relative."""
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    rng = np.random.default_rng(77)
    arrays, ref, sites, state = synth.random_tree(rng, 5000, genome_len=3000, n_sites=500)
    queries = [synth.random_query(rng, arrays, ref, sites, state, 3000, n_subst=(0,), n_ambig=(0,), name="E%d" % i) for i in range(900)]
    res = _check(arrays, queries, monkeypatch, every=2)
    assert (res["best_set_difference"] == 0).mean() > 0.9


def test_excluded_nodes_are_never_answered_from_the_descent(monkeypatch):
    """Add mode's exclusions (ugp_mat_update): a node rewritten since the flattening is no candidate -- neither for the walk nor for
    the descent's shortcut; the answers equal the oracle's search over the remaining nodes."""
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    arrays, queries = synth.make_case(611, n_leaves=4000, n_queries=800, n_sites=500, n_ambig=(0, 0, 1))
    batch = QueryBatch(queries)
    pl = Placer(arrays)
    first = pl.place(batch)
    # exclude the winners of every fifth sample (internal or leaf, never the root)
    victims = sorted({int(j) for j in first["best_j"][::5] if int(j) != 0})
    pl.update([{"flat_j": j, "leaf": False, "masked": False, "path": [], "own": []} for j in victims], [])
    second = pl.place(batch)
    pl.close()
    assert (second["num_best"] >= 1).all()
    assert not np.isin(second["best_j"], victims).any()
    ot = capi.OracleTree(arrays)
    allowed = np.array([j for j in range(int(arrays["n"])) if j not in set(victims)], np.int64)
    for i in range(0, len(queries), 7):
        w = ot.place_list(queries[i], allowed, tie_cap=1 << 16)
        assert (int(second["best_set_difference"][i]), int(second["num_best"][i]), int(second["best_j"][i])) == (w["best"], w["num_best"], int(allowed[w["best_j"]])), i
