"""The closed form the GPU evaluates (oracle/closed_form.py) must agree with the
literal restatement (oracle/ugp_oracle.c) bit for bit, on the recorded
reference fixtures and on seeded random trees covering the edge cases."""
import os

import numpy as np
import pytest

from oracle import capi, closed_form, refio
from tests import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _check(arrays, samples, scores=True):
    ot = capi.OracleTree(arrays)
    cf = closed_form.ClosedFormTree(arrays)
    for s in samples:
        a = ot.place(s, compute_scores=scores)
        b = cf.place(s, compute_scores=scores)
        for k in ("best", "num_best", "best_j", "has_unique"):
            assert a[k] == b[k], (s["name"], k, a[k], b[k])
        assert a["ties"].tolist() == b["ties"].tolist()
        assert a["ties_has_unique"].tolist() == b["ties_has_unique"].tolist()
        if scores:
            assert a["scores"].tolist() == b["scores"].tolist()
        # the default (two-pass) path must agree with the -p path too
        c = ot.place(s, compute_scores=False)
        for k in ("best", "num_best", "best_j", "has_unique"):
            assert a[k] == c[k]


def test_closed_form_on_global_fixture():
    T = refio.load_mutation_annotated_tree(os.path.join(GOLD, "survey_ref", "global", "global_assignments.pb"))
    samples = [refio.sample_to_arrays(s) for s in refio.read_vcf(T, os.path.join(GOLD, "ref_fixtures", "new_samples.vcf"))]
    _check(refio.tree_to_bfs_arrays(T), samples)


def test_closed_form_on_syn_fixture():
    T = refio.load_mutation_annotated_tree(os.path.join(GOLD, "survey_ref", "syn", "tree.pb"))
    samples = [refio.sample_to_arrays(s) for s in refio.read_vcf(T, os.path.join(GOLD, "survey_ref", "syn", "query.vcf"))]
    _check(refio.tree_to_bfs_arrays(T), samples[:20])


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_closed_form_random(seed):
    arrays, queries = synth.make_case(seed, n_leaves=60 + 20 * seed, n_queries=12, n_sites=40 + 10 * seed)
    _check(arrays, queries)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_closed_form_random_masked_and_root_muts(seed):
    arrays, queries = synth.make_case(seed, n_leaves=80, n_queries=12, n_sites=30, p_masked=0.15, root_muts=3)
    assert (arrays["mut_pos"] < 0).any()
    _check(arrays, queries)


def test_closed_form_tiny_trees():
    # single node, and a root with two leaves
    one = {"n": 1, "parent": np.array([-1]), "mut_off": np.array([0, 0]), "mut_pos": np.zeros(0, np.int32),
           "mut_ref": np.zeros(0, np.int8), "mut_par": np.zeros(0, np.int8), "mut_nuc": np.zeros(0, np.int8), "names": ["r"]}
    empty = {"name": "e", "pos": np.zeros(0, np.int32), "ref": np.zeros(0, np.int8), "nuc": np.zeros(0, np.int8),
             "is_missing": np.zeros(0, np.int8)}
    q = {"name": "q", "pos": np.array([5, 9], np.int32), "ref": np.array([1, 2], np.int8), "nuc": np.array([4, 15], np.int8),
         "is_missing": np.array([0, 1], np.int8)}
    _check(one, [empty, q])
    three = {"n": 3, "parent": np.array([-1, 0, 0]), "mut_off": np.array([0, 0, 1, 1]), "mut_pos": np.array([5], np.int32),
             "mut_ref": np.array([1], np.int8), "mut_par": np.array([1], np.int8), "mut_nuc": np.array([4], np.int8), "names": list("rab")}
    _check(three, [empty, q])


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_c_closed_form_and_pool_equal_literal_oracle(seed):
    """oracle/ugp_oracle.c's O(N+M) closed-form sweep (the full-size checker) and the pooled node-parallel
    literal routine (the timed CPU baseline) against the literal single-thread oracle: placements, tie
    lists, per-node scores."""
    from oracle import capi
    arrays, queries = synth.make_case(500 + seed, n_leaves=400, n_queries=50, n_sites=90, p_masked=0.04 if seed % 2 else 0.0,
                                      root_muts=seed % 3, n_ambig=(0, 0, 2, 5, 30))
    ot = capi.OracleTree(arrays)
    cf = capi.ClosedFormC(ot)
    lens = [len(s["pos"]) for s in queries]
    off = np.zeros(len(queries) + 1, np.int64)
    off[1:] = np.cumsum(lens)
    cat = lambda k, dt: np.concatenate([np.asarray(s[k]) for s in queries]).astype(dt)
    batch = cf.place_csr(off, cat("pos", np.int32), cat("ref", np.int8), cat("nuc", np.int8), cat("is_missing", np.int8), nthreads=3, tie_cap=4096)
    for i, s in enumerate(queries):
        w = ot.place(s, compute_scores=True)
        assert (int(batch["best"][i]), int(batch["num_best"][i]), int(batch["best_j"][i]), bool(batch["has_unique"][i])) == \
               (w["best"], w["num_best"], w["best_j"], w["has_unique"])
        assert batch["ties"][i].tolist() == w["ties"].tolist()
        assert batch["ties_has_unique"][i].tolist() == w["ties_has_unique"].tolist()
        assert cf.scores(s).tolist() == w["scores"].tolist()
        for threads in (1, 5):
            m = ot.place_mt(s, threads)
            assert (m["best"], m["num_best"], m["best_j"]) == (w["best"], w["num_best"], w["best_j"])
