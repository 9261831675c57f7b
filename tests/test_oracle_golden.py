"""Pin the CPU oracle (oracle/ugp_oracle.c) to the recorded reference outputs
(tests/golden/survey_ref, see its README for provenance) and to the
reference's only in-tree known-answer test (scripts/testBranchLen2)."""
import gzip
import os
import re

import numpy as np
import pytest

from oracle import capi, refio

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SURVEY = os.path.join(GOLD, "survey_ref")
FIX = os.path.join(GOLD, "ref_fixtures")


def _load(pb, vcf):
    T = refio.load_mutation_annotated_tree(pb)
    samples = refio.read_vcf(T, vcf)
    arrays = refio.tree_to_bfs_arrays(T)
    return T, arrays, capi.OracleTree(arrays), [refio.sample_to_arrays(s) for s in samples]


def _read_stats(path):
    rows = []
    with open(path) as f:
        for line in f:
            w = line.rstrip("\n").split("\t")
            if len(w) >= 3:
                rows.append((w[0], int(w[1]), int(w[2]), w[3] if len(w) > 3 else ""))
    return rows


def _read_scores(path):
    out = {}
    with gzip.open(path, "rt") as f:
        next(f)
        for line in f:
            w = line.rstrip("\n").split("\t")
            out.setdefault(w[0], []).append((w[1], int(w[2]), w[4], w[5]))
    return out


@pytest.fixture(scope="module")
def global_case():
    return _load(os.path.join(SURVEY, "global", "global_assignments.pb"), os.path.join(FIX, "new_samples.vcf"))


def test_global_pb_shape(global_case):
    T, arrays, ot, samples = global_case
    assert arrays["n"] == 474
    assert len(arrays["mut_pos"]) == 500
    assert len(T.condensed_nodes) == 69
    assert [s["name"] for s in samples] == ["Sample%d" % i for i in range(1, 6)]


def test_global_no_add_matches_reference(global_case):
    """usher -i g.pb -v new_samples.vcf -n  ->  1 2 / 2 2 / 2 2 / 3 2 / 3 2."""
    _, _, ot, samples = global_case
    want = _read_stats(os.path.join(SURVEY, "global", "out3", "placement_stats.tsv"))
    assert [(w[1], w[2]) for w in want] == [(1, 2), (2, 2), (2, 2), (3, 2), (3, 2)]
    for s, w in zip(samples, want):
        r = ot.place(s)
        assert (s["name"], r["best"], r["num_best"]) == w[:3]


def test_global_per_node_scores_match_reference(global_case):
    """usher -p: 2,370 rows; optimal nodes are node_7 and node_11 for every sample."""
    _, arrays, ot, samples = global_case
    want = _read_scores(os.path.join(SURVEY, "global", "out4", "parsimony-scores.tsv.gz"))
    total = 0
    for s in samples:
        r = ot.place(s, compute_scores=True)
        rows = want[s["name"]]
        assert [x[0] for x in rows] == arrays["names"]
        assert [x[1] for x in rows] == r["scores"].tolist()
        opt = [x[0] for x in rows if x[2] == "y"]
        assert opt == ["node_7", "node_11"]
        assert [arrays["names"][j] for j in r["ties"]] == opt
        total += len(rows)
    assert total == 2370


def _format_excess(vecs, k):
    return ",".join(refio.Mutation(p, r, pa, m).get_string() for (p, r, pa, m) in vecs["excess"][:k])


def test_global_excess_strings_match_reference(global_case):
    """The -p column 'Parsimony-increasing mutations' prints the first `score`
    entries of the node's excess vector (usher_common.cpp:565-572)."""
    _, arrays, ot, samples = global_case
    want = _read_scores(os.path.join(SURVEY, "global", "out4", "parsimony-scores.tsv.gz"))
    for s in samples:
        for j, (name, score, opt, txt) in enumerate(want[s["name"]]):
            if opt != "y":
                assert txt == "N/A"
                continue
            v = ot.node_vecs(s, j)
            got = ("*" if score == 0 else "") + _format_excess(v, score)
            assert got == txt


@pytest.fixture(scope="module")
def syn_case():
    return _load(os.path.join(SURVEY, "syn", "tree.pb"), os.path.join(SURVEY, "syn", "query.vcf"))


def test_syn_per_node_scores_match_reference(syn_case):
    """67,950 (node, sample) scores on a random tree with N / IUPAC queries."""
    _, arrays, ot, samples = syn_case
    want = _read_scores(os.path.join(SURVEY, "syn", "o2", "parsimony-scores.tsv.gz"))
    assert arrays["n"] == 1359 and len(samples) == 50
    n_rows = 0
    for s in samples:
        r = ot.place(s, compute_scores=True)
        rows = want[s["name"]]
        assert [x[0] for x in rows] == arrays["names"]
        assert [x[1] for x in rows] == r["scores"].tolist()
        assert sorted(arrays["names"][j] for j in r["ties"]) == sorted(x[0] for x in rows if x[2] == "y")
        n_rows += len(rows)
    assert n_rows == 67950


def _imputed_string(vecs):
    return ";".join("%d:%s" % (p, refio.get_nuc(m)) for (p, r, pa, m) in vecs["imputed"])


def test_syn_no_add_stats_and_imputed_match_reference(syn_case):
    _, arrays, ot, samples = syn_case
    want = _read_stats(os.path.join(SURVEY, "syn", "o3", "placement_stats.tsv"))
    assert len(want) == 50
    for s, w in zip(samples, want):
        r = ot.place(s)
        assert (s["name"], r["best"], r["num_best"]) == w[:3]
        v = ot.node_vecs(s, r["best_j"])
        assert _imputed_string(v) == w[3]


def test_syn_tie_sets_and_winner_match_debug_log(syn_case):
    """A -DDEBUG=1 build prints every tied node, the sibling/child label and a
    star on the chosen node (usher_common.cpp:495-512)."""
    _, arrays, ot, samples = syn_case
    by_name = {s["name"]: s for s in samples}
    cur = None
    seen = {}
    with open(os.path.join(SURVEY, "syn", "o3", "log")) as f:
        for line in f:
            m = re.search(r"Sample name: (\S+)\tParsimony score: (\d+)\tNumber of parsimony-optimal placements: (\d+)", line)
            if m:
                cur = m.group(1)
                seen[cur] = {"best": int(m.group(2)), "num_best": int(m.group(3)), "nodes": []}
                continue
            m = re.match(r"Best node \((sibling|child)\)(\*?): (\S+)\t", line)
            if m and cur:
                seen[cur]["nodes"].append((m.group(3), m.group(1), m.group(2) == "*"))
    assert len(seen) == 50
    multi = 0
    for name, info in seen.items():
        r = ot.place(by_name[name])
        assert (r["best"], r["num_best"]) == (info["best"], info["num_best"])
        got = {}
        is_leaf = {}
        child_count = np.bincount(arrays["parent"][1:], minlength=arrays["n"])
        for j, hu in zip(r["ties"], r["ties_has_unique"]):
            label = "sibling" if (child_count[j] == 0 or hu) else "child"
            got[arrays["names"][j]] = (label, j == r["best_j"])
        want = {n: (lab, star) for (n, lab, star) in info["nodes"]}
        assert got == want
        multi += info["num_best"] > 1
    assert multi >= 5


def test_big_no_add_matches_reference():
    import tempfile
    with gzip.open(os.path.join(SURVEY, "big", "tree.pb.gz"), "rb") as f, tempfile.NamedTemporaryFile(suffix=".pb") as tmp:
        tmp.write(f.read())
        tmp.flush()
        _, arrays, ot, samples = _load(tmp.name, os.path.join(SURVEY, "big", "query.vcf"))
    assert arrays["n"] == 44483
    want = _read_stats(os.path.join(SURVEY, "big", "o3", "placement_stats.tsv"))
    assert len(want) == 64
    for s, w in list(zip(samples, want))[:16]:
        r = ot.place(s)
        assert (s["name"], r["best"], r["num_best"]) == w[:3]
        assert ot.place_mt(s, 4) == {"best": r["best"] - 0, "num_best": r["num_best"], "best_j": r["best_j"]}


def test_fitch_sankoff_testBranchLen2():
    """scripts/testBranchLen2.sh: the input newick's branch lengths are the
    expected per-branch mutation counts."""
    T = refio.create_tree_from_newick(os.path.join(FIX, "testBranchLen2.nwk"))
    arrays = refio.tree_to_bfs_arrays(T)
    bfs = T.breadth_first_expansion()
    idx = {n.identifier: j for j, n in enumerate(bfs)}
    with open(os.path.join(FIX, "testBranchLen2.vcf")) as f:
        lines = [l.split() for l in f if not l.startswith("##")]
    header, rows = lines[0], lines[1:]
    names = header[9:]
    for w in rows:
        ref = refio.get_nuc_id(w[3][0])
        alts = w[4].split(",")
        var_node, var_nuc = [], []
        for name, cell in zip(names, w[9:]):
            if cell[0].isdigit():
                a = int(cell)
                if a > 0:
                    var_node.append(idx[name])
                    var_nuc.append(refio.get_nuc_id(alts[a - 1][0]))
            else:
                var_node.append(idx[name])
                var_nuc.append(15)
        state, mpar, mnuc = capi.fitch_site(arrays["parent"], ref, np.array(var_node), np.array(var_nuc))
        for j, node in enumerate(bfs):
            if mnuc[j]:
                node.add_mutation(refio.Mutation(int(w[1]), ref, int(mpar[j]), int(mnuc[j])))
    got = refio.get_newick_string(T, print_internal=True, print_branch_len=True)
    with open(os.path.join(SURVEY, "branchlen2", "final-tree.nh")) as f:
        want = f.read().strip()
    assert want == "((a:0,(b:0,(c:0,d:1)node_4:1)node_3:2,((e:0,f:1)node_6:3,g:0)node_5:4)node_2:5,h:0)node_1:0;"
    assert got == want
    # and the same counts as the branch lengths written in the reference's input newick
    with open(os.path.join(FIX, "testBranchLen2.nwk")) as f:
        src = f.read().strip()
    assert re.sub(r"node_\d+", "", got.replace(":0;", ";")) == src
