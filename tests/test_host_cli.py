"""The usher-compatible front end (C++ host: loaders, VCF ingest, tree update,
output files) against the recorded reference outputs.  CPU only: the oracle is
plugged in as the placement backend (tests/host_harness.py)."""
import gzip
import os
import shutil

import pytest

from oracle import refio
from tests.host_harness import HOST_LIB, run_usher
import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SURVEY = os.path.join(GOLD, "survey_ref")
FIX = os.path.join(GOLD, "ref_fixtures")


def _read(path):
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt") as f:
        return f.read()


def _pb_semantic(path):
    with (gzip.open if path.endswith(".gz") else open)(path, "rb") as f:
        newick, muts, cond, meta = refio.parse_parsimony_pb(f.read())
    return newick, muts, sorted((k, tuple(v)) for k, v in cond), meta


@pytest.fixture(scope="module")
def global_pb(tmp_path_factory):
    d = tmp_path_factory.mktemp("build")
    vcf = str(d / "global_samples.vcf")
    with gzip.open(os.path.join(FIX, "global_samples.vcf.gz"), "rb") as f, open(vcf, "wb") as o:
        shutil.copyfileobj(f, o)
    pb = str(d / "g.pb")
    rc = run_usher(["-t", os.path.join(FIX, "global_phylo.nh"), "-v", vcf, "-o", pb, "-d", str(d)])
    assert rc == 0
    return pb, str(d)


def test_build_mat_from_newick_and_vcf(global_pb):
    """usher -t global_phylo.nh -v global_samples.vcf -o g.pb  (Fitch-Sankoff + condensing + pb writer)."""
    pb, d = global_pb
    assert _pb_semantic(pb) == _pb_semantic(os.path.join(SURVEY, "global", "global_assignments.pb"))
    assert _read(os.path.join(d, "final-tree.nh")) == _read(os.path.join(SURVEY, "global", "build-final-tree.nh"))
    # and the writer is byte-compatible with the reference's serialisation up to condensed-node order
    assert os.path.getsize(pb) == os.path.getsize(os.path.join(SURVEY, "global", "global_assignments.pb"))


def test_default_add_mode_matches_reference(global_pb, tmp_path):
    """usher -i g.pb -v new_samples.vcf -u : 1 2 / 1 1 / 0 1 / 1 1 / 0 1, sequential insertion."""
    pb, _ = global_pb
    rc = run_usher(["-i", pb, "-v", os.path.join(FIX, "new_samples.vcf"), "-u", "-d", str(tmp_path)])
    assert rc == 0
    for name in ("placement_stats.tsv", "mutation-paths.txt", "uncondensed-final-tree.nh"):
        assert _read(str(tmp_path / name)) == _read(os.path.join(SURVEY, "global", "out2", name)), name


def test_no_add_mode_matches_reference(global_pb, tmp_path):
    pb, _ = global_pb
    rc = run_usher(["-i", pb, "-v", os.path.join(FIX, "new_samples.vcf"), "-n", "-d", str(tmp_path)])
    assert rc == 0
    assert _read(str(tmp_path / "placement_stats.tsv")) == _read(os.path.join(SURVEY, "global", "out3", "placement_stats.tsv"))
    assert _read(str(tmp_path / "final-tree.nh")) == _read(os.path.join(SURVEY, "global", "out3", "final-tree.nh"))


def test_per_node_scores_file_matches_reference(global_pb, tmp_path):
    pb, _ = global_pb
    rc = run_usher(["-i", pb, "-v", os.path.join(FIX, "new_samples.vcf"), "-p", "-d", str(tmp_path)])
    assert rc == 0
    assert _read(str(tmp_path / "parsimony-scores.tsv")) == _read(os.path.join(SURVEY, "global", "out4", "parsimony-scores.tsv.gz"))
    assert _read(str(tmp_path / "current-tree.nh")) == _read(os.path.join(SURVEY, "global", "out4", "current-tree.nh"))


def test_baseline_config0_tree_plus_new_samples(tmp_path):
    """BASELINE config 0, literally: usher -t test/global_phylo.nh -v test/new_samples.vcf."""
    rc = run_usher(["-t", os.path.join(FIX, "global_phylo.nh"), "-v", os.path.join(FIX, "new_samples.vcf"), "-d", str(tmp_path)])
    assert rc == 0
    for name in ("placement_stats.tsv", "mutation-paths.txt", "final-tree.nh"):
        assert _read(str(tmp_path / name)) == _read(os.path.join(SURVEY, "global", "out8", name)), name


def test_branchlen2_known_answer(tmp_path):
    pb = str(tmp_path / "t.pb")
    rc = run_usher(["-t", os.path.join(FIX, "testBranchLen2.nwk"), "-v", os.path.join(FIX, "testBranchLen2.vcf"), "-o", pb, "-l", "-d", str(tmp_path)])
    assert rc == 0
    assert _read(str(tmp_path / "final-tree.nh")) == _read(os.path.join(SURVEY, "branchlen2", "final-tree.nh"))
    assert open(pb, "rb").read() == open(os.path.join(SURVEY, "branchlen2", "tbl2.pb"), "rb").read()


def test_syn_no_add_stats_with_imputed_mutations(tmp_path):
    rc = run_usher(["-i", os.path.join(SURVEY, "syn", "tree.pb"), "-v", os.path.join(SURVEY, "syn", "query.vcf"), "-n", "-d", str(tmp_path)])
    assert rc == 0
    assert _read(str(tmp_path / "placement_stats.tsv")) == _read(os.path.join(SURVEY, "syn", "o3", "placement_stats.tsv"))


def test_pb_round_trip_and_gz(tmp_path):
    src = os.path.join(SURVEY, "syn", "tree.pb")
    out = str(tmp_path / "o.pb.gz")
    rc = run_usher(["-i", src, "-v", os.path.join(SURVEY, "syn", "query.vcf"), "-n", "-o", out, "-d", str(tmp_path)])
    assert rc == 0
    a, b = _pb_semantic(src), _pb_semantic(out)
    assert a[0] == b[0] and a[1] == b[1]


def test_cli_flag_surface(tmp_path):
    assert run_usher(["--version"]) == 0
    assert run_usher(["--help"]) == 0
    assert run_usher(["-i", "x.pb"]) == 1                       # --vcf is required
    assert run_usher(["-v", "x.vcf"]) == 1                      # no tree / MAT
    assert run_usher(["-i", os.path.join(SURVEY, "syn", "tree.pb"), "-v", os.path.join(SURVEY, "syn", "query.vcf"), "-r", "-d", str(tmp_path)]) == 1
    assert run_usher(["-i", os.path.join(SURVEY, "syn", "tree.pb"), "-v", os.path.join(SURVEY, "syn", "query.vcf"), "-p", "-M", "2", "-d", str(tmp_path)]) == 1


def test_vcf_cell_quirks_match_the_reference_reader(tmp_path):
    """read_vcf quirks (SURVEY 7 'quirk fidelity'): 'V' falls through to N, lower case only for
    acgtn, ALT uses its first character, GT cells go through stoi ('1:x' -> 1, '0/1' -> 0), '.'
    is missing.  The C++ reader must produce exactly what the python restatement of
    mutation_annotated_tree.cpp:2180-2277 produces: compare the -n placement statistics."""
    import numpy as np
    from oracle import capi
    pb = os.path.join(SURVEY, "syn", "tree.pb")
    T = refio.load_mutation_annotated_tree(pb)
    ref_at = {}
    for n in T.depth_first_expansion():
        for m in n.mutations:
            if m.position >= 0:
                ref_at[m.position] = refio.get_nuc(m.ref_nuc)
    sites = sorted(ref_at)[:60]
    alts = ["V", "a", "c", "g", "t", "n", "r", "ACGT", "N", "R", "Y", "K", "TTT", "H", "."]
    cells = ["0", "1", "1:x", "0/1", "1/1", ".", "2", "2:0", "1|0", "x"]
    rng = np.random.default_rng(4)
    names = ["E%d" % i for i in range(12)]
    vcf = str(tmp_path / "edge.vcf")
    with open(vcf, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
        for p in sites:
            a1, a2 = rng.choice(alts, 2, replace=False)
            row = [str(rng.choice(cells)) for _ in names]
            f.write("chr\t%d\t.\t%s\t%s,%s\t.\t.\t.\tGT\t%s\n" % (p, ref_at[p], a1, a2, "\t".join(row)))
    rc = run_usher(["-i", pb, "-v", vcf, "-n", "-d", str(tmp_path)])
    assert rc == 0
    got = _read(str(tmp_path / "placement_stats.tsv")).splitlines()
    samples = refio.read_vcf(T, vcf)
    ot = capi.OracleTree(refio.tree_to_bfs_arrays(T))
    assert len(got) == len(samples) == len(names)
    for line, s in zip(got, samples):
        sa = refio.sample_to_arrays(s)
        r = ot.place(sa)
        v = ot.node_vecs(sa, r["best_j"])
        imputed = ";".join("%d:%s" % (p, refio.get_nuc(m)) for (p, _, _, m) in v["imputed"])
        assert line == "%s\t%d\t%d\t%s" % (s.name, r["best"], r["num_best"], imputed)


def _evolve_vcf(rng, n_leaves, n_sites, n_new, path_tree, path_old, path_new):
    """A random binary tree with genotypes evolved along it (existing samples) and new samples
    derived from random leaves (a few substitutions, some N and IUPAC cells)."""
    import numpy as np
    nuc = "ACGT"
    ref = rng.integers(0, 4, n_sites)
    pos = np.sort(rng.choice(np.arange(1, 30000), n_sites, replace=False))
    nodes = [("S%d" % i, None, ref.copy()) for i in range(1)]
    # grow by splitting a random leaf
    leaves = [{"name": None, "g": ref.copy(), "nwk": None}]
    tree = {"children": [], "g": ref.copy()}
    tips = [tree]
    while len(tips) < n_leaves:
        t = tips.pop(int(rng.integers(0, len(tips))))
        for _ in range(2):
            g = t["g"].copy()
            for s in rng.choice(n_sites, int(rng.integers(0, 3)), replace=False):
                g[s] = rng.integers(0, 4)
            c = {"children": [], "g": g}
            t["children"].append(c)
            tips.append(c)
    for i, t in enumerate(tips):
        t["name"] = "S%d" % i

    def nwk(t):
        return t["name"] if not t["children"] else "(" + ",".join(nwk(c) for c in t["children"]) + ")"
    with open(path_tree, "w") as f:
        f.write(nwk(tree) + ";\n")

    def write(path, names, G, codes=None):
        with open(path, "w") as f:
            f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
            for s in range(n_sites):
                alts = [a for a in range(4) if a != ref[s]]
                cells = []
                for k in range(len(names)):
                    c = codes[k][s] if codes is not None and codes[k][s] is not None else None
                    if c == "N":
                        cells.append(".")
                    elif c is not None:
                        cells.append(c)
                    else:
                        cells.append("0" if G[k][s] == ref[s] else str(alts.index(G[k][s]) + 1))
                f.write("chr\t%d\t%s%d\t%s\t%s\t.\t.\t.\tGT\t%s\n" % (pos[s], nuc[ref[s]], pos[s], nuc[ref[s]],
                                                                     ",".join(nuc[a] for a in alts), "\t".join(cells)))
    write(path_old, [t["name"] for t in tips], [t["g"] for t in tips])
    new_g, new_codes = [], []
    for k in range(n_new):
        g = tips[int(rng.integers(0, len(tips)))]["g"].copy()
        for s in rng.choice(n_sites, int(rng.integers(0, 3)), replace=False):
            g[s] = rng.integers(0, 4)
        codes = [None] * n_sites
        for s in rng.choice(n_sites, int(rng.integers(0, 4)), replace=False):
            codes[s] = "N"
        new_g.append(g)
        new_codes.append(codes)
    write(path_new, ["NEW%d" % k for k in range(n_new)], new_g, new_codes)


@pytest.mark.parametrize("seed,n_leaves,n_new,max_touched", [(1, 60, 80, None), (2, 150, 120, None), (3, 30, 200, None),
                                                             (4, 80, 400, None), (5, 10, 150, "12"), (6, 200, 250, "40")])
def test_batched_add_mode_equals_per_sample_research(tmp_path, monkeypatch, seed, n_leaves, n_new, max_touched):
    """Add-mode places the remaining samples in batches on the tree as it was and re-derives each answer on
    the tree as it is now: untouched optimal nodes keep their cost, nodes created or rewritten since are
    evaluated on the host, and the winner is chosen with today's leaf counts and breadth-first order.  With
    USHER_AMD_MAX_TOUCHED=0 every sample is searched again on the updated tree, which is the reference's
    loop (usher_common.cpp:310-449).  Both must write the same files."""
    import numpy as np
    from tests.host_harness import OracleBackend
    rng = np.random.default_rng(seed)
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    _evolve_vcf(rng, n_leaves, 90, n_new, nh, old, new)
    pb = str(tmp_path / "base.pb")
    assert run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0
    outs, calls = {}, {}
    if seed % 2 == 0:   # split even these small trees' O(N) passes (tree -> arrays) across host threads
        monkeypatch.setenv("USHER_AMD_GRAIN", "1")
        monkeypatch.setenv("USHER_AMD_THREADS", "5")
    for mode, env in (("research", "0"), ("batched", max_touched)):
        if env is None:
            monkeypatch.delenv("USHER_AMD_MAX_TOUCHED", raising=False)
        else:
            monkeypatch.setenv("USHER_AMD_MAX_TOUCHED", env)
        d = tmp_path / mode
        d.mkdir()
        be = OracleBackend()
        assert run_usher(["-i", pb, "-v", new, "-u", "-o", str(d / "out.pb"), "-d", str(d)], backend=be) == 0
        outs[mode] = {n: _read(str(d / n)) for n in ("placement_stats.tsv", "mutation-paths.txt", "uncondensed-final-tree.nh")}
        outs[mode]["pb"] = _pb_semantic(str(d / "out.pb"))
        calls[mode] = be.calls
    assert outs["batched"] == outs["research"]
    assert calls["batched"] < calls["research"]


@pytest.mark.parametrize("seed,max_trees", [(5, 4), (6, 2), (7, 6)])
def test_multiple_placements_equal_the_restated_driver_loop(seed, max_trees, tmp_path):
    """--multiple-placements (usher_common.cpp:584-649): trees are copied through the newick round trip, later trees are
    searched again for every sample, tied nodes are taken in breadth-first order, the flag that decides sibling vs
    child placement is read from the per-node vector with the loop counter (:647).  No recorded reference output
    exists for this mode; the C++ front end (oracle backend) is compared with tests/usher_model.py, a python
    restatement of that loop on the oracle's own tree model."""
    import numpy as np
    from tests import host_harness, usher_model
    rng = np.random.default_rng(seed)
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    _evolve_vcf(rng, 60, 40, 14, nh, old, new)
    pb = str(tmp_path / "base.pb")
    assert host_harness.run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0
    d = tmp_path / "multi"
    d.mkdir()
    assert host_harness.run_usher(["-i", pb, "-v", new, "-M", str(max_trees), "-d", str(d)]) == 0
    T = refio.load_mutation_annotated_tree(pb)
    want = usher_model.run(T, refio.read_vcf(T, new), max_trees=max_trees)
    got = {n: open(str(d / n)).read() for n in sorted(os.listdir(str(d)))}
    assert sorted(got) == sorted(want)
    for name in want:
        assert got[name] == want[name], name
    assert len([n for n in want if n.startswith("final-tree")]) > 1          # several trees were really produced


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_collapse_input_and_output_trees(seed, tmp_path):
    """--collapse-tree (usher_common.cpp:120-148: collapse_tree + condense_leaves before placing, condensed-tree.nh) and
    --collapse-output-tree (:808-822) against the python restatement of move_node / remove_node / collapse_tree."""
    import numpy as np
    from tests import host_harness, usher_model
    rng = np.random.default_rng(seed)
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    _evolve_vcf(rng, 150, 30, 25, nh, old, new)      # few sites: many mutation-free branches and identical leaves
    pb = str(tmp_path / "base.pb")
    assert host_harness.run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0
    for flags in (["-c"], ["-C"], ["-c", "-C"]):
        d = tmp_path / ("o" + "".join(f.strip("-") for f in flags))
        d.mkdir()
        assert host_harness.run_usher(["-i", pb, "-v", new, "-d", str(d)] + flags) == 0
        T = refio.load_mutation_annotated_tree(pb)
        missing = refio.read_vcf(T, new)
        want = {}
        if "-c" in flags:
            usher_model.collapse_tree(T)
            usher_model.condense_leaves(T)
            want["condensed-tree.nh"] = refio.get_newick_string(T, T.root, True, True) + "\n"
        out = usher_model.run(T, missing)
        if "-C" in flags:
            usher_model.collapse_tree(T)
            out["final-tree.nh"] = refio.get_newick_string(T, T.root, True, True)
            out["mutation-paths.txt"] = usher_model._paths(T, [m.name for m in missing])
        want.update(out)
        got = {n: open(str(d / n)).read() for n in sorted(os.listdir(str(d)))}
        assert sorted(got) == sorted(want), flags
        for name in want:
            assert got[name] == want[name], (flags, name)


def _newick_leaves(text):
    import re
    return [t for t in re.split(r"[(),;]", re.sub(r":[-+0-9.eE]+", "", text)) if t and not re.fullmatch(r"node_\d+", t)]


def test_subtrees_around_new_samples(tmp_path):
    """--write-single-subtree / --write-subtrees-size (usher_common.cpp:973-1012): the leaf choice draws from the C
    library's rand() exactly as the reference does; get_subtree, rotate_for_display and the three files per subtree
    are checked against the python restatement.  (-k sorts candidate leaves by distance with std::sort, whose order
    among equal distances is unspecified, so for -k the chosen leaf sets are taken from the files and everything
    derived from them is recomputed; sizes and the coverage of the new samples are checked directly.)"""
    import numpy as np
    from tests import host_harness, usher_model
    rng = np.random.default_rng(21)
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    _evolve_vcf(rng, 120, 60, 9, nh, old, new)
    pb = str(tmp_path / "base.pb")
    assert host_harness.run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0
    d = tmp_path / "sub"
    d.mkdir()
    libc = usher_model._LibcRand()
    libc.srand(1)            # the state a fresh process starts with (the front end runs inside this process here)
    assert host_harness.run_usher(["-i", pb, "-v", new, "-d", str(d), "-K", "7", "-k", "4"]) == 0
    T = refio.load_mutation_annotated_tree(pb)
    missing = refio.read_vcf(T, new)
    usher_model.run(T, missing)
    names = [m.name for m in missing]
    usher_model.uncondense_leaves(T)
    got = {n: open(str(d / n)).read() for n in sorted(os.listdir(str(d)))}
    libc.srand(1)
    want = usher_model.single_subtree(T, names, 7, libc)
    for name in want:
        assert got[name] == want[name], name
    # -k 4: every placed sample appears in one of the subtree files the loop produced, each with at most 4 leaves
    subs = sorted(n for n in got if n.startswith("subtree-") and n.endswith(".nh"))
    assert subs
    shown = set()
    for f in subs:
        leaves = _newick_leaves(got[f])
        assert 2 <= len(leaves) <= 4
        shown.update(leaves)
        files = usher_model._subtree_files(T, usher_model.get_subtree(T, leaves), f[:-3])
        for name, text in files.items():
            assert got[name] == text, name
    assert all(n in shown for n in names if T.get_node(n) is not None)


@pytest.mark.parametrize("seed", [31, 32])
def test_out_of_order_and_duplicated_rows_follow_the_reference_scans(seed, tmp_path):
    """VCF rows that are not sorted by position, or repeat a position: the reference's scans are order-dependent there
    (usher_mapper.cpp:204-242, 393-445).  Such samples are searched on the host with the literal routine instead of
    being sorted first; -n, -p and the default add mode must equal the oracle's literal answers on the rows as given."""
    import numpy as np
    from tests import host_harness, usher_model
    rng = np.random.default_rng(seed)
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    _evolve_vcf(rng, 90, 50, 12, nh, old, new)
    pb = str(tmp_path / "base.pb")
    assert host_harness.run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0
    # shuffle the VCF's data lines and repeat some of them with another allele
    lines = open(new).read().splitlines()
    head = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if not l.startswith("#")]
    dup = []
    for l in body[::3]:
        w = l.split("\t")
        w[4] = "ACGT".replace(w[3], "")[int(rng.integers(0, 3))]
        w[9:] = [c if c in ("0", ".") else "1" for c in w[9:]]
        dup.append("\t".join(w))
    body = body + dup
    rng.shuffle(body)
    odd_vcf = str(tmp_path / "odd.vcf")
    open(odd_vcf, "w").write("\n".join(head + body) + "\n")
    T = refio.load_mutation_annotated_tree(pb)
    missing = refio.read_vcf(T, odd_vcf)
    assert any(any(b.position <= a.position for a, b in zip(m.mutations, m.mutations[1:])) for m in missing)
    arrays = refio.tree_to_bfs_arrays(T)
    from oracle import capi
    ot = capi.OracleTree(arrays)
    # -n
    d = tmp_path / "n"; d.mkdir()
    assert host_harness.run_usher(["-i", pb, "-v", odd_vcf, "-n", "-d", str(d)]) == 0
    want = ""
    for m in missing:
        r = ot.place(refio.sample_to_arrays(m))
        nv = ot.node_vecs(refio.sample_to_arrays(m), r["best_j"])
        want += "%s\t%d\t%d\t%s\n" % (m.name, r["best"], r["num_best"], ";".join("%d:%s" % (p, refio.get_nuc(mu)) for (p, rf, pa, mu) in nv["imputed"]))
    assert open(str(d / "placement_stats.tsv")).read() == want
    # -p: score column of every (sample, node)
    d = tmp_path / "p"; d.mkdir()
    assert host_harness.run_usher(["-i", pb, "-v", odd_vcf, "-p", "-d", str(d)]) == 0
    rows = [l.split("\t") for l in open(str(d / "parsimony-scores.tsv")).read().splitlines()[1:]]
    k = 0
    for m in missing:
        sc = ot.place(refio.sample_to_arrays(m), compute_scores=True)["scores"]
        for j in range(arrays["n"]):
            assert rows[k][0] == m.name and rows[k][1] == arrays["names"][j] and int(rows[k][2]) == int(sc[j]), (m.name, j)
            k += 1
    # default add mode against the restated loop (searches on the rows as given)
    d = tmp_path / "a"; d.mkdir()
    assert host_harness.run_usher(["-i", pb, "-v", odd_vcf, "-d", str(d)]) == 0
    T2 = refio.load_mutation_annotated_tree(pb)
    want = usher_model.run(T2, refio.read_vcf(T2, odd_vcf))
    for name in want:
        assert open(str(d / name)).read() == want[name], name


def test_samples_already_in_the_tree_are_ignored(tmp_path, capfd):
    """A VCF column named like a leaf of the loaded MAT is not a new sample: the reader warns and skips it
    (mutation_annotated_tree.cpp:2216-2226), and the run writes what it writes without that column."""
    import re
    pb = os.path.join(SURVEY, "syn", "tree.pb")
    vcf = os.path.join(SURVEY, "syn", "query.vcf")
    leaf = re.findall(r"[(,]([^(),:;]+)", _read(os.path.join(SURVEY, "syn", "tree.nh")))[3]
    lines = _read(vcf).splitlines()
    extra = str(tmp_path / "extra.vcf")
    with open(extra, "w") as f:
        for l in lines:
            if l.startswith("##"):
                f.write(l + "\n")
            elif l.startswith("#CHROM"):
                f.write(l + "\t" + leaf + "\n")
            else:
                f.write(l + "\t1\n")          # the column carries calls: they must not reach the placement
    outs = []
    for k, v in enumerate((vcf, extra)):
        d = tmp_path / ("o%d" % k)
        d.mkdir()
        assert run_usher(["-i", pb, "-v", v, "-u", "-d", str(d)]) == 0
        outs.append({n: _read(str(d / n)) for n in ("placement_stats.tsv", "mutation-paths.txt", "uncondensed-final-tree.nh")})
    assert outs[0] == outs[1]
    assert "already in the tree" in capfd.readouterr().err


# ---------------------------------------------------------------------------
# Round 4: the loaders on host threads (bulk newick, per-node mutation decode, VCF line blocks)
# ---------------------------------------------------------------------------

def _newick_digest(nwk, bulk):
    import ctypes as C
    L = C.CDLL(HOST_LIB)
    L.uh_newick_digest.restype = C.c_long
    L.uh_newick_digest.argtypes = [C.c_char_p, C.c_int, C.c_char_p, C.c_size_t]
    buf = C.create_string_buffer(1 << 22)
    n = L.uh_newick_digest(nwk.encode(), bulk, buf, len(buf))
    return n, buf.value.decode()


@pytest.mark.parametrize("threads", ["1", "5"])
def test_bulk_newick_parse_equals_the_general_routine(threads, monkeypatch):
    """tree_from_newick_bulk (items tokenised on the host threads, nodes in one block, name index by a parallel sort) against
    tree_from_newick (mutation_annotated_tree.cpp:415-508 restated): ids, parents, levels, branch lengths -- including the
    reference's quirks (a length that survives a ')' without a new ':', missing lengths = -1, internal labels ignored, junk
    inside a number skipped) -- children counts, the internal-node counter, and name lookups, on hand-made and random trees."""
    monkeypatch.setenv("USHER_AMD_THREADS", threads)
    monkeypatch.setenv("USHER_AMD_GRAIN", "1")
    cases = ["(A:1,B:2)r:3;", "((A,B),C);", "((A:0.5,B:2)x:1e-1,(C:3,D)y,E:7);", "(A:1,(B:2,(C:3,(D:4,E:5)))));"[:-2] + ";",
             "((((((A:1,B:2):3,C):4,D:5):6,E):7,F:8):9,G);", "(A:1x2,B:-3,C:+4.5E1);", "(A ,B\t:2 ,C:3);", "(L1:1,L2:2,(L3:0,L4:0,L5:0)node_7:2,L6)lab;"]
    rng = np.random.default_rng(17)

    def rand_tree(n_leaves):
        names = ["S%d" % i for i in range(n_leaves)]
        nodes = [n + (":%g" % rng.integers(0, 9) if rng.random() < 0.8 else "") for n in names]
        while len(nodes) > 1:
            k = int(rng.integers(2, min(6, len(nodes)) + 1))
            i = int(rng.integers(0, len(nodes) - k + 1))
            grp = "(" + ",".join(nodes[i:i + k]) + ")" + ("lbl" if rng.random() < 0.3 else "") + (":%g" % (rng.integers(0, 50) / 4) if rng.random() < 0.7 else "")
            nodes[i:i + k] = [grp]
        return nodes[0] + ";"
    cases += [rand_tree(int(n)) for n in (2, 3, 17, 200, 3000)]
    for nwk in cases:
        a, b = _newick_digest(nwk, 0), _newick_digest(nwk, 1)
        assert a[0] > 0 and a == b, nwk[:200]
        assert "INDEX MISMATCH" not in b[1]
    # malformed input is refused by both; a duplicated leaf name too
    for bad in ("((A,B);", "(A,B));", "(A,B,A);"):
        assert _newick_digest(bad, 0)[0] == -1 and _newick_digest(bad, 1)[0] == -1, bad
    # a branch length that is no number (std::stof would throw -- on a worker thread: std::terminate) is an error, not a
    # crash, in both routines; one longer than the bulk routine's buffer is parsed in full (by the general routine)
    for bad in ("(A:-,B:1);", "(A:1,B:e);", "((A:1,B:2):+,C);", "(A:1e99,B);"):
        assert _newick_digest(bad, 0)[0] == -1 and _newick_digest(bad, 1)[0] == -1, bad
    long_len = "(A:1." + "0" * 80 + "5,B:3);"
    a, b = _newick_digest(long_len, 0), _newick_digest(long_len, 1)
    assert a[0] > 0 and a == b


@pytest.mark.parametrize("fixture", ["global", "syn", "big"])
def test_threaded_loaders_write_the_same_files(fixture, tmp_path, monkeypatch):
    """parsimony.proto and VCF read on several host threads (bulk newick, per-node mutation decode, VCF line blocks merged in
    file order: mutation_annotated_tree.cpp:556-612, :2108-2279) give byte-identical outputs to the one-thread general path --
    placement stats, trees, and the MAT saved again -- on the recorded fixtures (plain and .gz)."""
    if fixture == "global":
        pb, vcf = os.path.join(SURVEY, "global", "global_assignments.pb"), os.path.join(FIX, "new_samples.vcf")
    elif fixture == "syn":
        pb, vcf = os.path.join(SURVEY, "syn", "tree.pb"), os.path.join(SURVEY, "syn", "query.vcf")
    else:
        pb, vcf = os.path.join(SURVEY, "big", "tree.pb.gz"), os.path.join(SURVEY, "big", "query.vcf")
    outs = []
    for k, env in enumerate(({"USHER_AMD_THREADS": "1"}, {"USHER_AMD_THREADS": "7", "USHER_AMD_GRAIN": "1"}, {"USHER_AMD_THREADS": "3", "USHER_AMD_GRAIN": "2"})):
        for key in ("USHER_AMD_THREADS", "USHER_AMD_GRAIN"):
            monkeypatch.delenv(key, raising=False)
        for key, v in env.items():
            monkeypatch.setenv(key, v)
        d = tmp_path / ("o%d" % k)
        d.mkdir()
        assert run_usher(["-i", pb, "-v", vcf, "-d", str(d), "-u", "-o", str(d / "out.pb")]) == 0
        outs.append({n: open(str(d / n), "rb").read() for n in sorted(os.listdir(str(d)))})
    assert len(outs[0]) >= 4
    for o in outs[1:]:
        assert o.keys() == outs[0].keys()
        for n in o:
            assert o[n] == outs[0][n], n


@pytest.mark.parametrize("seed,n_leaves,n_new,batch,rnd", [(11, 60, 90, "32", "8"), (12, 150, 160, "64", "5"), (13, 30, 220, "4096", "64"),
                                                           (14, 12, 130, "16", "1"), (15, 200, 260, "50", "16")])
def test_add_mode_with_the_edits_on_the_device_equals_per_sample_research(tmp_path, monkeypatch, seed, n_leaves, n_new, batch, rnd):
    """The add mode that keeps the flattened tree and reports the edits (Backend::update / touched_*, include/usher_amd.h "add mode"):
    one flattening for the whole run, batches and rounds of several sizes -- including rounds of one sample and batches shorter than
    the list of samples -- against the reference's loop (a full search of the current tree per sample, USHER_AMD_MAX_TOUCHED=0) and
    against tests/usher_model.py (the oracle searching, the restated tree edits).  The backend here restates the device library's
    semantics in numpy (running minima that go stale when their holders are rewritten, excluded nodes), so every "ask again" path of
    the driver is exercised on CPU; on the GPU box the same driver runs on the HIP library (tests/test_add_mode_gpu.py)."""
    import numpy as np
    from tests import usher_model
    from tests.host_harness import OracleBackend
    rng = np.random.default_rng(seed)
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    _evolve_vcf(rng, n_leaves, 70, n_new, nh, old, new)
    pb = str(tmp_path / "base.pb")
    assert run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0
    outs = {}
    for mode in ("research", "device"):
        for k in ("USHER_AMD_MAX_TOUCHED", "USHER_AMD_BATCH", "USHER_AMD_ROUND"):
            monkeypatch.delenv(k, raising=False)
        if mode == "research":
            monkeypatch.setenv("USHER_AMD_MAX_TOUCHED", "0")
        else:
            monkeypatch.setenv("USHER_AMD_BATCH", batch)
            monkeypatch.setenv("USHER_AMD_ROUND", rnd)
        d = tmp_path / mode
        d.mkdir()
        be = OracleBackend(add_mode=(mode == "device"))
        assert run_usher(["-i", pb, "-v", new, "-u", "-o", str(d / "out.pb"), "-d", str(d)], backend=be) == 0
        outs[mode] = {n: _read(str(d / n)) for n in ("placement_stats.tsv", "mutation-paths.txt", "uncondensed-final-tree.nh")}
        outs[mode]["pb"] = _pb_semantic(str(d / "out.pb"))
        if mode == "device":
            assert be.stat["update"] > 0 and be.stat["open"] >= (n_new + int(batch) - 1) // int(batch) and be.calls >= be.stat["open"]
            assert len(be.records) >= n_new        # every insertion left at least its leaf
    assert outs["device"] == outs["research"]
    # ... and the restated driver loop with the oracle doing every search
    T = refio.load_mutation_annotated_tree(pb)
    want = usher_model.run(T, refio.read_vcf(T, new))
    assert outs["device"]["placement_stats.tsv"] == want["placement_stats.tsv"]
    assert outs["device"]["mutation-paths.txt"] == want["mutation-paths.txt"]
    assert sum(1 for l in want["placement_stats.tsv"].splitlines() if l.split("\t")[2] != "1") > 5      # ties were exercised


@pytest.mark.parametrize("seed", [21, 22])
def test_add_mode_on_the_device_with_host_searched_samples_in_front_of_the_first_batch(tmp_path, monkeypatch, seed):
    """Samples whose rows are out of order are searched on the host (the reference's scans depend on the order); when they come in
    front of the first batch, their insertions are part of the one flattening -- not edits of it -- and the leaf counts used by the
    tie-break must not count them twice.  Mixed list (out-of-order and ordinary samples) against the per-sample re-search and the model."""
    import numpy as np
    from tests import usher_model
    from tests.host_harness import OracleBackend
    rng = np.random.default_rng(seed)
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    _evolve_vcf(rng, 80, 60, 70, nh, old, new)
    pb = str(tmp_path / "base.pb")
    assert run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0
    lines = open(new).read().splitlines()
    head = [l for l in lines if l.startswith("#")]
    body = [l for l in lines if not l.startswith("#")]
    # move late rows that the first samples carry to the top: those samples become out of order, most others stay ordinary
    moved = [l for l in body[len(body) // 2:] if l.split("\t")[9] != "0"][:3]
    assert moved
    body = moved + [l for l in body if l not in moved]
    odd_vcf = str(tmp_path / "odd.vcf")
    open(odd_vcf, "w").write("\n".join(head + body) + "\n")
    T = refio.load_mutation_annotated_tree(pb)
    missing = refio.read_vcf(T, odd_vcf)
    is_odd = [any(b.position <= a.position for a, b in zip(m.mutations, m.mutations[1:])) for m in missing]
    assert is_odd[0] and not all(is_odd) and sum(is_odd) >= 3
    outs = {}
    for mode in ("research", "device"):
        for k in ("USHER_AMD_MAX_TOUCHED", "USHER_AMD_BATCH", "USHER_AMD_ROUND"):
            monkeypatch.delenv(k, raising=False)
        if mode == "research":
            monkeypatch.setenv("USHER_AMD_MAX_TOUCHED", "0")
        else:
            monkeypatch.setenv("USHER_AMD_BATCH", "16")
            monkeypatch.setenv("USHER_AMD_ROUND", "4")
        d = tmp_path / mode
        d.mkdir()
        be = OracleBackend(add_mode=(mode == "device"))
        assert run_usher(["-i", pb, "-v", odd_vcf, "-u", "-d", str(d)], backend=be) == 0
        outs[mode] = {n: _read(str(d / n)) for n in ("placement_stats.tsv", "mutation-paths.txt", "uncondensed-final-tree.nh")}
    assert outs["device"] == outs["research"]
    T2 = refio.load_mutation_annotated_tree(pb)
    want = usher_model.run(T2, refio.read_vcf(T2, odd_vcf))
    for n in ("placement_stats.tsv", "mutation-paths.txt"):
        assert outs["device"][n] == want[n], n


def test_add_mode_on_the_device_counts_leaves_added_in_front_of_the_flattening_once(tmp_path, monkeypatch):
    """Targeted: A = {a1, a2} (mutation m1) and B = {b1, b2, b3} (m2).  The first sample (its last two rows out of order -> searched
    and inserted on the host before the one flattening) lands below A: A and B now hold 3 leaves each.  The second sample {m1, m2}
    ties between A and B, and the tie-break compares the sizes of the two subtrees (usher_mapper.cpp:455-494): counting the first
    sample's leaf both in the flattening and in the driver's "leaves added since" table makes A look larger than B and flips the
    choice.  Outputs equal the per-sample re-search and tests/usher_model.py."""
    from tests import usher_model
    from tests.host_harness import OracleBackend
    site = {"m1": 100, "m2": 110, "x1": 120, "x2": 130, "y1": 140, "y2": 150, "y3": 160, "z": 170, "o": 180, "w": 190}
    def write(path, names, alts_of, order):
        with open(path, "w") as f:
            f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
            for k in order:
                f.write("chr\t%d\t.\tA\tC\t.\t.\t.\tGT\t%s\n" % (site[k], "\t".join("1" if k in alts_of[n] else "0" for n in names)))
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    open(nh, "w").write("((a1,a2),(b1,b2,b3),OUT);\n")
    in_order = sorted(site, key=lambda k: site[k])
    write(old, ["a1", "a2", "b1", "b2", "b3", "OUT"], {"a1": {"m1", "x1"}, "a2": {"m1", "x2"}, "b1": {"m2", "y1"}, "b2": {"m2", "y2"},
                                                        "b3": {"m2", "y3"}, "OUT": {"o"}}, in_order)
    write(new, ["O", "S"], {"O": {"m1", "z", "w"}, "S": {"m1", "m2"}}, [k for k in in_order if k not in "zw"] + ["w", "z"])
    pb = str(tmp_path / "base.pb")
    assert run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0
    outs = {}
    for mode in ("research", "device"):
        for k in ("USHER_AMD_MAX_TOUCHED", "USHER_AMD_BATCH", "USHER_AMD_ROUND"):
            monkeypatch.delenv(k, raising=False)
        if mode == "research":
            monkeypatch.setenv("USHER_AMD_MAX_TOUCHED", "0")
        d = tmp_path / mode
        d.mkdir()
        be = OracleBackend(add_mode=(mode == "device"))
        assert run_usher(["-i", pb, "-v", new, "-u", "-d", str(d)], backend=be) == 0
        outs[mode] = {n: _read(str(d / n)) for n in ("placement_stats.tsv", "mutation-paths.txt", "uncondensed-final-tree.nh")}
        if mode == "device":
            assert be.stat["open"] == 1            # one batch: the second sample; the first went through the host search
    assert outs["device"] == outs["research"]
    T = refio.load_mutation_annotated_tree(pb)
    want = usher_model.run(T, refio.read_vcf(T, new))
    assert want["placement_stats.tsv"].splitlines()[1].split("\t")[2] == "2"        # the tie was there
    assert outs["device"]["placement_stats.tsv"] == want["placement_stats.tsv"]
    assert outs["device"]["mutation-paths.txt"] == want["mutation-paths.txt"]


@pytest.mark.parametrize("tcap", [None, "1"])
def test_add_mode_asks_again_when_the_holders_of_a_minimum_were_rewritten(tmp_path, monkeypatch, capfd, tcap):
    """The three "ask again" paths of the device add mode, on a tree built to need them.  K leaves with branch {a, b, c} each.
    Batch 1: t1 = {a, b} + a private site -> sibling of the leaf: the leaf is rewritten (excluded from the flattened tree's search).
    Batch 2, round 1: t2 = {a} + private -> splits the new internal node {a, b} into {a} above {b}; round 2: t3 = {b}, whose
    record minimum was held by the node t2 rewrote, and everything that replaced it costs one more -> the record results of t3
    are stale (asked again on the device).  t4 = {c} in batch 1 behind t1: its only optimal flattened node is the leaf t1 rewrote,
    and the rewritten leaf {c} under {a, b} costs more -> the flattened tree is searched again for t4.  With lists of one entry
    (USHER_AMD_TCAP=1) every tie among records is a truncated list (all records evaluated on the host).  Outputs equal the
    per-sample research and tests/usher_model.py."""
    from tests import usher_model
    from tests.host_harness import OracleBackend
    K = 12
    nuc = "ACGT"
    sites = [100 + 10 * i for i in range(3 * K + 3 * K + 2)]     # 3 per group, then private sites, then 2 for the outgroup
    def write(path, names, alts_of):
        with open(path, "w") as f:
            f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join(names) + "\n")
            for si, p in enumerate(sites):
                f.write("chr\t%d\t.\tA\tC\t.\t.\t.\tGT\t%s\n" % (p, "\t".join("1" if si in alts_of[n] else "0" for n in names)))
    old_names = ["L%d" % k for k in range(K)] + ["OUT1", "OUT2"]
    old_alts = {"L%d" % k: {3 * k, 3 * k + 1, 3 * k + 2} for k in range(K)}
    old_alts["OUT1"] = {6 * K}; old_alts["OUT2"] = {6 * K + 1}
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    with open(nh, "w") as f:
        f.write("(" + ",".join(old_names) + ");\n")
    write(old, old_names, old_alts)
    new_names, new_alts = [], {}
    for k in range(K):            # batch 1 (2K samples): t1_k then t4_k
        new_names.append("T1_%d" % k); new_alts[new_names[-1]] = {3 * k, 3 * k + 1, 3 * K + 3 * k}
    for k in range(K):
        new_names.append("T4_%d" % k); new_alts[new_names[-1]] = {3 * k + 2}
    for k in range(K):            # batch 2: t2_k (round 1 ..), then t3_k
        new_names.append("T2_%d" % k); new_alts[new_names[-1]] = {3 * k, 3 * K + 3 * k + 1}
    for k in range(K):
        new_names.append("T3_%d" % k); new_alts[new_names[-1]] = {3 * k + 1}
    for k in range(0, K, 3):      # t6 = {a, b, c} + t1's private site: the new leaf of t1 and the rewritten leaf {c} tie -- two records
        new_names.append("T6_%d" % k); new_alts[new_names[-1]] = {3 * k, 3 * k + 1, 3 * k + 2, 3 * K + 3 * k}
    write(new, new_names, new_alts)
    pb = str(tmp_path / "base.pb")
    assert run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0
    outs = {}
    for mode in ("research", "device"):
        for kk in ("USHER_AMD_MAX_TOUCHED", "USHER_AMD_BATCH", "USHER_AMD_ROUND", "USHER_AMD_TCAP", "USHER_AMD_PROFILE"):
            monkeypatch.delenv(kk, raising=False)
        if mode == "research":
            monkeypatch.setenv("USHER_AMD_MAX_TOUCHED", "0")
        else:
            monkeypatch.setenv("USHER_AMD_BATCH", str(2 * K))
            monkeypatch.setenv("USHER_AMD_ROUND", str(K // 2))
            monkeypatch.setenv("USHER_AMD_PROFILE", "1")
            if tcap:
                monkeypatch.setenv("USHER_AMD_TCAP", tcap)
        d = tmp_path / mode
        d.mkdir()
        be = OracleBackend(add_mode=(mode == "device"))
        capfd.readouterr()
        assert run_usher(["-i", pb, "-v", new, "-u", "-d", str(d)], backend=be) == 0
        err = capfd.readouterr().err
        outs[mode] = {n: _read(str(d / n)) for n in ("placement_stats.tsv", "mutation-paths.txt", "uncondensed-final-tree.nh")}
        if mode == "device":
            line = [l for l in err.splitlines() if "add mode on the device" in l][0]
            again = [int(x) for x in line.split("asked again:")[1].replace(",", " ").split() if x.isdigit()]
            assert again[0] >= K // 2 and again[1] >= 1, line            # the flattened tree and the records were asked again
            if tcap:
                assert again[2] >= 1, line                                # ... and truncated lists went to the host
            assert be.stat["rescore"] == again[1]
    assert outs["device"] == outs["research"]
    T = refio.load_mutation_annotated_tree(pb)
    want = usher_model.run(T, refio.read_vcf(T, new))
    assert outs["device"]["placement_stats.tsv"] == want["placement_stats.tsv"]
    assert outs["device"]["mutation-paths.txt"] == want["mutation-paths.txt"]


@pytest.mark.parametrize("seed", [31, 32, 33])
def test_newick_written_on_threads_equals_the_loop(seed, tmp_path, monkeypatch):
    """final-tree.nh (internal names, branch length = number of mutations) and the newick inside the saved MAT (no internal names)
    written by the threaded writer (lengths bottom-up, offsets top-down, every node writes its own pieces) are byte-identical to the
    paren-by-paren loop that restates mutation_annotated_tree.cpp:215-346, after a run that added samples."""
    import numpy as np
    from tests.host_harness import OracleBackend
    rng = np.random.default_rng(seed)
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    _evolve_vcf(rng, 40 + 30 * (seed % 3), 60, 25, nh, old, new)
    pb = str(tmp_path / "base.pb")
    assert run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0
    outs = []
    for env in ({"USHER_AMD_NEWICK_LOOP": "1"}, {"USHER_AMD_GRAIN": "1", "USHER_AMD_THREADS": "6"}, {"USHER_AMD_GRAIN": "3", "USHER_AMD_THREADS": "2"}):
        for k in ("USHER_AMD_NEWICK_LOOP", "USHER_AMD_GRAIN", "USHER_AMD_THREADS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        d = tmp_path / ("o%d" % len(outs))
        d.mkdir()
        assert run_usher(["-i", pb, "-v", new, "-o", str(d / "out.pb"), "-d", str(d)], backend=OracleBackend()) == 0
        outs.append({n: open(str(d / n), "rb").read() for n in ("final-tree.nh", "out.pb", "mutation-paths.txt")})
    assert outs[0]["final-tree.nh"].count(b"(") > 30
    assert outs[1] == outs[0] and outs[2] == outs[0]


def test_large_pb_gz_is_inflated_in_a_pipeline_and_loads_the_same_tree(tmp_path, monkeypatch):
    """Round 5: a large .pb.gz (the public MAT ships compressed, mutation_annotated_tree.cpp:522-547) is inflated on a thread of its own
    while the field scan follows it and the newick string is parsed: the tree loaded equals the one loaded from the plain file -- also
    for a file of several gzip members, whose trailer is no size hint (the loader starts over in one piece), with the pipeline switched
    off, and on one host thread."""
    import ctypes as C
    import gzip
    import shutil
    from tools.time_load import host_lib, ptr, write_workload
    from usher_amd import synth as gsynth
    L = host_lib()
    L.uh_pb_to_arrays.argtypes = [C.c_char_p] + [C.c_void_p] * 8
    st = gsynth.SynthTree(150_000, n_sites=1500, seed=11)
    pb = str(tmp_path / "t.pb")
    write_workload(L, st, None, 0, pb, None)
    raw = open(pb, "rb").read()
    assert len(raw) > (2 << 20)
    with gzip.open(pb + ".gz", "wb", compresslevel=1) as f:
        f.write(raw)
    with open(str(tmp_path / "two.pb.gz"), "wb") as f:                # two members: ISIZE names the second one's length only
        cut = len(raw) - (1 << 20) - 12345
        f.write(gzip.compress(raw[:cut], 1)); f.write(gzip.compress(raw[cut:], 1))

    def arrays(path):
        counts = (C.c_uint64 * 2)()
        assert L.uh_pb_to_arrays(path.encode(), counts, None, None, None, None, None, None, None) == 0
        n, m = int(counts[0]), int(counts[1])
        a = [np.zeros(n, np.int64), np.zeros(n + 1, np.int64), np.zeros(m, np.int32), np.zeros(m, np.int8), np.zeros(m, np.int8), np.zeros(m, np.int8)]
        assert L.uh_pb_to_arrays(path.encode(), counts, *[ptr(x) for x in a], None) == 0
        return a
    want = arrays(pb)
    assert len(want[0]) == int(st.arrays["n"])
    for env in ({}, {"USHER_AMD_THREADS": "1"}, {"USHER_AMD_NO_GZ_PIPELINE": "1"}):
        for k in ("USHER_AMD_THREADS", "USHER_AMD_NO_GZ_PIPELINE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        for name in ("t.pb.gz", "two.pb.gz"):
            got = arrays(str(tmp_path / name))
            assert all((a == b).all() for a, b in zip(want, got)), (env, name)
    # a truncated file is an error, not a hang
    with open(str(tmp_path / "cut.pb.gz"), "wb") as f:
        z = open(pb + ".gz", "rb").read()
        f.write(z[:len(z) // 2])
    counts = (C.c_uint64 * 2)()
    assert L.uh_pb_to_arrays(str(tmp_path / "cut.pb.gz").encode(), counts, None, None, None, None, None, None, None) != 0


@pytest.mark.parametrize("flags", [["-n"], ["-n", "-u"], [], ["-s"]])
def test_tree_handed_over_under_the_vcf_read_and_final_tree_written_ahead(flags, tmp_path, monkeypatch):
    """Round 5 (end-to-end latency of the drop-in): the front end calls Backend::warm first thing and hands the loaded tree to
    Backend::prepare on a thread of its own while the VCF is read; with -n the final tree's text is produced while the samples are
    placed.  Every output file equals the run with both switched off (USHER_AMD_NO_PREBUILD), and the backend saw one `prepare`
    of version 1 and no second flattening of the same tree."""
    from tests import host_harness
    pb, vcf = os.path.join(SURVEY, "syn", "tree.pb"), os.path.join(SURVEY, "syn", "query.vcf")
    outs = []
    for k, off in enumerate((False, True)):
        monkeypatch.delenv("USHER_AMD_NO_PREBUILD", raising=False)
        if off:
            monkeypatch.setenv("USHER_AMD_NO_PREBUILD", "1")
        d = tmp_path / ("o%d" % k)
        d.mkdir()
        be = host_harness.OracleBackend()
        assert host_harness.run_usher(["-i", pb, "-v", vcf, "-d", str(d)] + flags, be) == 0
        assert be.warmed == 1
        assert be.prepared == ([] if off else [1])
        outs.append({n: _read(str(d / n)) for n in sorted(os.listdir(str(d)))})
    assert outs[0] == outs[1] and "placement_stats.tsv" in outs[0]


@pytest.mark.parametrize("pad_to_page", [False, True])
def test_vcf_mapped_instead_of_read_gives_the_same_files(pad_to_page, tmp_path, monkeypatch):
    """Round 6: a large VCF is mapped, not copied (mat.cpp view_file; 1.4 s of a 100 000-sample add-mode run was the zero fill of the
    read buffer).  With USHER_AMD_MMAP_MIN=1 the small fixture takes that path: same output files as the read path -- also when the file's
    size is a whole number of pages (the byte behind the last one must still be readable: one zero page is mapped behind the file)."""
    src = os.path.join(SURVEY, "syn", "query.vcf")
    vcf = str(tmp_path / "q.vcf")
    data = open(src, "rb").read()
    if pad_to_page:   # comment lines in front of the header are skipped by the reader (and by the reference's)
        page = os.sysconf("SC_PAGE_SIZE")
        room = (-len(data)) % page
        if room < 2:
            room += page
        data = b"#" + b"x" * (room - 2) + b"\n" + data
        assert len(data) % page == 0
    open(vcf, "wb").write(data)
    outs = []
    for env in (None, "1"):
        if env is None:
            monkeypatch.setenv("USHER_AMD_NO_MMAP", "1")
        else:
            monkeypatch.delenv("USHER_AMD_NO_MMAP", raising=False)
            monkeypatch.setenv("USHER_AMD_MMAP_MIN", env)
        d = tmp_path / ("o%s" % env)
        d.mkdir()
        assert run_usher(["-i", os.path.join(SURVEY, "syn", "tree.pb"), "-v", vcf, "-d", str(d)]) == 0
        outs.append({n: _read(str(d / n)) for n in sorted(os.listdir(str(d)))})
    assert outs[0] == outs[1] and "placement_stats.tsv" in outs[0]
