"""CPU checks of the product's host flattening (libusher_amd.so, no GPU): the
DFS record stream interpreted by tests/stream_interp.py must reproduce the
oracle for every chunking / grouping, and the C-ABI must export every symbol
include/usher_amd.h declares."""
import os
import re

import numpy as np
import pytest

from oracle import capi, refio
from tests import stream_interp, synth
from usher_amd import FlatTreeView, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def test_capi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "usher_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(ugp_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), "libusher_amd.so does not export %s" % name
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))


def _check(arrays, samples, chunk_nodes, n_groups, scores=True):
    flat = FlatTreeView(arrays, chunk_nodes=chunk_nodes)
    ot = capi.OracleTree(arrays)
    for s in samples:
        want = ot.place(s, compute_scores=scores)
        got = stream_interp.place(flat, s, n_groups=n_groups, want_scores=scores)
        for k in ("best", "num_best", "best_j", "has_unique"):
            assert got[k] == want[k], (s["name"], k, got[k], want[k])
        if scores:
            assert got["scores"].tolist() == want["scores"].tolist()
        for ub in (None, 0x7F7F, want["best"]):   # no pruning / the kernel's start bound / the tightest legal bound
            got8 = stream_interp.place8(flat, s, n_groups=n_groups, prune_ub=ub)
            for k in ("best", "num_best", "best_j", "has_unique"):
                assert got8[k] == want[k], ("packed", ub, s["name"], k, got8[k], want[k])
    return flat


@pytest.mark.parametrize("chunk_nodes,n_groups", [(0, 1), (16, 1), (16, 7), (5, 1000), (64, 3)])
def test_stream_matches_oracle_on_global_fixture(chunk_nodes, n_groups):
    T = refio.load_mutation_annotated_tree(os.path.join(GOLD, "survey_ref", "global", "global_assignments.pb"))
    samples = [refio.sample_to_arrays(s) for s in refio.read_vcf(T, os.path.join(GOLD, "ref_fixtures", "new_samples.vcf"))]
    flat = _check(refio.tree_to_bfs_arrays(T), samples, chunk_nodes, n_groups)
    assert len(flat.dfs2bfs) == 474 and sorted(flat.dfs2bfs.tolist()) == list(range(474))


@pytest.mark.parametrize("seed", [21, 22, 23, 24])
def test_stream_matches_oracle_random(seed):
    arrays, queries = synth.make_case(seed, n_leaves=150, n_queries=10, n_sites=60, p_masked=0.05 if seed % 2 else 0.0,
                                      root_muts=seed % 3)
    _check(arrays, queries, chunk_nodes=11, n_groups=9)
    _check(arrays, queries, chunk_nodes=0, n_groups=1, scores=False)


def test_stream_matches_oracle_when_the_numbering_is_not_breadth_first():
    """parent[j] < j is all the boundary asks for: the same tree numbered in preorder gives the same placements
    (node indices translated), through the serial fallback of the flattening's level passes."""
    arrays, queries = synth.make_case(27, n_leaves=160, n_queries=8, n_sites=60, p_masked=0.05, root_muts=1)
    pre, new = synth.relabel_preorder(arrays)
    assert (np.diff(pre["parent"][1:]) < 0).any()           # really not a breadth-first expansion
    _check(pre, queries, chunk_nodes=13, n_groups=5)
    ot, otp = capi.OracleTree(arrays), capi.OracleTree(pre)
    for s in queries:
        a, b = ot.place(s, compute_scores=True), otp.place(s, compute_scores=True)
        assert (a["best"], a["num_best"]) == (b["best"], b["num_best"])
        assert b["scores"][new].tolist() == a["scores"].tolist()


def test_stream_on_syn_fixture():
    T = refio.load_mutation_annotated_tree(os.path.join(GOLD, "survey_ref", "syn", "tree.pb"))
    samples = [refio.sample_to_arrays(s) for s in refio.read_vcf(T, os.path.join(GOLD, "survey_ref", "syn", "query.vcf"))]
    _check(refio.tree_to_bfs_arrays(T), samples[:8], chunk_nodes=100, n_groups=5)


def test_slot_depth_is_logarithmic():
    # a caterpillar (every internal node has one leaf child and one internal child) is the
    # worst case for a per-level stack; largest-child-last keeps the slot count tiny
    n_int = 300
    parent = [-1]
    for i in range(n_int):
        # BFS order: node 2i+1 = leaf child, 2i+2 = internal child of node (2(i-1)+2 or 0)
        par = 0 if i == 0 else 2 * (i - 1) + 2
        parent += [par, par]
    n = len(parent)
    arrays = {"n": n, "parent": np.array(parent), "mut_off": np.zeros(n + 1, np.int64), "mut_pos": np.zeros(0, np.int32),
              "mut_ref": np.zeros(0, np.int8), "mut_par": np.zeros(0, np.int8), "mut_nuc": np.zeros(0, np.int8)}
    flat = FlatTreeView(arrays)
    assert flat.max_slots <= 2
    rng = np.random.default_rng(5)
    arrays2, _ = synth.make_case(3, n_leaves=2000, n_queries=0)
    assert FlatTreeView(arrays2).max_slots <= int(np.log2(arrays2["n"])) + 2


def test_pruning_records_carry_the_true_depth_and_second_hit_counts():
    """Every pruning pseudo-record of the tie stream names the node behind it (its key word) and carries hsub = the largest
    number of mutations on a path node -> descendant and the largest number of SECOND HITS on such a path: mutations of a
    site that is not at its reference base in the parent.  Recomputed here from the tree arrays by brute force (states
    tracked down every root path); an undercount would make the pruning of both phases skip real ties."""
    arrays, _ = synth.make_case(17, n_leaves=700, n_queries=0, n_sites=60, p_masked=0.03, mut_counts=(0, 0, 1, 1, 2, 3, 6))
    n = int(arrays["n"])
    par = np.asarray(arrays["parent"]).astype(np.int64)
    off = np.asarray(arrays["mut_off"]).astype(np.int64)
    pos, ref, nuc = (np.asarray(arrays[k]) for k in ("mut_pos", "mut_ref", "mut_nuc"))
    state = [None] * n            # per node: {position: allele} of the sites off the reference on its root path
    own = np.zeros(n, np.int64); sec = np.zeros(n, np.int64)
    for j in range(n):            # BFS numbering: parents first
        st = dict(state[par[j]]) if j else {}
        for i in range(off[j], off[j + 1]):
            if pos[i] < 0:
                continue          # masked mutations carry no site
            own[j] += 1
            sec[j] += int(pos[i]) in st        # the parent's state at this site is not the reference base
            if nuc[i] == ref[i]:
                st.pop(int(pos[i]), None)
            else:
                st[int(pos[i])] = int(nuc[i])
        state[j] = st
    hsub = np.zeros(n, np.int64); hsec = np.zeros(n, np.int64)
    for j in range(n - 1, 0, -1):
        hsub[par[j]] = max(hsub[par[j]], own[j] + hsub[j])
        hsec[par[j]] = max(hsec[par[j]], sec[j] + hsec[j])
    flat = FlatTreeView(arrays, chunk_nodes=50)
    st_words = flat.stream_t
    seen = 0
    for c in range(len(flat.chunk_t_off) - 1):
        i, hi = int(flat.chunk_t_off[c]), int(flat.chunk_t_off[c + 1])
        while i < hi:
            w0 = int(st_words[i]); w1 = int(st_words[i + 1]); i += 2
            if (w0 & 0xFFFF) == 0xFFFF:       # a pruning pseudo-record: the node's own record follows
                node = int(flat.rank2bfs[int(st_words[i + 1]) >> 1])
                assert (w1 >> 24) == hsub[node], (node, w1 >> 24, hsub[node])
                assert ((w0 >> 16) & 0xFF) == min(int(hsec[node]), 255), (node, (w0 >> 16) & 0xFF, hsec[node])
                seen += 1
                continue
            i += w0 & 0xFFFF
    assert seen > 20 and hsec.max() >= 2      # (60 sites: second hits are common in this tree)


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_both_lower_bounds_hold_for_every_node_and_sample(seed):
    """The two bounds the pruning rests on, checked as stated, by brute force: for every node n, every strict
    descendant d and every sample s,   cost(d, s) >= D(n, s) - hsub(n)   and   cost(d, s) >= B(n, s) - hsec(n),
    with B = the mismatching sites at which the sample's set holds the reference base.  Samples with N and IUPAC cells, trees
    with masked mutations, few sites (second hits and reversions are common)."""
    arrays, queries = synth.make_case(seed, n_leaves=350, n_queries=30, n_sites=40, p_masked=0.03, mut_counts=(0, 0, 1, 1, 2, 3, 5),
                                      n_ambig=(0, 2, 5, 12))
    n = int(arrays["n"])
    par = np.asarray(arrays["parent"]).astype(np.int64)
    off = np.asarray(arrays["mut_off"]).astype(np.int64)
    pos, ref, nuc = (np.asarray(arrays[k]).astype(np.int64) for k in ("mut_pos", "mut_ref", "mut_nuc"))
    state = [None] * n
    prev = np.zeros(len(pos), np.int64)
    own = np.zeros(n, np.int64); sec = np.zeros(n, np.int64)
    for j in range(n):
        st = dict(state[par[j]]) if j else {}
        for i in range(off[j], off[j + 1]):
            if pos[i] < 0:
                continue
            p = int(pos[i])
            prev[i] = st.get(p, int(ref[i]))
            own[j] += 1
            sec[j] += p in st
            if nuc[i] == ref[i]:
                st.pop(p, None)
            else:
                st[p] = int(nuc[i])
        state[j] = st
    hsub = np.zeros(n, np.int64); hsec = np.zeros(n, np.int64)
    for j in range(n - 1, 0, -1):
        hsub[par[j]] = max(hsub[par[j]], own[j] + hsub[j])
        hsec[par[j]] = max(hsec[par[j]], sec[j] + hsec[j])
    checked = tight = 0
    for s in queries:
        S = {}
        d_bot = 0
        for p, r, a, mis in zip(s["pos"], s["ref"], s["nuc"], s["is_missing"]):
            S[int(p)] = 0xF if mis else int(a)
            d_bot += (not mis) and (int(a) & int(r)) == 0
        D = np.zeros(n, np.int64); B = np.zeros(n, np.int64); cost = np.zeros(n, np.int64)
        for j in range(n):
            dp, bp = (D[par[j]], B[par[j]]) if j else (d_bot, 0)
            dsum = bsum = neg = 0
            masked = False
            for i in range(off[j], off[j + 1]):
                if pos[i] < 0:
                    masked = True
                    continue
                sp = S.get(int(pos[i]), int(ref[i]))
                delta = (1 if sp & prev[i] else 0) - (1 if sp & nuc[i] else 0)
                dsum += delta
                bsum += delta if sp & ref[i] else 0
                if not masked:
                    neg += min(delta, 0)
            D[j], B[j] = dp + dsum, bp + bsum
            cost[j] = D[j] if j == 0 else dp + neg
            assert 0 <= B[j] <= D[j]
        below = np.full(n, 1 << 30, np.int64)     # smallest cost among the strict descendants
        for j in range(n - 1, 0, -1):
            below[par[j]] = min(below[par[j]], cost[j], below[j])
        inner = below < (1 << 30)
        assert (below[inner] >= D[inner] - hsub[inner]).all()
        assert (below[inner] >= B[inner] - hsec[inner]).all()
        checked += int(inner.sum())
        tight += int((below[inner] == np.maximum(D[inner] - hsub[inner], B[inner] - hsec[inner])).sum())
    assert checked > 5000 and tight > 0      # (the bounds are attained somewhere: they are not vacuous)


def test_second_bound_is_withheld_on_deep_trees():
    """k_best8 keeps B (the part of D at sites where the sample holds the reference base) in one byte per sample; B never
    exceeds the mutations on a root path, so a tree with more than 255 of them gets pruning records that say "second hits:
    not available" everywhere -- and the walk stays exact with the first bound alone."""
    arrays, queries = synth.caterpillar_case(7, depth=200, muts_per_node=2, n_queries=3)
    flat = FlatTreeView(arrays, chunk_nodes=40)
    assert flat.max_path_muts > 255
    recs = [int(w) for w in flat.stream8 if (int(w) >> 31) and (int(w) >> 30) & 1]
    assert recs and all((w >> stream_interp.INFO_HR_SHIFT) & 7 == stream_interp.INFO_HR_NONE for w in recs)
    ot = capi.OracleTree(arrays)
    n_chunks = len(flat.chunk8_body_off) - 1
    for s in queries:
        want = ot.place(s)
        res = stream_interp.place8(flat, s, n_groups=max(1, n_chunks // 3), prune_ub=want["best"] + 1)
        assert (res["best"], res["num_best"], res["best_j"]) == (want["best"], want["num_best"], want["best_j"])
    arrays2, _ = synth.make_case(5, n_leaves=300, n_queries=0)
    flat2 = FlatTreeView(arrays2, chunk_nodes=40)   # a shallow tree: the records carry counts
    recs2 = [int(w) for w in flat2.stream8 if (int(w) >> 31) and (int(w) >> 30) & 1]
    assert flat2.max_path_muts <= 255 and any((w >> stream_interp.INFO_HR_SHIFT) & 7 != stream_interp.INFO_HR_NONE for w in recs2)


def test_hot_slots_are_capped(monkeypatch):
    """k_best8 keeps the B halves of the on-chip slots in two 16-element register vectors: however many are asked for, the
    stream is encoded for at most 16 (the rest go through the cold path)."""
    monkeypatch.setenv("UGP_LDS_SLOTS", "40")
    # a comb of combs: many non-last-child edges on the deepest path
    arrays, _ = synth.polytomy_case(5, fanouts=(3, 3, 3, 3, 3, 3, 3), n_queries=0) if hasattr(synth, "polytomy_case") else synth.make_case(3, n_leaves=2000, n_queries=0)
    flat = FlatTreeView(arrays)
    assert flat.lds_slots <= 16 and flat.lds_slots <= max(flat.max_slots, 1)


def test_flatten_rejects_bad_trees():
    from usher_amd import UgpError
    base = {"n": 2, "parent": np.array([-1, 0]), "mut_off": np.array([0, 0, 1]), "mut_pos": np.array([7], np.int32),
            "mut_ref": np.array([1], np.int8), "mut_par": np.array([1], np.int8), "mut_nuc": np.array([3], np.int8)}
    with pytest.raises(UgpError) as e:
        FlatTreeView(base)            # ambiguous tree allele
    assert e.value.code == -2
    bad = dict(base, mut_nuc=np.array([2], np.int8), parent=np.array([-1, 1]))
    with pytest.raises(UgpError) as e:
        FlatTreeView(bad)             # parent[j] >= j
    assert e.value.code == -1


def test_packed_stream_flushes_long_branches():
    """A node with more than 15 mutations exercises the 4-bit counter spill (M_FLUSH)."""
    rng = np.random.default_rng(9)
    arrays, queries = synth.make_case(50, n_leaves=40, n_queries=8, n_sites=200, mut_counts=(0, 1, 17, 33, 40))
    flat = _check(arrays, queries, chunk_nodes=7, n_groups=4)
    assert (((flat.stream8 >> 31) == 0) & ((flat.stream8 & (1 << 28)) != 0)).any() and flat.max_path_muts > 30


def test_pruning_records_skip_far_subtrees_exactly():
    """Trees large enough to carry pruning records (subtrees >= 256 stream words): with a tight
    upper bound most of the stream is skipped and the result does not change."""
    arrays, queries = synth.make_case(61, n_leaves=3000, n_queries=6, n_sites=400, n_ambig=(0, 0, 2))
    flat = FlatTreeView(arrays, chunk_nodes=700)
    assert ((flat.stream8 >> 30) == 3).sum() > 5            # H_TAG | H_INFO words exist
    ot = capi.OracleTree(arrays)
    skipped = 0
    for s in queries:
        want = ot.place(s)
        for ub in (0x7F7F, want["best"]):
            st = {}
            got = stream_interp.place8(flat, s, n_groups=3, prune_ub=ub, stats=st)
            for k in ("best", "num_best", "best_j", "has_unique"):
                assert got[k] == want[k], (ub, k)
            skipped += st.get("skipped", 0)
    assert skipped > len(flat.stream8)                      # pruning really happened


def test_sibling_records_skip_runs_of_siblings_exactly():
    """Sibling records (H_INFO | H_SIB): one jump over a child and all later non-last siblings when
    D(parent) minus the largest downward mutation count among them exceeds the bound.  Present in the
    stream, used by the model, and the results do not change."""
    arrays, queries = synth.make_case(63, n_leaves=5000, n_queries=6, n_sites=300, n_ambig=(0, 0, 2))
    flat = FlatTreeView(arrays, chunk_nodes=900)
    is_sib = ((flat.stream8 >> 30) == 3) & ((flat.stream8 & (1 << 21)) != 0)
    assert is_sib.sum() > 20
    ot = capi.OracleTree(arrays)
    jumps = 0
    for s in queries:
        want = ot.place(s)
        for ub in (0x7F7F, want["best"] + 2, want["best"]):
            st = {}
            got = stream_interp.place8(flat, s, n_groups=4, prune_ub=ub, stats=st)
            for k in ("best", "num_best", "best_j", "has_unique"):
                assert got[k] == want[k], (ub, k)
            jumps += st.get("sibling_jumps", 0)
    assert jumps > 10


def _flat_arrays(arrays, **kw):
    v = FlatTreeView(arrays, **kw)
    out = {}
    for name in dir(v):
        a = getattr(v, name)
        if not name.startswith("_") and isinstance(a, (np.ndarray, int, bool)):
            out[name] = np.array(a)
    return out


@pytest.mark.parametrize("kind", ["random", "masked", "caterpillar", "polytomy", "synth200k", "not_bfs"])
def test_flattening_does_not_depend_on_the_thread_count(kind, monkeypatch):
    """The flattening runs on host threads (level passes, DFS segments, chunks); every stream must be bit-identical
    whatever the number of threads, including more threads than work."""
    from usher_amd import synth as gsynth
    kw = {"chunk_nodes": 24}
    if kind == "random":
        arrays, _ = synth.make_case(5, n_leaves=900, n_queries=1, n_sites=150)
    elif kind == "masked":
        arrays, _ = synth.make_case(41, n_leaves=400, n_queries=1, n_sites=60, p_masked=0.15, root_muts=3)
    elif kind == "caterpillar":
        arrays, _ = synth.caterpillar_case(3, depth=600, muts_per_node=2, n_queries=1)
    elif kind == "polytomy":
        arrays, _ = synth.polytomy_case(4, n_queries=1)
    elif kind == "synth200k":
        arrays, kw = gsynth.SynthTree(200_000, n_sites=3000, seed=9, shape="sars2").arrays, {}
    else:   # parents before children but not a breadth-first order: the serial fallback of the level passes
        arrays, _ = synth.make_case(6, n_leaves=300, n_queries=1, n_sites=80)
        arrays, _ = synth.relabel_preorder(arrays)
    monkeypatch.setenv("UGP_FLATTEN_THREADS", "1")
    ref = _flat_arrays(arrays, **kw)
    if kind != "synth200k":
        monkeypatch.setenv("UGP_FLATTEN_GRAIN", "1")   # split even the smallest passes
    for threads in ("2", "5", "64"):
        monkeypatch.setenv("UGP_FLATTEN_THREADS", threads)
        got = _flat_arrays(arrays, **kw)
        assert set(got) == set(ref)
        for k in ref:
            assert got[k].shape == ref[k].shape and (got[k] == ref[k]).all(), (k, threads)


@pytest.mark.parametrize("seed,chunk_nodes", [(81, 9), (82, 30), (83, 64)])
def test_preamble_records_skip_far_units_exactly(seed, chunk_nodes, monkeypatch):
    """Every path node of a chunk's preamble carries a pruning record (hsub, reversions below, where the body goes on
    behind its subtree).  With them, and with the second bound D - (V_s + hrev), a unit far from the sample ends its
    replay a few nodes below the point where its root path leaves the sample's and skips most of its body; the chunk
    minima that matter must be unchanged, for loose and tight upper bounds, units of 1, 3 and 8 chunks."""
    monkeypatch.setenv("UGP_LDS_SLOTS", "3")
    arrays, queries = synth.make_case(seed, n_leaves=2500, n_queries=8, n_sites=3000, n_ambig=(0, 0, 2), p_masked=0.02,
                                      mut_counts=(0, 0, 1, 1, 1, 2, 3, 17))
    flat = FlatTreeView(arrays, chunk_nodes=chunk_nodes)
    n_chunks = len(flat.chunk8_body_off) - 1
    ot = capi.OracleTree(arrays)
    pre_skipped = body = 0
    for s in queries:
        want = ot.place(s)
        nib, dbot = stream_interp.sample_site_alleles(flat, s)
        for ub0 in (0x7F7F, want["best"] + 2, want["best"]):
            for unit in (1, 3, 8, 16):
                st = {}
                lb_pre, lb_plain = {}, {}
                ub_a, ub_b = [ub0], [ub0]
                for c0 in range(0, n_chunks, unit):
                    c1 = min(c0 + unit, n_chunks)
                    lb_pre.update(stream_interp.best8_group(flat, nib, dbot, c0, c1, ub_a, st))
                    lb_plain.update(stream_interp.best8_group(flat, nib, dbot, c0, c1, ub_b, None, use_pre_records=False))
                assert min(lb_pre.values()) == want["best"] == min(lb_plain.values())
                assert [c for c in range(n_chunks) if lb_pre[c] == want["best"]] == [c for c in range(n_chunks) if lb_plain[c] == want["best"]]
                if ub0 == want["best"]:
                    pre_skipped += st.get("pre_skipped", 0)
                    body += len(flat.stream8)
        res = stream_interp.place8(flat, s, n_groups=max(1, n_chunks // 4), prune_ub=want["best"] + 1)
        assert (res["best"], res["num_best"], res["best_j"]) == (want["best"], want["num_best"], want["best_j"])
    assert pre_skipped > 0.25 * body   # a large part of the tree is ruled out during the replays (5,000 nodes: units are coarse)


@pytest.mark.parametrize("seed", [1301, 1302])
def test_third_lower_bound_holds_for_every_node_and_sample(seed):
    """Round 5 -- the bound the third test rests on, checked as stated (ugp_flatten.hpp "B3") by brute force on the tree arrays: a
    mutation lowers D for sample s only if it is USEFUL for s (its allele lies in the sample's set and the reference base does not)
    or a SECOND HIT (the parent state is not the reference base); hence for a tile T of samples, every node n, every strict
    descendant d and every s in T:    cost(d, s) >= D(n, s) - (hU_T(n) + hsec(n)),
    hU_T(n) = the largest number, over the paths below n, of mutations useful for some sample of T.  Few sites, N and IUPAC cells,
    masked mutations; tiles of 1, 4 and 30 samples; the bound is attained somewhere and beats hsub somewhere."""
    arrays, queries = synth.make_case(seed, n_leaves=300, n_queries=30, n_sites=40, p_masked=0.03, mut_counts=(0, 0, 1, 1, 2, 3, 5), n_ambig=(0, 2, 5, 12))
    n = int(arrays["n"])
    par = np.asarray(arrays["parent"]).astype(np.int64)
    off = np.asarray(arrays["mut_off"]).astype(np.int64)
    pos, ref, nuc = (np.asarray(arrays[k]).astype(np.int64) for k in ("mut_pos", "mut_ref", "mut_nuc"))
    state = [None] * n
    prev = np.zeros(len(pos), np.int64)
    own = np.zeros(n, np.int64); sec = np.zeros(n, np.int64)
    for j in range(n):
        st = dict(state[par[j]]) if j else {}
        for i in range(off[j], off[j + 1]):
            if pos[i] < 0:
                continue
            p = int(pos[i])
            prev[i] = st.get(p, int(ref[i]))
            own[j] += 1
            sec[j] += p in st
            if nuc[i] == ref[i]:
                st.pop(p, None)
            else:
                st[p] = int(nuc[i])
        state[j] = st
    hsub = np.zeros(n, np.int64); hsec = np.zeros(n, np.int64)
    for j in range(n - 1, 0, -1):
        hsub[par[j]] = max(hsub[par[j]], own[j] + hsub[j])
        hsec[par[j]] = max(hsec[par[j]], sec[j] + hsec[j])
    sets = []
    for s in queries:
        sets.append({int(p): (0xF if mis else int(a)) for p, a, mis in zip(s["pos"], s["nuc"], s["is_missing"])})
    checked = tight = better = 0
    for tile in ([0], [3, 4, 5, 6], list(range(len(queries)))):
        useful = np.zeros(len(pos), bool)          # per mutation: useful for some sample of the tile
        for i in range(len(pos)):
            if pos[i] < 0:
                continue
            for t in tile:
                sp = sets[t].get(int(pos[i]), int(ref[i]))
                if (sp & nuc[i]) and not (sp & ref[i]):
                    useful[i] = True
                    break
        un = np.array([useful[off[j]:off[j + 1]].sum() for j in range(n)], np.int64)
        hu = np.zeros(n, np.int64)
        for j in range(n - 1, 0, -1):
            hu[par[j]] = max(hu[par[j]], un[j] + hu[j])
        for t in tile:
            S = sets[t]
            s = queries[t]
            d_bot = sum(1 for r, a, mis in zip(s["ref"], s["nuc"], s["is_missing"]) if not mis and (int(a) & int(r)) == 0)
            D = np.zeros(n, np.int64); cost = np.zeros(n, np.int64)
            for j in range(n):
                dp = D[par[j]] if j else d_bot
                dsum = neg = 0
                masked = False
                for i in range(off[j], off[j + 1]):
                    if pos[i] < 0:
                        masked = True
                        continue
                    sp = S.get(int(pos[i]), int(ref[i]))
                    delta = (1 if sp & prev[i] else 0) - (1 if sp & nuc[i] else 0)
                    dsum += delta
                    if not masked:
                        neg += min(delta, 0)
                D[j] = dp + dsum
                cost[j] = D[j] if j == 0 else dp + neg
            below = np.full(n, 1 << 30, np.int64)
            for j in range(n - 1, 0, -1):
                below[par[j]] = min(below[par[j]], cost[j], below[j])
            inner = below < (1 << 30)
            assert (below[inner] >= D[inner] - (hu[inner] + hsec[inner])).all()
            checked += int(inner.sum())
            tight += int((below[inner] == D[inner] - (hu[inner] + hsec[inner])).sum())
            better += int(((hu + hsec)[inner] < hsub[inner]).sum())
    assert checked > 5000 and tight > 0 and better > 0


def _tables_never_below_the_stream_truth(flat, b3):
    """What the tables give (stream_interp.b3_hu) against the true path maximum of useful events below every record, recomputed
    from the packed stream itself."""
    s8 = flat.stream8.astype(np.int64)
    H_TAG = 1 << 31
    # against the stream: walk the body once, keeping for every open node the running count of useful words on its root path
    useful = b3["useful"]
    INFO, RARE, SIB, CE, NOP, ENDF = 1 << 30, 1 << 29, 1 << 21, 1 << 8, 1 << 9, 1 << 3
    recs = []            # (position of the node's last word, jump, cum at the node)
    cum_at = np.zeros(len(s8), np.int64)     # cum of the node that owns each word (headers and mutation words)
    # subtree extents from the pruning records themselves: a record in front of a node gives the words of its descendants
    # cum(node) = cum(parent) + useful words of the node; parents via a stack of (end position, cum)
    stack = []           # (end of subtree, cum)
    i = 0
    pend = None
    node_cum = 0
    while i < len(s8):
        w = int(s8[i])
        if w & H_TAG:
            if w & INFO:
                if not (w & SIB):
                    pend = w & ((1 << 18) - 1)
                i += 1
                continue
            if w & (CE | NOP) and w & RARE:
                i += 1
                continue
            while stack and stack[-1][0] <= i:
                stack.pop()
            base = stack[-1][1] if stack else 0
            k = i + 1
            cnt = 0
            if not (w & ENDF):
                while True:
                    m = int(s8[k])
                    cnt += (useful[m & 0x3FFFFF] >> ((m >> 22) & 3)) & 1
                    k += 1
                    if m & (1 << 30):
                        break
            node_cum = base + cnt
            last = k - 1
            if pend is not None:
                recs.append((last, pend, node_cum))
                stack.append((last + 1 + pend, node_cum))
                pend = None
            i = k
            continue
        i += 1
    assert len(recs) > 50
    # true maximum below a record = the largest cum among the records / nodes inside its range: recompute by brute force over words
    # (cum of any node inside the range is bounded by the cum of nodes that carry records or not -- walk again, cheaply, per record)
    pos_cum = {}
    stack = []
    i = 0
    pend = None
    while i < len(s8):
        w = int(s8[i])
        if w & H_TAG:
            if w & INFO:
                if not (w & SIB):
                    pend = w & ((1 << 18) - 1)
                i += 1
                continue
            if w & (CE | NOP) and w & RARE:
                i += 1
                continue
            while stack and stack[-1][0] <= i:
                stack.pop()
            base = stack[-1][1] if stack else 0
            k = i + 1
            cnt = 0
            if not (w & ENDF):
                while True:
                    m = int(s8[k])
                    cnt += (useful[m & 0x3FFFFF] >> ((m >> 22) & 3)) & 1
                    k += 1
                    if m & (1 << 30):
                        break
            pos_cum[k - 1] = base + cnt
            if pend is not None:
                stack.append((k + pend, base + cnt))
                pend = None
            i = k
            continue
        i += 1
    lasts = np.array(sorted(pos_cum), np.int64)
    cums = np.array([pos_cum[p] for p in lasts], np.int64)
    loose = 0
    for last, jump, c in recs:
        a, b = np.searchsorted(lasts, last + 1), np.searchsorted(lasts, last + 1 + jump)
        true_hu = max(0, int(cums[a:b].max()) - c) if b > a else 0
        # (nodes without a record of their own inherit the cum of the nearest recorded ancestor in this bookkeeping: an
        # under-estimate of their cum, hence of true_hu -- the inequality tested is still the one that matters: tables >= truth seen)
        got = stream_interp.b3_hu(b3, last, jump)
        assert got >= true_hu, (last, jump, got, true_hu)
        loose += got - true_hu
    return len(recs), loose


@pytest.mark.parametrize("seed,chunk_nodes", [(71, 700), (72, 64), (73, 0)])
def test_third_bound_tables_skip_more_and_change_nothing(seed, chunk_nodes):
    """The third bound as the device builds and reads it (tests/stream_interp.b3_tables / b3_hu: posting lists of the flattening ->
    cum_over / cum_under per block of 16 stream words -> the maximum over the descendants' blocks at the coarsest 64-ary level that
    fits, minus cum_under of the node's own block): (1) the event lists hold every mutation word of the packed body exactly once; (2) what the tables give is never below the true path maximum of useful events -- recomputed from the
    stream itself -- for every record; (3) the model of the walk, with the tables of the whole batch as one tile and each sample's
    own exact score as its bound, returns the oracle's answers and skips more of the stream than without."""
    arrays, queries = synth.make_case(seed, n_leaves=2600, n_queries=10, n_sites=300, n_ambig=(0, 0, 2, 6), p_masked=0.01)
    flat = FlatTreeView(arrays, chunk_nodes=chunk_nodes)
    s8 = flat.stream8.astype(np.int64)
    H_TAG = 1 << 31
    is_mut = (s8 & H_TAG) == 0
    # (1) every mutation word once
    ng = len(flat.b3_group_off) // 4 - 1
    off = flat.b3_group_off.astype(np.int64).reshape(4, ng + 1)
    ev = flat.b3_events.astype(np.int64)
    assert off[0, 0] == 0 and (np.diff(off, axis=1) >= 0).all() and off[1, 0] == off[0, ng] and off[2, 0] == off[1, ng] and off[3, 0] == off[2, ng] and off[3, ng] == len(ev)
    assert off[3, 1] == off[3, 0]   # (nothing is open in front of the first group)
    assert off[1, ng] == int(is_mut.sum()) and off[2, ng] - off[2, 0] == off[1, ng] - off[1, 0]   # one start and one end per spanning event
    pairs = (s8[is_mut] & 0x3FFFFF) * 4 + ((s8[is_mut] >> 22) & 3)
    np.testing.assert_array_equal(np.bincount(pairs, minlength=4 * len(flat.site_ref)), np.bincount(ev[:off[1, ng]] & 0xFFFFFF, minlength=4 * len(flat.site_ref)))
    np.testing.assert_array_equal(np.bincount(ev[off[1, 0]:off[1, ng]] & 0xFFFFFF), np.bincount(ev[off[2, 0]:off[2, ng]] & 0xFFFFFF))
    # lists 0 and 1 follow the stream: an event is listed under the block of its own mutation word (round 6)
    blk = lambda k: np.repeat(np.arange(ng), np.diff(off[k])) * 256 + (ev[off[k, 0]:off[k, ng]] >> 24)
    assert (np.sort(np.concatenate([blk(0), blk(1)])) == np.flatnonzero(is_mut) >> 4).all() and blk(2).max() <= (len(s8) - 1) >> 4
    nibs = [stream_interp.sample_site_alleles(flat, s)[0] for s in queries]
    b3 = stream_interp.b3_tables(flat, nibs)
    # (2) against the stream
    _tables_never_below_the_stream_truth(flat, b3)
    # (3) exact results, more skipped
    ot = capi.OracleTree(arrays)
    sk0 = sk1 = asked = 0
    for s in queries:
        want = ot.place(s)
        for tables in (None, b3):
            st = {}
            got = stream_interp.place8(flat, s, n_groups=3, prune_ub=want["best"], stats=st, b3=tables)
            for k in ("best", "num_best", "best_j", "has_unique"):
                assert got[k] == want[k], (k, tables is not None)
            if tables is None:
                sk0 += st.get("skipped", 0)
            else:
                sk1 += st.get("skipped", 0); asked += st.get("b3_asked", 0)
    assert asked > 0 and sk1 >= sk0


@pytest.mark.parametrize("seed", [81, 82])
def test_third_bound_tables_with_long_branches(seed):
    """a branch with hundreds of mutations that the queries share.  Until round 6 every event of a node was listed
    under the block of the node's HEADER word, so such a node put >= 256 range starts into one block and the table kernel's 8-bit
    counter wrapped into its neighbour: cum_under came out as 65535 and the walk pruned a subtree that held the answer.  Now an
    event is listed under the block of its own word: (1) no block starts more events than it has words, whatever the tile;
    (2) ends per block stay below the 16-bit field; (3) the tables are still bounds; (4) the model of the walk returns the oracle."""
    arrays, queries = synth.make_case(seed, n_leaves=900, n_queries=8, genome_len=4000, n_sites=900, n_ambig=(0, 0, 2),
                                      mut_counts=(0, 1, 1, 1, 2, 3) * 8 + (300, 420))
    nm = np.diff(np.asarray(arrays["mut_off"]))
    assert (nm >= 300).sum() >= 5
    flat = FlatTreeView(arrays, chunk_nodes=0)
    ng = len(flat.b3_group_off) // 4 - 1
    assert ng > 0, "no event lists"
    off = flat.b3_group_off.astype(np.int64).reshape(4, ng + 1)
    ev = flat.b3_events.astype(np.int64)
    per_block = []
    for k in range(3):
        e = ev[off[k, 0]:off[k, ng]]
        blk = np.repeat(np.arange(ng), np.diff(off[k])) * 256 + (e >> 24)
        per_block.append(np.bincount(blk, minlength=ng * 256))
    assert per_block[0].max() <= 16 and per_block[1].max() <= 16 and (per_block[0] + per_block[1]).max() <= 16
    assert per_block[2].max() <= int(flat.max_path_muts) < 0x7F7F
    # the queries below a long branch share its mutations: the pairs are useful for the tile
    nibs = [stream_interp.sample_site_alleles(flat, s)[0] for s in queries]
    b3 = stream_interp.b3_tables(flat, nibs)
    assert b3["over"].max() >= 256, "no sample sits below a long branch: the case does not exercise the counters"
    n_rec, _ = _tables_never_below_the_stream_truth(flat, b3)
    assert n_rec > 20
    ot = capi.OracleTree(arrays)
    for s in queries:
        want = ot.place(s)
        got = stream_interp.place8(flat, s, n_groups=2, prune_ub=want["best"], stats={}, b3=b3)
        for k in ("best", "num_best", "best_j", "has_unique"):
            assert got[k] == want[k], k
