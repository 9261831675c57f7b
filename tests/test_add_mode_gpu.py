"""Add mode on the device (ugp_mat_update / ugp_touched_*): the flattened tree stays on the device, the nodes created or rewritten
by insertions are scored from records, rewritten nodes leave the flattened tree's candidate set -- and the merge of the two is
the answer of a full search of the tree as it is now.  Checked against the oracle searching the EDITED tree (tests/usher_model's
restatement of usher_common.cpp:652-765 does the insertions), never against the product itself."""
import os

import numpy as np
import pytest

from oracle import capi, refio
from tests import synth, usher_model
from usher_amd import Placer, QueryBatch

pytestmark = pytest.mark.gpu
INT_MAX = 2 ** 31 - 1


def tree_from_arrays(arrays):
    """refio.Tree with the nodes of `arrays` (names n<j>), children in index order."""
    T = refio.Tree()
    nodes = []
    for j in range(int(arrays["n"])):
        p = int(arrays["parent"][j])
        nd = T.create_node("n%d" % j, nodes[p] if p >= 0 else None)
        for i in range(int(arrays["mut_off"][j]), int(arrays["mut_off"][j + 1])):
            m = refio.Mutation(int(arrays["mut_pos"][i]), int(arrays["mut_ref"][i]), int(arrays["mut_par"][i]), int(arrays["mut_nuc"][i]))
            nd.mutations.append(m)           # (already in the stored order)
        nodes.append(nd)
    return T, nodes


def record_of(node, flat_index):
    """What the driver reports for a touched node: the parent's state wherever it is not the reference base, the own mutations in
    front of the first masked one with their true parent state."""
    state, ref = {}, {}
    a = node.parent
    chain = []
    while a is not None:
        chain.append(a)
        a = a.parent
    for a in reversed(chain):                       # root first: later mutations overwrite
        for m in a.mutations:
            if m.is_masked():
                continue
            state[m.position] = m.mut_nuc
            ref[m.position] = m.ref_nuc
    path = [(p, s, ref[p]) for p, s in state.items() if s != ref[p]]
    own, masked = [], False
    for m in node.mutations:
        if m.is_masked():
            masked = True
            break
        own.append((m.position, m.mut_nuc, state.get(m.position, m.ref_nuc), m.ref_nuc))
    return {"flat_j": flat_index, "leaf": node.is_leaf(), "masked": masked, "path": path, "own": own}


def merged_answer(flat_best, flat_ties, t_best, t_ties):
    """(best, {node: has_unique}) from the flattened tree's untouched nodes and the records."""
    best = min(flat_best, t_best)
    out = {}
    if flat_best == best:
        out.update(flat_ties)
    if t_best == best:
        out.update(t_ties)
    return best, out


@pytest.mark.parametrize("seed,n_leaves,n_new,chunk", [(5, 400, 60, 16), (6, 1500, 90, 64), (7, 120, 40, 8)])
def test_flattened_tree_plus_records_equal_a_search_of_the_edited_tree(seed, n_leaves, n_new, chunk, monkeypatch):
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")     # the locality pre-pass and its coarse tree too (their nodes are excluded as well)
    arrays, queries = synth.make_case(seed, n_leaves=n_leaves, n_queries=n_new, n_sites=90, n_ambig=(0, 0, 1, 3), p_masked=0.01)
    T, flat_nodes = tree_from_arrays(arrays)
    flat_of = {id(n): j for j, n in enumerate(flat_nodes)}
    pl = Placer(arrays, chunk_nodes=chunk)
    batch = QueryBatch(queries)
    pl.touched_open(batch)
    rec_of_node = {}          # id(node) -> live record id
    node_of_rec = {}
    retired = set()
    rng = np.random.default_rng(seed)
    multi = 0
    for i, s in enumerate(queries):
        # ---- the answer for sample i on the tree as it is now, from the device: flattened part + records
        # (a few pending samples at a time, in front of 600 others so that the locality pre-pass, its coarse tree and the seed descent run)
        part = QueryBatch(queries[i:i + 3] + [queries[(7 * k + i) % len(queries)] for k in range(600)])
        fr = pl.place(part)
        tj, th, tc = pl.tied_nodes(part, 4096)
        tb, tcnt, tids, thu = pl.touched_fetch(i, min(3, len(batch) - i))
        bfs_now = T.breadth_first_expansion()
        ot = capi.OracleTree(refio.tree_to_bfs_arrays(T))
        for k in range(min(3, len(batch) - i)):
            want = ot.place(queries[i + k])
            flat_ties = {id(flat_nodes[int(j)]): bool(h) for j, h in zip(tj[k], th[k])}
            assert int(tc[k]) == len(flat_ties) == int(fr["num_best"][k])
            assert all(id(flat_nodes[int(j)]) not in rec_of_node for j in tj[k])   # rewritten nodes are no candidates any more
            live = [(int(r), bool(h)) for r, h in zip(tids[k][:int(tcnt[k])], thu[k][:int(tcnt[k])]) if int(r) not in retired]
            assert int(tcnt[k]) <= 64
            if int(tcnt[k]) and not live:                  # every listed record was retired since: ask again
                pl.touched_rescore(i + k)
                b2, c2, i2, h2 = pl.touched_fetch(i + k, 1)
                tb[k], live = b2[0], [(int(r), bool(h)) for r, h in zip(i2[0][:int(c2[0])], h2[0][:int(c2[0])])]
                assert all(r not in retired for r, _ in live)
            t_ties = {id(node_of_rec[r]): h for r, h in live}
            best, ties = merged_answer(int(fr["best_set_difference"][k]), flat_ties, int(tb[k]) if live else INT_MAX, t_ties)
            want_ties = {id(bfs_now[int(j)]): bool(h) for j, h in zip(want["ties"], want["ties_has_unique"])}
            assert best == want["best"], (i, k, best, want["best"])
            assert ties == want_ties, (i, k)
            multi += len(ties) > 1
        # ---- insert sample i where the oracle puts it (the reference's rule), report the touched nodes
        want = ot.place(s)
        node = bfs_now[want["best_j"]]
        nv = ot.node_vecs(s, want["best_j"])
        excess = [refio.Mutation(p, rf, pa, mu) for (p, rf, pa, mu) in nv["excess"]]
        as_sibling = node.is_leaf() or want["has_unique"]
        usher_model.insert(T, node, as_sibling, "NEW%d" % i, excess)
        leaf = T.get_node("NEW%d" % i)
        touched = [leaf.parent, leaf, node] if as_sibling else [leaf]
        gone = []
        recs = []
        for nd in touched:
            if id(nd) in rec_of_node:
                gone.append(rec_of_node[id(nd)])
            recs.append(record_of(nd, flat_of.get(id(nd))))
        first = pl.update(recs, gone)
        retired.update(gone)
        for k, nd in enumerate(touched):
            rec_of_node[id(nd)] = first + k
            node_of_rec[first + k] = nd
        pl.touched_score(first, i + 1)
        if rng.random() < 0.15:        # now and then: the running results equal a fresh evaluation of every live record
            a = pl.touched_fetch(i + 1, len(batch) - i - 1)
            pl.touched_open(batch)
            b = pl.touched_fetch(i + 1, len(batch) - i - 1)
            for q in range(len(a[0])):
                la = sorted(int(r) for r in a[2][q][:min(64, int(a[1][q]))] if int(r) not in retired)
                lb = sorted(int(r) for r in b[2][q][:min(64, int(b[1][q]))])
                # (a running list may hold retired records; when every holder of its minimum was retired the minimum is too low)
                if la:
                    assert a[0][q] == b[0][q] and (la == lb or int(a[1][q]) > 64), (i, q)
                else:
                    assert a[0][q] <= b[0][q], (i, q)
    assert multi > 3
    pl.close()


EXE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "usher_amd", "bin", "usher-amd")


@pytest.mark.parametrize("seed,n_leaves,n_new,env", [(21, 300, 620, {}), (22, 80, 520, {"USHER_AMD_BATCH": "128", "USHER_AMD_ROUND": "7"}),
                                                     (23, 500, 700, {"USHER_AMD_BATCH": "200", "USHER_AMD_ROUND": "64", "USHER_AMD_TCAP": "2"})])
def test_usher_cli_add_mode_on_the_device_equals_the_restated_driver_loop(seed, n_leaves, n_new, env, tmp_path):
    """bin/usher-amd in its default mode (sequential insertion, usher_common.cpp:310-792) with the edits kept on the device: more than
    500 insertions, multi-way ties among flattened nodes, new nodes and both; every line of placement_stats.tsv (score, number of
    optimal placements, imputed mutations) and every mutation path equal tests/usher_model.py -- the reference's loop restated in python
    with the ORACLE searching the current tree for every sample -- and the tree written equals the per-sample-research run of the same
    binary (a full search per sample: USHER_AMD_MAX_TOUCHED=0)."""
    import subprocess
    from tests.test_host_cli import _evolve_vcf, _read
    from tests.host_harness import run_usher
    rng = np.random.default_rng(seed)
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    _evolve_vcf(rng, n_leaves, 110, n_new, nh, old, new)
    pb = str(tmp_path / "base.pb")
    assert run_usher(["-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)]) == 0      # (MAT construction on the host harness: not what is tested here)
    outs = {}
    for mode, e in (("device", dict(env, USHER_AMD_PROFILE="1")), ("research", {"USHER_AMD_MAX_TOUCHED": "0"})):
        d = tmp_path / mode
        d.mkdir()
        r = subprocess.run([EXE, "-i", pb, "-v", new, "-u", "-d", str(d)], capture_output=True, text=True, timeout=1500, env=dict(os.environ, **e))
        assert r.returncode == 0, r.stderr[-3000:]
        outs[mode] = {n: _read(str(d / n)) for n in ("placement_stats.tsv", "mutation-paths.txt", "uncondensed-final-tree.nh")}
        if mode == "device":
            line = [l for l in r.stderr.splitlines() if "add mode on the device" in l]
            assert line and "tree -> arrays" in r.stderr
            flat = [l for l in r.stderr.splitlines() if "tree -> arrays" in l and "times)" in l][0]
            assert "(1 times)" in flat, flat                                       # one flattening for the whole run (made under the VCF read and taken over, or the loop's own)
    assert outs["device"] == outs["research"]
    T = refio.load_mutation_annotated_tree(pb)
    want = usher_model.run(T, refio.read_vcf(T, new))
    assert outs["device"]["placement_stats.tsv"] == want["placement_stats.tsv"]
    assert outs["device"]["mutation-paths.txt"] == want["mutation-paths.txt"]
    n_ties = sum(1 for l in want["placement_stats.tsv"].splitlines() if l.split("\t")[2] != "1")
    assert n_ties > 40 and len(want["placement_stats.tsv"].splitlines()) == n_new


def test_add_mode_at_one_million_nodes_device_equals_research_and_the_oracle_on_the_edited_tree(tmp_path):
    """the default mode at scale.  A 1M-node MAT, 1,000 sequential insertions through bin/usher-amd: the run that keeps
    the edits on the device (one flattening, records, exclusions) writes the same files as the run that searches the whole tree again
    for every sample (USHER_AMD_MAX_TOUCHED=0: the reference's loop, usher_common.cpp:310-792) -- and at 20 sampled steps k the tree
    after k insertions (saved by a run over the first k samples, read back as arrays) is searched by the literal ORACLE for sample k:
    score and number of optimal placements equal line k of placement_stats.tsv."""
    import ctypes as C
    import subprocess
    from tools.time_load import host_lib, write_workload, ptr
    from usher_amd import synth as gsynth
    n_new = 1000
    st = gsynth.SynthTree(1_000_000, n_sites=25000, seed=2)
    q = st.queries(n_new, seed=9, max_subst=3, n_lo=0, n_hi=3, iupac_hi=0)
    L = host_lib()
    L.uh_pb_to_arrays.argtypes = [C.c_char_p] + [C.c_void_p] * 8
    pb, vcf = str(tmp_path / "base.pb"), str(tmp_path / "new.vcf")
    write_workload(L, st, q, n_new, pb, vcf)

    def run(vcf_path, out, env, extra=()):
        os.makedirs(out, exist_ok=True)
        e = dict(os.environ)
        e.pop("USHER_AMD_MAX_TOUCHED", None)
        e.update(env)
        r = subprocess.run([EXE, "-i", pb, "-v", vcf_path, "-d", out] + list(extra), capture_output=True, text=True, timeout=3000, env=e)
        assert r.returncode == 0, r.stderr[-3000:]
        return r.stderr

    err = run(vcf, str(tmp_path / "device"), {"USHER_AMD_PROFILE": "1"})
    flat = [l for l in err.splitlines() if "tree -> arrays" in l and "times)" in l]
    assert flat and "(1 times)" in flat[0], err[-2000:]                       # one flattening for the whole run: the device mode ran
    run(vcf, str(tmp_path / "research"), {"USHER_AMD_MAX_TOUCHED": "0"})
    files = ("placement_stats.tsv", "final-tree.nh")
    got = {n: open(str(tmp_path / "device" / n)).read() for n in files}
    for n in files:
        assert got[n] == open(str(tmp_path / "research" / n)).read(), n
    stats = [l.split("\t") for l in got["placement_stats.tsv"].splitlines()]
    assert len(stats) == n_new
    assert sum(1 for l in stats if l[2] != "1") > 100                         # multi-way ties are common
    rng = np.random.default_rng(4)
    threads = os.cpu_count() or 1
    for k in sorted(rng.choice(np.arange(50, n_new), 20, replace=False).tolist()):
        vk, dk = str(tmp_path / ("first%d.vcf" % k)), str(tmp_path / ("step%d" % k))
        write_workload(L, st, q, k, None, vk)
        run(vk, dk, {}, ["-o", os.path.join(dk, "step.pb")])
        counts = (C.c_uint64 * 2)()
        assert L.uh_pb_to_arrays(os.path.join(dk, "step.pb").encode(), counts, None, None, None, None, None, None, None) == 0
        n, m = int(counts[0]), int(counts[1])
        assert n >= int(st.arrays["n"]) + k                                   # k new leaves (and the internal nodes of sibling placements)
        arr = {"n": n, "parent": np.zeros(n, np.int64), "mut_off": np.zeros(n + 1, np.int64), "mut_pos": np.zeros(m, np.int32),
               "mut_ref": np.zeros(m, np.int8), "mut_par": np.zeros(m, np.int8), "mut_nuc": np.zeros(m, np.int8)}
        assert L.uh_pb_to_arrays(os.path.join(dk, "step.pb").encode(), counts, ptr(arr["parent"]), ptr(arr["mut_off"]), ptr(arr["mut_pos"]), ptr(arr["mut_ref"]),
                                 ptr(arr["mut_par"]), ptr(arr["mut_nuc"]), None) == 0
        w = capi.OracleTree(arr).place_mt(gsynth.csr_sample(q, k), threads)
        assert [str(w["best"]), str(w["num_best"])] == stats[k][1:3], (k, w, stats[k][:3])
