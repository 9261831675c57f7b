"""GPU parity tests proper: the HIP path, called through the C ABI
(libusher_amd.so), against the oracle on the recorded reference fixtures and on
seeded random trees.  Bit-exact: placement node, score, tie count, sibling flag,
per-node scores and tie sets."""
import gzip
import os
import tempfile

import numpy as np
import pytest

from oracle import capi, refio
from tests import synth
from usher_amd import Placer, QueryBatch, UgpError

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SURVEY = os.path.join(GOLD, "survey_ref")


def _load(pb, vcf):
    T = refio.load_mutation_annotated_tree(pb)
    arrays = refio.tree_to_bfs_arrays(T)
    samples = [refio.sample_to_arrays(s) for s in refio.read_vcf(T, vcf)]
    return arrays, samples


def _read_stats(path):
    rows = []
    with open(path) as f:
        for line in f:
            w = line.rstrip("\n").split("\t")
            if len(w) >= 3:
                rows.append((w[0], int(w[1]), int(w[2])))
    return rows


def _assert_same(res, i, want, name=""):
    got = (int(res["best_set_difference"][i]), int(res["num_best"][i]), int(res["best_j"][i]), bool(res["best_has_unique"][i]))
    exp = (want["best"], want["num_best"], want["best_j"], want["has_unique"])
    assert got == exp, (name, got, exp)


@pytest.mark.parametrize("chunk_nodes", [None, 16, 3])
def test_global_fixture_matches_reference_and_oracle(chunk_nodes):
    arrays, samples = _load(os.path.join(SURVEY, "global", "global_assignments.pb"), os.path.join(GOLD, "ref_fixtures", "new_samples.vcf"))
    pl = Placer(arrays, chunk_nodes=chunk_nodes)
    res = pl.place(QueryBatch(samples))
    want = _read_stats(os.path.join(SURVEY, "global", "out3", "placement_stats.tsv"))
    assert [(int(r["best_set_difference"]), int(r["num_best"])) for r in res] == [(w[1], w[2]) for w in want]
    ot = capi.OracleTree(arrays)
    for i, s in enumerate(samples):
        _assert_same(res, i, ot.place(s), s["name"])
    pl.close()


def test_global_fixture_per_node_scores_match_reference():
    arrays, samples = _load(os.path.join(SURVEY, "global", "global_assignments.pb"), os.path.join(GOLD, "ref_fixtures", "new_samples.vcf"))
    pl = Placer(arrays, chunk_nodes=32)
    got = pl.scores_per_node(QueryBatch(samples))
    rows = {}
    with gzip.open(os.path.join(SURVEY, "global", "out4", "parsimony-scores.tsv.gz"), "rt") as f:
        next(f)
        for line in f:
            w = line.split("\t")
            rows.setdefault(w[0], []).append(int(w[2]))
    for i, s in enumerate(samples):
        assert got[i].tolist() == rows[s["name"]]
    ties, ties_hu, tc = pl.tied_nodes(QueryBatch(samples), cap=8)
    for i in range(len(samples)):
        assert [arrays["names"][j] for j in ties[i]] == ["node_7", "node_11"]
    pl.close()


def test_syn_fixture_scores_ties_and_stats():
    arrays, samples = _load(os.path.join(SURVEY, "syn", "tree.pb"), os.path.join(SURVEY, "syn", "query.vcf"))
    pl = Placer(arrays, chunk_nodes=64)
    batch = QueryBatch(samples)
    res = pl.place(batch)
    want = _read_stats(os.path.join(SURVEY, "syn", "o3", "placement_stats.tsv"))
    assert [(s["name"], int(r["best_set_difference"]), int(r["num_best"])) for s, r in zip(samples, res)] == want
    ot = capi.OracleTree(arrays)
    scores = pl.scores_per_node(batch)
    ties, ties_hu, tc = pl.tied_nodes(batch, cap=4096)
    for i, s in enumerate(samples):
        w = ot.place(s, compute_scores=True)
        _assert_same(res, i, w, s["name"])
        assert scores[i].tolist() == w["scores"].tolist()
        assert int(tc[i]) == w["num_best"]
        assert ties[i].tolist() == w["ties"].tolist()
        assert ties_hu[i].tolist() == w["ties_has_unique"].tolist()
    pl.close()


def test_big_fixture_matches_reference():
    with gzip.open(os.path.join(SURVEY, "big", "tree.pb.gz"), "rb") as f, tempfile.NamedTemporaryFile(suffix=".pb") as tmp:
        tmp.write(f.read())
        tmp.flush()
        arrays, samples = _load(tmp.name, os.path.join(SURVEY, "big", "query.vcf"))
    pl = Placer(arrays)
    res = pl.place(QueryBatch(samples))
    want = _read_stats(os.path.join(SURVEY, "big", "o3", "placement_stats.tsv"))
    assert [(s["name"], int(r["best_set_difference"]), int(r["num_best"])) for s, r in zip(samples, res)] == want
    ot = capi.OracleTree(arrays)
    for i in range(0, 64, 7):
        _assert_same(res, i, ot.place(samples[i]), samples[i]["name"])
    pl.close()


@pytest.mark.parametrize("seed,n_leaves,n_queries,chunk", [(31, 200, 70, 13), (32, 400, 130, None), (33, 900, 65, 50), (34, 60, 1, 4)])
def test_random_trees_match_oracle(seed, n_leaves, n_queries, chunk):
    arrays, queries = synth.make_case(seed, n_leaves=n_leaves, n_queries=n_queries, n_sites=80,
                                      p_masked=0.05 if seed % 2 else 0.0, root_muts=seed % 3)
    pl = Placer(arrays, chunk_nodes=chunk)
    res = pl.place(QueryBatch(queries))
    ot = capi.OracleTree(arrays)
    for i, s in enumerate(queries):
        _assert_same(res, i, ot.place(s), s["name"])
    scores = pl.scores_per_node(QueryBatch(queries[:5]))
    for i in range(min(5, len(queries))):
        assert scores[i].tolist() == ot.place(queries[i], compute_scores=True)["scores"].tolist()
    pl.close()


def test_edge_cases():
    # single-node tree, empty batch, empty sample, sample with only N rows
    one = {"n": 1, "parent": np.array([-1]), "mut_off": np.array([0, 0]), "mut_pos": np.zeros(0, np.int32),
           "mut_ref": np.zeros(0, np.int8), "mut_par": np.zeros(0, np.int8), "mut_nuc": np.zeros(0, np.int8)}
    empty = {"name": "e", "pos": np.zeros(0, np.int32), "ref": np.zeros(0, np.int8), "nuc": np.zeros(0, np.int8),
             "is_missing": np.zeros(0, np.int8)}
    # positions beyond the synthetic genome (2000) so REF cannot clash with a tree site
    only_n = {"name": "n", "pos": np.array([3003, 3009], np.int32), "ref": np.array([1, 2], np.int8), "nuc": np.array([15, 15], np.int8),
              "is_missing": np.array([1, 1], np.int8)}
    q = {"name": "q", "pos": np.array([3005], np.int32), "ref": np.array([1], np.int8), "nuc": np.array([4], np.int8),
         "is_missing": np.array([0], np.int8)}
    pl = Placer(one)
    assert len(pl.place(QueryBatch([]))) == 0
    res = pl.place(QueryBatch([empty, only_n, q]))
    ot = capi.OracleTree(one)
    for i, s in enumerate([empty, only_n, q]):
        _assert_same(res, i, ot.place(s), s["name"])
    pl.close()
    arrays, queries = synth.make_case(40, n_leaves=50, n_queries=3)
    pl = Placer(arrays)
    res = pl.place(QueryBatch([empty, only_n] + queries))
    ot = capi.OracleTree(arrays)
    for i, s in enumerate([empty, only_n] + queries):
        _assert_same(res, i, ot.place(s), s["name"])
    # unsorted / duplicated rows are refused loudly, not mis-scored
    bad = dict(q, pos=np.array([3009, 3005], np.int32), ref=np.array([1, 1], np.int8), nuc=np.array([4, 4], np.int8),
               is_missing=np.array([0, 0], np.int8))
    with pytest.raises(UgpError) as e:
        pl.place(QueryBatch([bad]))
    assert e.value.code == -2
    pl.close()


def test_usher_cli_on_gpu_matches_reference(tmp_path):
    """bin/usher-amd (C++ host + HIP backend through the C ABI) end to end on the in-tree fixture:
    default add-mode, -n and -p outputs identical to the recorded reference outputs."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "usher_amd", "bin", "usher-amd")
    pb = os.path.join(SURVEY, "global", "global_assignments.pb")
    vcf = os.path.join(GOLD, "ref_fixtures", "new_samples.vcf")

    def read(p):
        with (gzip.open(p, "rt") if p.endswith(".gz") else open(p)) as f:
            return f.read()

    for flags, sub, files in ((["-u"], "out2", ["placement_stats.tsv", "mutation-paths.txt", "uncondensed-final-tree.nh"]),
                              (["-n"], "out3", ["placement_stats.tsv", "final-tree.nh"]),
                              (["-p"], "out4", ["parsimony-scores.tsv", "current-tree.nh"])):
        d = tmp_path / sub
        d.mkdir()
        r = subprocess.run([exe, "-i", pb, "-v", vcf, "-d", str(d)] + flags, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        for name in files:
            want = os.path.join(SURVEY, "global", sub, name)
            if not os.path.exists(want):
                want += ".gz"
            assert read(str(d / name)) == read(want), (flags, name)


def test_usher_cli_builds_mat_on_gpu(tmp_path):
    """usher-amd -t global_phylo.nh -v global_samples.vcf -o g.pb: the Fitch-Sankoff assignment runs in
    ugp_fitch_sankoff; the written MAT and tree equal the reference's (and testBranchLen2's known answer)."""
    import shutil
    import subprocess
    from oracle import refio
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "usher_amd", "bin", "usher-amd")
    fix = os.path.join(GOLD, "ref_fixtures")
    vcf = str(tmp_path / "global_samples.vcf")
    with gzip.open(os.path.join(fix, "global_samples.vcf.gz"), "rb") as f, open(vcf, "wb") as o:
        shutil.copyfileobj(f, o)
    pb = str(tmp_path / "g.pb")
    r = subprocess.run([exe, "-t", os.path.join(fix, "global_phylo.nh"), "-v", vcf, "-o", pb, "-d", str(tmp_path)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]

    def sem(path):
        with open(path, "rb") as f:
            newick, muts, cond, meta = refio.parse_parsimony_pb(f.read())
        return newick, muts, sorted((k, tuple(v)) for k, v in cond), meta
    assert sem(pb) == sem(os.path.join(SURVEY, "global", "global_assignments.pb"))
    with open(str(tmp_path / "final-tree.nh")) as f, open(os.path.join(SURVEY, "global", "build-final-tree.nh")) as g:
        assert f.read() == g.read()


def test_locality_sort_and_pruning_paths_are_exact(monkeypatch):
    """Force the speed-only machinery (coarse-MAT locality sort, pruning records, work queues) onto a
    mid-size tree and a multi-tile batch; results must not change, with every switch combination."""
    arrays, queries = synth.make_case(71, n_leaves=6000, n_queries=1300, n_sites=500, n_ambig=(0, 0, 2, 5))
    ot = capi.OracleTree(arrays)
    want = [ot.place(s, want_ties=False) for s in queries]
    batch = QueryBatch(queries)
    for env in ({"UGP_COARSE_MIN_NODES": "0"}, {"UGP_COARSE_MIN_NODES": "0", "UGP_UNIT_CHUNKS": "1"},
                {"UGP_NO_SORT": "1"}, {"UGP_NO_PRUNE": "1"}, {"UGP_FORCE_V1": "1"}):
        for k in ("UGP_COARSE_MIN_NODES", "UGP_UNIT_CHUNKS", "UGP_NO_SORT", "UGP_NO_PRUNE", "UGP_FORCE_V1"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pl = Placer(arrays, chunk_nodes=300)
        res = pl.place(batch)
        for i, w in enumerate(want):
            _assert_same(res, i, w, "%s #%d" % (env, i))
        pl.close()


def test_full_size_properties_1m_nodes(monkeypatch):
    """Size-independent properties on a 1M-node synthetic MAT (too large for the oracle to sweep):
    the packed/pruned/sorted path, the plain 32-bit path and a permuted batch must agree sample by
    sample; a query equal to a tree node's genotype scores 0; and a sample of queries is checked
    against the oracle."""
    from usher_amd import synth as gsynth
    for k in ("UGP_FORCE_V1", "UGP_NO_SORT", "UGP_NO_PRUNE"):
        monkeypatch.delenv(k, raising=False)
    st = gsynth.SynthTree(1_000_000, n_sites=8000, seed=5)
    q = st.queries(3000, seed=9, max_subst=3, n_lo=0, n_hi=40, iupac_hi=4)
    q0 = st.queries(600, seed=10, max_subst=0)            # exact genotypes of tree nodes
    batch = QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
    batch0 = QueryBatch.from_csr(q0["ent_off"], q0["pos"], q0["ref"], q0["nuc"], q0["is_missing"])
    pl = Placer(st.arrays)
    assert pl.info()["n_nodes"] == 1_000_000
    fast = pl.place(batch)
    again = pl.place(batch)
    assert (fast.view(np.int32) == again.view(np.int32)).all()                      # idempotent
    perm = np.random.default_rng(3).permutation(len(batch))
    samples = [gsynth.csr_sample(q, int(i)) for i in perm]
    shuffled = pl.place(QueryBatch(samples))
    assert (shuffled.view(np.int32).reshape(-1, 4) == fast.view(np.int32).reshape(-1, 4)[perm]).all()   # order-independent
    z = pl.place(batch0)
    assert (z["best_set_difference"] == 0).all() and (z["num_best"] >= 1).all()     # self-placement costs nothing
    monkeypatch.setenv("UGP_FORCE_V1", "1")
    pl.reload_knobs()                                                               # (the switches are read when the handle is made)
    slow = pl.place(batch)                                                          # 32-bit, one sample per lane, no pruning
    monkeypatch.delenv("UGP_FORCE_V1")
    pl.reload_knobs()
    assert (slow.view(np.int32) == fast.view(np.int32)).all()
    # the work-unit machinery at its extremes: units that are never cut, units cut at every opportunity (thousands of
    # entries through the shared list, every wave waiting and exiting through it), one unit per side of a tile's ring
    for env in ({"UGP_SPLIT_CYCLES": "0"}, {"UGP_SPLIT_CYCLES": "1"}, {"UGP_LDS_BITS": "1"}, {"UGP_LDS_BITS": "1", "UGP_SPLIT_CYCLES": "1"}, {"UGP_SPLIT_CYCLES": "1", "UGP_UNIT_GROW": "1", "UGP_UNIT_MAX": "1000000"},
                {"UGP_UNIT_GROW": "0", "UGP_SPLIT_CYCLES": "5000"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pl.reload_knobs()
        other = pl.place(batch)
        for k in env:
            monkeypatch.delenv(k)
        pl.reload_knobs()
        assert (other.view(np.int32) == fast.view(np.int32)).all(), env
    ot = capi.OracleTree(st.arrays)
    for i in range(0, 3000, 500):
        w = ot.place_mt(gsynth.csr_sample(q, i), 8)
        assert (w["best"], w["num_best"], w["best_j"]) == (int(fast["best_set_difference"][i]), int(fast["num_best"][i]), int(fast["best_j"][i]))
    pl.close()


def test_consecutive_device_calls_share_the_device_and_stay_exact(monkeypatch):
    """ugp_place_device_overlapped overlaps consecutive calls on one handle (two internal streams, two sets of workspaces).  A
    sequence of calls on different query sets and output buffers, with no synchronisation in between and a host-buffer call
    (workspace set 0 on the default stream) thrown in, must give each batch its own exact answers; the summed timing
    reports every call; the stream-ordered ugp_place_device gives the same results on the caller's stream."""
    import torch
    from usher_amd import synth as gsynth
    st = gsynth.SynthTree(400_000, n_sites=5000, seed=21)
    qs = [st.queries(n, seed=50 + i, max_subst=3, n_lo=0, n_hi=20, iupac_hi=3) for i, n in enumerate((2100, 700, 4096))]
    batches = [QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"]) for q in qs]
    pl = Placer(st.arrays)
    want = [pl.place(b).view(np.int32).reshape(-1, 4).copy() for b in batches]
    cf = capi.ClosedFormC(capi.OracleTree(st.arrays))
    for q, w in zip(qs, want):
        c = cf.place_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
        assert (w[:, 0] == c["best"]).all() and (w[:, 1] == c["num_best"]).all() and (w[:, 2] == c["best_j"]).all()
    handles = [pl.upload(b) for b in batches]
    stream = torch.cuda.current_stream().cuda_stream
    first = torch.zeros((len(batches[1]), 4), dtype=torch.int32, device="cuda")
    pl.place_device_overlapped(handles[1], first.data_ptr(), stream)
    torch.cuda.synchronize()
    assert (first.cpu().numpy() == want[1]).all()
    pl.timing_sum()
    outs = []
    # (the output buffers are made up front: an overlapped call waits for what was on the caller's stream when the call BEFORE it
    # was made -- include/usher_amd.h -- a fill queued right in front of a call sits behind the completion of the call before it)
    bufs = [[torch.full((len(batches[i]), 4), -7, dtype=torch.int32, device="cuda") for i in (0, 1, 2, 1, 0)] for _ in range(4)]
    torch.cuda.synchronize()
    for rnd in range(4):
        for k, i in enumerate((0, 1, 2, 1, 0)):
            o = bufs[rnd][k]
            pl.place_device_overlapped(handles[i], o.data_ptr(), stream)
            outs.append((i, o))
        if rnd == 1:
            mid = pl.place(batches[2]).view(np.int32).reshape(-1, 4)     # host buffers, synchronous, in the middle of the pipeline
            assert (mid == want[2]).all()
    torch.cuda.synchronize()
    assert pl.timing_sum()["calls"] == 21
    for i, o in outs:
        assert (o.cpu().numpy() == want[i]).all(), i
    # as many output buffers as calls in flight, used in turn, each result consumed on the caller's stream before the next call is
    # made: the consumer of call k's buffer is finished before call k + depth overwrites it (depth - 1 calls of lag)
    depth = pl.pipeline_depth()                     # calls kept on the device at a time = output buffers to cycle through
    assert 2 <= depth <= 4
    two = [torch.zeros((max(len(b) for b in batches), 4), dtype=torch.int32, device="cuda") for _ in range(depth)]
    sums = []
    for k in range(40):
        i = (0, 1, 2)[k % 3]
        o = two[k % depth]
        pl.place_device_overlapped(handles[i], o.data_ptr(), stream)
        sums.append((i, o[:len(batches[i])].to(torch.int64).sum(0)))     # (queued on the caller's stream behind the call's completion)
    torch.cuda.synchronize()
    for i, sm in sums:
        assert (sm.cpu().numpy() == want[i].astype(np.int64).sum(0)).all(), i
    o = torch.zeros((len(batches[0]), 4), dtype=torch.int32, device="cuda")
    pl.place_device(handles[0], o.data_ptr(), stream)          # stream-ordered form
    torch.cuda.synchronize()
    assert (o.cpu().numpy() == want[0]).all()
    for h in handles:
        pl.free_qset(h)
    pl.close()


@pytest.mark.parametrize("seed,n_queries", [(61, 70), (62, 33), (63, 1), (64, 257)])
def test_per_node_scores_by_levels_and_by_depth_first_walk(seed, n_queries, monkeypatch):
    """ugp_scores_per_node (-p): the level-by-level kernel (k_scores_level: one thread per node, 32 samples per step, 4-bit
    counters with the plain-sum path for branches of more than 15 mutations) and the depth-first walk (UGP_SCORES_DFS) against
    the oracle's per-node scores -- long branches, masked mutations, root mutations, N / IUPAC cells, batches that are no
    multiple of 8 or 32 samples."""
    arrays, queries = synth.make_case(seed, n_leaves=700, n_queries=n_queries, n_sites=150, p_masked=0.04, root_muts=seed % 3,
                                      mut_counts=(0, 0, 1, 1, 1, 2, 3, 17, 40), n_ambig=(0, 0, 2, 5, 30))
    ot = capi.OracleTree(arrays)
    want = np.stack([ot.place(s, compute_scores=True)["scores"] for s in queries])
    batch = QueryBatch(queries)
    for env in ({}, {"UGP_SCORES_DFS": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pl = Placer(arrays, chunk_nodes=40)
        got = pl.scores_per_node(batch)
        pl.close()
        for k in env:
            monkeypatch.delenv(k)
        assert got.shape == want.shape and (got == want).all(), (env, np.argwhere(got != want)[:5])


def test_score_matrix_larger_than_the_staging_buffers():
    """-p at a size where the matrix (48 samples x 1M nodes x 4 bytes = 192 MB) leaves the device through the two pinned staging
    buffers of copy_d2h_staged (pieces of 32 MB moved to their place by host threads while the next piece is on the link) instead of
    one hipMemcpy: every row equals the C closed form, and equals the row of a one-sample call (a matrix below the threshold)."""
    from usher_amd import synth as gsynth
    st = gsynth.SynthTree(1_000_000, n_sites=8000, seed=5)
    q = st.queries(48, seed=9, max_subst=3, n_lo=0, n_hi=20, iupac_hi=3)
    batch = QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
    pl = Placer(st.arrays)
    n = pl.info()["n_nodes"]
    assert 48 * n * 4 >= 4 * (32 << 20)            # (the staged path's threshold)
    got = pl.scores_per_node(batch)
    cf = capi.ClosedFormC(capi.OracleTree(st.arrays))
    for i in range(48):
        assert (cf.scores(gsynth.csr_sample(q, i)) == got[i]).all(), i
    for i in (0, 17, 47):
        assert (pl.scores_per_node(batch.slice(i, i + 1))[0] == got[i]).all(), i
    pl.close()


def test_host_buffer_batches_two_in_flight():
    """ugp_place_batch_async / ugp_job_wait: batches from host buffers with as many in flight as the handle keeps on the device
    (ugp_pipeline_depth: three by default) give the answers of ugp_place_batch; the batch's arrays may be overwritten as soon as
    the call returns; one job more is refused until the oldest has been waited for; a batch with out-of-order rows is reported by
    ITS ugp_job_wait and leaves the handle usable."""
    from usher_amd import synth as gsynth
    from usher_amd.placement import UgpError
    st = gsynth.SynthTree(300_000, n_sites=4000, seed=33)
    qs = [st.queries(n, seed=70 + i, max_subst=3, n_lo=0, n_hi=10, iupac_hi=2) for i, n in enumerate((1500, 600, 2600, 40))]
    batches = [QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"]) for q in qs]
    pl = Placer(st.arrays)
    depth = pl.pipeline_depth()
    assert depth == 3
    want = [pl.place(b).copy() for b in batches]
    jobs = []
    got = {}
    order = [0, 1, 2, 3, 2, 1, 0, 3, 3, 0]
    for k, i in enumerate(order):
        b = batches[i]
        scratch = QueryBatch.from_csr(b.ent_off.copy(), b.pos.copy(), b.ref.copy(), b.nuc.copy(), b.is_missing.copy())
        jobs.append((k, i, pl.place_async(scratch)))
        scratch.pos[:] = 0; scratch.nuc[:] = 0     # the rows were copied out before the call returned
        if len(jobs) == depth:
            with pytest.raises(UgpError):           # one job more than the handle keeps in flight
                pl.place_async(batches[3])
            k0, i0, j0 = jobs.pop(0)
            got[k0] = (i0, pl.job_wait(j0))
    while jobs:
        k0, i0, j0 = jobs.pop(0)
        got[k0] = (i0, pl.job_wait(j0))
    assert len(got) == len(order)
    for k, (i, r) in got.items():
        assert (r.view(np.int32) == want[i].view(np.int32)).all(), (k, i)
    # a bad batch between two good ones
    bad = QueryBatch.from_csr(batches[1].ent_off.copy(), batches[1].pos.copy(), batches[1].ref.copy(), batches[1].nuc.copy(), batches[1].is_missing.copy())
    smp = int(np.flatnonzero(np.diff(bad.ent_off.astype(np.int64)) >= 2)[3])                     # a sample with at least two rows ...
    e0 = int(bad.ent_off[smp])                                                                   # ... gets them out of order
    for a in (bad.pos, bad.ref, bad.nuc, bad.is_missing):
        a[e0], a[e0 + 1] = a[e0 + 1], a[e0]
    ja = pl.place_async(batches[0]); jb = pl.place_async(bad)
    assert (pl.job_wait(ja).view(np.int32) == want[0].view(np.int32)).all()
    jc = pl.place_async(batches[2])
    with pytest.raises(UgpError) as ei:
        pl.job_wait(jb)
    assert ei.value.code == -2 and ("sample %d " % smp) in str(ei.value), str(ei.value)
    assert (pl.job_wait(jc).view(np.int32) == want[2].view(np.int32)).all()
    assert (pl.place(batches[3]).view(np.int32) == want[3].view(np.int32)).all()
    pl.close()


def test_tie_lists_from_phase_two_and_from_the_full_walk(monkeypatch):
    """ugp_tied_nodes: on the packed path the lists are written by phase 2 itself (k_ties<LIST>, from the chunks that attain the
    minimum); UGP_TIES_DFS keeps the second, one-sample-per-lane walk of the whole tree.  Both against the oracle's tie sets on a
    multi-tile batch with the locality sort on, and with a cap smaller than some tie sets (true counts, a subset of the ties)."""
    arrays, queries = synth.make_case(91, n_leaves=5000, n_queries=1100, n_sites=260, n_ambig=(0, 0, 2, 5, 30), p_masked=0.02)
    monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    ot = capi.OracleTree(arrays)
    want = [ot.place(s) for s in queries]
    batch = QueryBatch(queries)
    assert max(w["num_best"] for w in want) > 3
    for env in ({}, {"UGP_TIES_DFS": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pl = Placer(arrays, chunk_nodes=48)
        ties, hu, tc = pl.tied_nodes(batch, 64)
        ties2, hu2, tc2 = pl.tied_nodes(batch, 2)
        pl.close()
        for k in env:
            monkeypatch.delenv(k)
        for i, w in enumerate(want):
            assert int(tc[i]) == w["num_best"] == int(tc2[i]), (env, i)
            if w["num_best"] <= 64:
                assert ties[i].tolist() == w["ties"].tolist() and hu[i].tolist() == w["ties_has_unique"].tolist(), (env, i)
            full = dict(zip(w["ties"].tolist(), w["ties_has_unique"].tolist()))
            assert len(ties2[i]) == min(2, w["num_best"]) and all(full.get(int(j)) == bool(h) for j, h in zip(ties2[i], hu2[i])), (env, i)


def test_sub_batching_and_tiny_batches():
    """More than 262,144 samples in one call (the library splits into sub-batches) and batches smaller
    than one tile give the same per-sample answers."""
    arrays, queries = synth.make_case(81, n_leaves=120, n_queries=40, n_sites=50)
    ot = capi.OracleTree(arrays)
    want = [ot.place(s, want_ties=False) for s in queries]
    pl = Placer(arrays, chunk_nodes=20)
    for n in (1, 2, 63, 65):
        res = pl.place(QueryBatch([queries[i % 40] for i in range(n)]))
        for i in range(n):
            _assert_same(res, i, want[i % 40], "n=%d" % n)
    reps = 262144 // 40 + 30                       # 40 * reps > 262,144: two launch sequences
    big = QueryBatch(queries * reps)
    assert len(big) > 262144
    res = pl.place(big)
    w = np.array([[x["best"], x["num_best"], x["best_j"], int(x["has_unique"])] for x in want], dtype=np.int64)
    got = res.view(np.int32).reshape(-1, 4).astype(np.int64)
    assert (got == np.tile(w, (reps, 1))).all()
    pl.close()


def test_deep_caterpillar_tree_packed_and_fallback_paths():
    """Maximally deep trees.  (a) 700 levels x 2 mutations: root paths of 1,400 mutations -- D in the
    thousands, hsub beyond its 8-bit field near the top (no pruning record there), deep preambles; still
    inside the 16-bit packed path.  (b) 400 levels x 85 mutations: 34,000 mutations on a root path exceed
    what the packed counters can hold, so the library must take the 32-bit kernel.  Checked against the
    closed-form restatement (the literal oracle is quadratic in the path length) and, for (a), two samples
    against the literal oracle as well."""
    from oracle.closed_form import ClosedFormTree
    arrays, queries = synth.caterpillar_case(11, depth=700, muts_per_node=2, n_queries=70)
    cf = ClosedFormTree(arrays)
    for chunk_nodes in (None, 37):
        pl = Placer(arrays, chunk_nodes=chunk_nodes)
        res = pl.place(QueryBatch(queries))
        assert pl.timing()["packed_path"] == 1
        for i, s in enumerate(queries):
            _assert_same(res, i, cf.place(s), "caterpillar #%d" % i)
        pl.close()
    ot = capi.OracleTree(arrays)
    for i in (0, 1):
        _assert_same(res, i, ot.place(queries[i], want_ties=False), "caterpillar literal #%d" % i)

    arrays, queries = synth.caterpillar_case(12, depth=400, muts_per_node=85, n_queries=12)
    cf = ClosedFormTree(arrays)
    pl = Placer(arrays, chunk_nodes=50)
    res = pl.place(QueryBatch(queries))
    assert pl.timing()["packed_path"] == 0          # 16-bit counters would overflow: 32-bit kernel
    for i, s in enumerate(queries):
        _assert_same(res, i, cf.place(s), "very deep #%d" % i)
    pl.close()


def test_usher_cli_add_mode_batched_equals_research_on_gpu(tmp_path):
    """bin/usher-amd, default (sequential add) mode on a 3,000-node tree built by the GPU Fitch-Sankoff path:
    the batched scheme (answers re-derived on the changing tree) and a full search per
    sample (USHER_AMD_MAX_TOUCHED=0, the reference's loop) must write identical files."""
    import subprocess
    import numpy as np
    from tests.test_host_cli import _evolve_vcf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "usher_amd", "bin", "usher-amd")
    nh, old, new = str(tmp_path / "t.nh"), str(tmp_path / "old.vcf"), str(tmp_path / "new.vcf")
    _evolve_vcf(np.random.default_rng(21), 1500, 200, 350, nh, old, new)
    pb = str(tmp_path / "base.pb")
    r = subprocess.run([exe, "-t", nh, "-v", old, "-o", pb, "-d", str(tmp_path)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    outs = {}
    for mode, env in (("research", {"USHER_AMD_MAX_TOUCHED": "0"}), ("batched", {})):
        d = tmp_path / mode
        d.mkdir()
        e = dict(os.environ)
        e.pop("USHER_AMD_MAX_TOUCHED", None)
        e.update(env)
        r = subprocess.run([exe, "-i", pb, "-v", new, "-u", "-d", str(d)], capture_output=True, text=True, timeout=900, env=e)
        assert r.returncode == 0, r.stderr[-2000:]
        outs[mode] = {n: open(str(d / n)).read() for n in ("placement_stats.tsv", "mutation-paths.txt", "uncondensed-final-tree.nh")}
    assert outs["batched"] == outs["research"]
    assert sum(1 for l in outs["batched"]["placement_stats.tsv"].splitlines() if l.split("\t")[2] != "1") > 20   # ties were exercised


@pytest.mark.parametrize("seed", range(10))
def test_randomised_trees_and_scheduling_knobs(seed, monkeypatch):
    """Differential fuzz: random trees (masked mutations, long branches, root mutations), multi-tile batches
    with N / IUPAC cells, and random values of every speed-only knob (chunk size, unit sizes, bound-exchange
    period, locality sort, longest-first scheduling); results must equal the oracle's sample by sample."""
    rng = np.random.default_rng(1000 + seed)
    n_leaves = int(rng.integers(150, 2500))
    arrays, queries = synth.make_case(2000 + seed, n_leaves=n_leaves, n_queries=int(rng.integers(530, 1100)),
                                      n_sites=int(rng.integers(60, 400)), p_masked=float(rng.choice([0.0, 0.02])),
                                      root_muts=int(rng.integers(0, 3)), mut_counts=(0, 0, 1, 1, 1, 2, 3, int(rng.choice([3, 18]))),
                                      n_ambig=(0, 0, 2, 5, 30))
    ot = capi.OracleTree(arrays)
    want = [ot.place(s, want_ties=False) for s in queries]
    knobs = {"UGP_COARSE_MIN_NODES": "0", "UGP_UNIT_CHUNKS": str(int(rng.integers(1, 9))), "UGP_HEAVY_CHUNKS": str(int(rng.integers(1, 9))),
             "UGP_LDS_SLOTS": str(int(rng.integers(1, 10))),   # few LDS slots: the others go through the global scratch
             "UGP_UB_EVERY": str(int(rng.choice([1, 2, 7, 1000]))), "UGP_PRUNE_MIN_WORDS": str(int(rng.choice([2, 8, 40])))}
    if rng.random() < 0.3:
        knobs["UGP_NO_LPT"] = "1"
    if rng.random() < 0.2:
        knobs["UGP_NO_SEED"] = "1"
    if rng.random() < 0.3:
        knobs["UGP_NO_DESCENT"] = "1"        # seeds from the coarse pass only
    if rng.random() < 0.3:
        knobs["UGP_NO_BOUND2"] = "1"         # first lower bound only
    if rng.random() < 0.5:
        knobs["UGP_UNIT_CHUNKS"] = "16"
    if rng.random() < 0.7:   # units that grow with the distance from the tile's own region, and units cut while they run
        knobs["UGP_UNIT_GROW"] = str(int(rng.choice([0, 1, 3])))
        knobs["UGP_UNIT_MAX"] = str(int(rng.choice([2, 40, 100000])))
        knobs["UGP_SPLIT_CYCLES"] = str(int(rng.choice([0, 1, 3000, 400000])))   # 1: a unit is cut at every restart while anyone waits
    if rng.random() < 0.5:
        knobs["UGP_LDS_BITS"] = str(int(rng.integers(0, 2)))    # the kernel variant with the tiles' active-row bitmaps in LDS
    if rng.random() < 0.3:
        knobs["UGP_PRE_WEIGHT"] = "50"
    if rng.random() < 0.3:
        knobs["UGP_PHASE2_PACKED"] = "1"     # phase 2 as a mode of the packed walk (k_best8<TIES>) instead of k_ties
    if rng.random() < 0.4:
        knobs["UGP_NMASK"] = "1"             # tiles built from per-sample N bit masks (the path of batches with many missing rows)
    if rng.random() < 0.3:
        knobs["UGP_COARSE_PHASE2"] = "1"     # the coarse pass with its full phase 2 (the tie-break winner seeds the sort and the descent)
    if rng.random() < 0.3:
        knobs["UGP_RADIX_SORT"] = "1"        # the samples sorted by the device radix sort instead of the counting sort over the coarse nodes
    if rng.random() < 0.3:
        knobs["UGP_SPLIT_MANY"] = str(int(rng.choice([0, 1, 8])))   # how a running unit is cut: one half / as many pieces as waves wait
        knobs["UGP_SPLIT_MANY_HEAVY"] = str(int(rng.choice([0, 1, 5])))
    for k in ("UGP_COARSE_MIN_NODES", "UGP_UNIT_CHUNKS", "UGP_HEAVY_CHUNKS", "UGP_UB_EVERY", "UGP_PRUNE_MIN_WORDS", "UGP_NO_LPT", "UGP_NO_SEED",
              "UGP_NO_SORT", "UGP_NO_PRUNE", "UGP_FORCE_V1", "UGP_LDS_SLOTS", "UGP_NO_DESCENT", "UGP_NO_BOUND2", "UGP_UNIT_GROW", "UGP_UNIT_MAX", "UGP_SPLIT_CYCLES", "UGP_LDS_BITS", "UGP_PRE_WEIGHT", "UGP_COARSE_PHASE2", "UGP_NMASK", "UGP_PHASE2_PACKED", "UGP_RADIX_SORT", "UGP_SPLIT_MANY", "UGP_SPLIT_MANY_HEAVY"):
        monkeypatch.delenv(k, raising=False)
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    pl = Placer(arrays, chunk_nodes=int(rng.integers(3, 80)), experiments="UGP_PHASE2_PACKED" in knobs)   # (the experiment exists in libusher_amd_exp.so only)
    res = pl.place(QueryBatch(queries))
    for i, w in enumerate(want):
        _assert_same(res, i, w, "%s #%d" % (knobs, i))
    pl.close()


def test_huge_polytomies(monkeypatch):
    """Nodes with tens to thousands of children (root with 2,500), most leaves identical to their parent: the
    sibling records, the child ordering they rely on and the slot numbering are exercised at their extremes."""
    for k in ("UGP_COARSE_MIN_NODES",):
        monkeypatch.setenv(k, "0")
    for fan, nq in (((2500, 6, 4), 700), ((30, 50, 20), 900)):
        arrays, queries = synth.polytomy_case(31 + len(fan) + fan[0], fanouts=fan, n_queries=nq)
        ot = capi.OracleTree(arrays)
        want = [ot.place(s, want_ties=False) for s in queries]
        for chunk_nodes in (None, 11):
            pl = Placer(arrays, chunk_nodes=chunk_nodes)
            res = pl.place(QueryBatch(queries))
            for i, w in enumerate(want):
                _assert_same(res, i, w, "polytomy %s #%d" % (fan, i))
            pl.close()


# ---------------------------------------------------------------------------
# BASELINE.json configurations at their stated sizes
# ---------------------------------------------------------------------------

def _csr_batch(q):
    return QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])


def _rows(res):
    return np.stack([res["best_set_difference"].astype(np.int64), res["num_best"].astype(np.int64), res["best_j"].astype(np.int64),
                     res["best_has_unique"].astype(np.int64)], 1)


def _cf_rows(r):
    return np.stack([r["best"].astype(np.int64), r["num_best"].astype(np.int64), r["best_j"].astype(np.int64), r["has_unique"].astype(np.int64)], 1)


@pytest.mark.parametrize("coarse", [False, True])
def test_config2_100k_nodes_1k_queries_every_sample_vs_oracle(coarse, monkeypatch):
    """BASELINE config 2: 100k-node synthetic MAT, 1,024 SARS-CoV-2-length queries; every sample against the
    literal oracle (node-parallel pool) and the closed form, with the default knobs (no coarse pass at this
    size) and with the locality sort / seeded bounds forced on."""
    from usher_amd import synth as gsynth
    for k in ("UGP_FORCE_V1", "UGP_NO_SORT", "UGP_NO_PRUNE", "UGP_COARSE_MIN_NODES"):
        monkeypatch.delenv(k, raising=False)
    if coarse:
        monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    st = gsynth.SynthTree(100_000, n_sites=1500, seed=2)
    q = st.queries(1024, seed=77)
    pl = Placer(st.arrays)
    assert pl.info()["n_nodes"] >= 100_000
    res = pl.place(_csr_batch(q))
    assert pl.timing()["packed_path"] == 1
    ot = capi.OracleTree(st.arrays)
    cf = capi.ClosedFormC(ot).place_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
    assert (_rows(res) == _cf_rows(cf)).all()
    threads = os.cpu_count() or 1
    for i in range(1024):
        w = ot.place_mt(gsynth.csr_sample(q, i), threads)
        assert (w["best"], w["num_best"], w["best_j"]) == (int(res["best_set_difference"][i]), int(res["num_best"][i]), int(res["best_j"][i])), i
    pl.close()


def test_config3_10m_nodes_10k_queries(monkeypatch):
    """BASELINE config 3 at the size of the 10M-node substitute (the public SARS-CoV-2 MAT is not in the image):
    10,000 queries; the packed / pruned / sorted path, the plain 32-bit kernel and a permuted batch agree sample
    by sample; 2,048 samples against the closed form; 32 against the literal oracle."""
    from usher_amd import synth as gsynth
    for k in ("UGP_FORCE_V1", "UGP_NO_SORT", "UGP_NO_PRUNE", "UGP_COARSE_MIN_NODES"):
        monkeypatch.delenv(k, raising=False)
    st = gsynth.SynthTree(10_000_000, n_sites=25000, seed=1)
    q = st.queries(10_000, seed=4242)
    batch = _csr_batch(q)
    pl = Placer(st.arrays)
    assert pl.info()["n_nodes"] == 10_000_000
    fast = pl.place(batch)
    assert pl.timing()["packed_path"] == 1
    perm = np.random.default_rng(8).permutation(len(batch))
    shuffled = pl.place(QueryBatch([gsynth.csr_sample(q, int(i)) for i in perm]))
    assert (_rows(shuffled) == _rows(fast)[perm]).all()
    monkeypatch.setenv("UGP_FORCE_V1", "1")
    pl.reload_knobs()
    slow = pl.place(batch)
    assert pl.timing()["packed_path"] == 0
    monkeypatch.delenv("UGP_FORCE_V1")
    assert (_rows(slow) == _rows(fast)).all()
    pl.close()
    ot = capi.OracleTree(st.arrays)
    n_cf = 2048
    e1 = int(q["ent_off"][n_cf])
    cf = capi.ClosedFormC(ot).place_csr(q["ent_off"][:n_cf + 1], q["pos"][:e1], q["ref"][:e1], q["nuc"][:e1], q["is_missing"][:e1])
    assert (_rows(fast)[:n_cf] == _cf_rows(cf)).all()
    threads = os.cpu_count() or 1
    for i in range(5000, 5032):
        w = ot.place_mt(gsynth.csr_sample(q, i), threads)
        assert (w["best"], w["num_best"], w["best_j"]) == (int(fast["best_set_difference"][i]), int(fast["num_best"][i]), int(fast["best_j"][i])), i


def test_config5_high_ambiguity_results_and_tie_lists(monkeypatch):
    """BASELINE config 5 on one device: 100-5,000 N cells + 0-30 IUPAC cells per query on a 1M-node MAT;
    placements AND the lists of equally parsimonious nodes (ugp_tied_nodes) against the closed form for every
    sample, a handful against the literal oracle."""
    from usher_amd import synth as gsynth
    for k in ("UGP_FORCE_V1", "UGP_NO_SORT", "UGP_NO_PRUNE", "UGP_COARSE_MIN_NODES"):
        monkeypatch.delenv(k, raising=False)
    st = gsynth.SynthTree(1_000_000, n_sites=8000, seed=6)
    q = st.queries(1536, seed=55, max_subst=3, n_lo=100, n_hi=5000, iupac_hi=30)
    assert int(np.diff(q["ent_off"].astype(np.int64)).max()) > 3000
    batch = _csr_batch(q)
    pl = Placer(st.arrays)
    res = pl.place(batch)
    assert pl.timing()["packed_path"] == 1
    cap = 64
    ties, ties_hu, tc = pl.tied_nodes(batch, cap)
    pl.close()
    ot = capi.OracleTree(st.arrays)
    cf = capi.ClosedFormC(ot).place_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"], tie_cap=cap)
    assert (_rows(res) == _cf_rows(cf)).all()
    assert (tc.astype(np.int64) == cf["num_best"]).all()
    assert int((cf["num_best"] > 1).sum()) > 50            # ties are common with this much ambiguity
    for i in range(len(batch)):
        if cf["num_best"][i] <= cap:
            assert ties[i].tolist() == cf["ties"][i].tolist(), i
            assert ties_hu[i].tolist() == cf["ties_has_unique"][i].tolist(), i
    for i in range(0, 1536, 256):
        w = ot.place(gsynth.csr_sample(q, i), tie_cap=cap)
        _assert_same(res, i, w, "literal #%d" % i)
        if w["num_best"] <= cap:
            assert ties[i].tolist() == w["ties"].tolist()


def test_sars2_shape_10m_nodes_full_size(monkeypatch):
    """The SARS-CoV-2-shaped generator of `bench.py --shape sars2` (polytomies, recent-biased attachment, skewed
    site spectrum, queries next to recent leaves) at the headline size: packed path == 32-bit kernel == permuted
    batch for every sample, 1,536 samples against the closed form, 16 against the literal oracle."""
    from usher_amd import synth as gsynth
    for k in ("UGP_FORCE_V1", "UGP_NO_SORT", "UGP_NO_PRUNE", "UGP_COARSE_MIN_NODES"):
        monkeypatch.delenv(k, raising=False)
    st = gsynth.SynthTree(10_000_000, n_sites=25000, seed=3, shape="sars2")
    q = st.queries(8192, seed=99, recent=True)
    batch = _csr_batch(q)
    pl = Placer(st.arrays)
    assert pl.info()["n_nodes"] >= 9_000_000
    fast = pl.place(batch)
    assert pl.timing()["packed_path"] == 1
    perm = np.random.default_rng(9).permutation(len(batch))
    shuffled = pl.place(QueryBatch([gsynth.csr_sample(q, int(i)) for i in perm]))
    assert (_rows(shuffled) == _rows(fast)[perm]).all()
    monkeypatch.setenv("UGP_FORCE_V1", "1")
    pl.reload_knobs()
    slow = pl.place(batch)
    assert pl.timing()["packed_path"] == 0
    monkeypatch.delenv("UGP_FORCE_V1")
    assert (_rows(slow) == _rows(fast)).all()
    pl.close()
    ot = capi.OracleTree(st.arrays)
    n_cf = 1536
    e1 = int(q["ent_off"][n_cf])
    cf = capi.ClosedFormC(ot).place_csr(q["ent_off"][:n_cf + 1], q["pos"][:e1], q["ref"][:e1], q["nuc"][:e1], q["is_missing"][:e1])
    assert (_rows(fast)[:n_cf] == _cf_rows(cf)).all()
    threads = os.cpu_count() or 1
    for i in range(4000, 4016):
        w = ot.place_mt(gsynth.csr_sample(q, i), threads)
        assert (w["best"], w["num_best"], w["best_j"]) == (int(fast["best_set_difference"][i]), int(fast["num_best"][i]), int(fast["best_j"][i])), i


@pytest.mark.parametrize("coarse", [False, True])
def test_sixteen_bit_boundary_of_the_packed_path(coarse, monkeypatch):
    """Costs right below the largest value the packed 16-bit path admits (the shared upper bounds start at
    0x7F7F) and right above it (the library must fall back to the 32-bit kernel): samples with ~32,600
    mismatching rows at positions the tree never mutates, next to ordinary samples in the same tiles."""
    for k in ("UGP_FORCE_V1", "UGP_NO_SORT", "UGP_NO_PRUNE", "UGP_COARSE_MIN_NODES", "UGP_NO_SEED"):
        monkeypatch.delenv(k, raising=False)
    if coarse:
        monkeypatch.setenv("UGP_COARSE_MIN_NODES", "0")
    arrays, queries = synth.make_case(91, n_leaves=700, n_queries=600, n_sites=100)
    ot = capi.OracleTree(arrays)
    cfc = capi.ClosedFormC(ot)
    from usher_amd import FlatTreeView
    path = FlatTreeView(arrays).max_path_muts

    def fat(base, n_rows, name):
        keep = base["pos"] <= 2000                                        # the test genome has 2,000 bases
        k = int(keep.sum())
        n_extra = n_rows - k
        extra = np.arange(3001, 3001 + n_extra, dtype=np.int32)
        return {"name": name, "pos": np.concatenate([base["pos"][keep], extra]),
                "ref": np.concatenate([base["ref"][keep], np.full(n_extra, 1, np.int8)]),
                "nuc": np.concatenate([base["nuc"][keep], np.full(n_extra, 4, np.int8)]),
                "is_missing": np.concatenate([base["is_missing"][keep], np.zeros(n_extra, np.int8)])}

    for max_rows, want_packed in ((0x7F7F - 3 - path, 1), (0x7F7F - 2 - path, 0), (0x7FFE, 0)):
        big = [fat(queries[i], max_rows - 2 * i, "F%d" % i) for i in (0, 1, 2)]
        qs = queries[:300] + big[:1] + queries[300:] + big[1:]
        assert (max(len(s["pos"]) for s in qs) + path + 2 < 0x7F7F) == bool(want_packed)
        pl = Placer(arrays, chunk_nodes=40)
        res = pl.place(QueryBatch(qs))
        assert pl.timing()["packed_path"] == want_packed
        for i, s in enumerate(qs):
            _assert_same(res, i, cfc.place(s), "%s rows %d" % (s["name"], max_rows))
        pl.close()


def test_usher_cli_multiple_placements_on_gpu(tmp_path):
    """bin/usher-amd -M 4 on the recorded synthetic fixture (50 samples, many with several optimal nodes): every file
    equals what the python restatement of the reference's multi-tree loop produces with the oracle doing the searches."""
    import subprocess
    from tests import usher_model
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "usher_amd", "bin", "usher-amd")
    pb, vcf = os.path.join(SURVEY, "syn", "tree.pb"), os.path.join(SURVEY, "syn", "query.vcf")
    r = subprocess.run([exe, "-i", pb, "-v", vcf, "-M", "4", "-d", str(tmp_path)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    T = refio.load_mutation_annotated_tree(pb)
    want = usher_model.run(T, refio.read_vcf(T, vcf), max_trees=4)
    got = {n: open(str(tmp_path / n)).read() for n in sorted(os.listdir(str(tmp_path)))}
    assert sorted(got) == sorted(want)
    for name in want:
        assert got[name] == want[name], name
    assert "final-tree-4.nh" in got


# ---------------------------------------------------------------------------
# Round 4: BASELINE config 4's workload and config 3's size on one device; the boundary's ordering and threading rules
# ---------------------------------------------------------------------------

def test_config4_workload_one_million_queries_on_one_device(monkeypatch):
    """BASELINE config 4's workload (1,000,000 queries against the 10M-node MAT, the loop usher_common.cpp:310 shards) on ONE
    device, in one call: four sub-batches of 262,144 samples.  Size-independent properties: a permuted copy of the batch gives
    every sample the same answer (a sample's result depends neither on its tile, its sub-batch nor its neighbours); a 16,384
    slice placed on its own and through the plain 32-bit kernel agrees with its rows of the big call; 4,096 samples against the
    C closed form; 32 against the literal oracle."""
    from usher_amd import synth as gsynth
    for k in ("UGP_FORCE_V1", "UGP_NO_SORT", "UGP_NO_PRUNE", "UGP_COARSE_MIN_NODES"):
        monkeypatch.delenv(k, raising=False)
    st = gsynth.SynthTree(10_000_000, n_sites=25000, seed=1)
    Q = 1_000_000
    q = st.queries(Q, seed=31337)
    batch = _csr_batch(q)
    pl = Placer(st.arrays)
    big = pl.place(batch)
    tm = pl.timing()
    assert tm["packed_path"] == 1 and tm["place_launches"] >= 4          # sub-batches of at most 262,144 samples
    # the same samples in another order (each lands in another tile and, mostly, another sub-batch)
    perm = np.random.default_rng(11).permutation(Q)
    lens = np.diff(q["ent_off"].astype(np.int64))
    starts = q["ent_off"].astype(np.int64)[:-1]
    new_off = np.zeros(Q + 1, np.uint64); new_off[1:] = np.cumsum(lens[perm])
    idx = np.repeat(starts[perm] - new_off[:-1].astype(np.int64), lens[perm]) + np.arange(int(new_off[-1]))
    shuffled = pl.place(QueryBatch.from_csr(new_off, q["pos"][idx], q["ref"][idx], q["nuc"][idx], q["is_missing"][idx]))
    assert (_rows(shuffled) == _rows(big)[perm]).all()
    # a slice on its own, packed and 32-bit
    lo, hi = 500_000, 516_384
    part = batch.slice(lo, hi)
    alone = pl.place(part)
    assert (_rows(alone) == _rows(big)[lo:hi]).all()
    monkeypatch.setenv("UGP_FORCE_V1", "1")
    pl.reload_knobs()
    slow = pl.place(part)
    assert pl.timing()["packed_path"] == 0
    monkeypatch.delenv("UGP_FORCE_V1")
    assert (_rows(slow) == _rows(big)[lo:hi]).all()
    pl.close()
    ot = capi.OracleTree(st.arrays)
    n_cf, base = 4096, 300_000                                          # (a slice that straddles the second sub-batch boundary region)
    e0, e1 = int(q["ent_off"][base]), int(q["ent_off"][base + n_cf])
    cf = capi.ClosedFormC(ot).place_csr(q["ent_off"][base:base + n_cf + 1] - q["ent_off"][base], q["pos"][e0:e1], q["ref"][e0:e1], q["nuc"][e0:e1],
                                        q["is_missing"][e0:e1])
    assert (_rows(big)[base:base + n_cf] == _cf_rows(cf)).all()
    threads = os.cpu_count() or 1
    for i in list(range(262_140, 262_156)) + list(range(999_984, 1_000_000)):   # around a sub-batch boundary and at the very end
        w = ot.place_mt(gsynth.csr_sample(q, i), threads)
        assert (w["best"], w["num_best"], w["best_j"]) == (int(big["best_set_difference"][i]), int(big["num_best"][i]), int(big["best_j"][i])), i


def test_config3_size_15m_node_sars2_tree(monkeypatch):
    """BASELINE config 3's size: a 15M-node SARS-CoV-2-shaped MAT (the public .pb.gz is not in the image) x 10,000 queries next
    to recent leaves.  Packed path == 32-bit kernel == permuted batch for every sample, 2,048 samples against the closed form,
    16 against the literal oracle; the device footprint of the handle stays far below one GPU's 288 GB."""
    import torch
    from usher_amd import synth as gsynth
    for k in ("UGP_FORCE_V1", "UGP_NO_SORT", "UGP_NO_PRUNE", "UGP_COARSE_MIN_NODES"):
        monkeypatch.delenv(k, raising=False)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    st = gsynth.SynthTree(15_000_000, n_sites=25000, seed=7, shape="sars2")
    q = st.queries(10_000, seed=123, recent=True)
    batch = _csr_batch(q)
    pl = Placer(st.arrays)
    assert pl.info()["n_nodes"] == 15_000_000
    fast = pl.place(batch)
    assert pl.timing()["packed_path"] == 1
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 << 30, (free0 - free1)                    # streams + tables + one batch's workspaces: a few GB
    perm = np.random.default_rng(5).permutation(len(batch))
    shuffled = pl.place(QueryBatch([gsynth.csr_sample(q, int(i)) for i in perm]))
    assert (_rows(shuffled) == _rows(fast)[perm]).all()
    monkeypatch.setenv("UGP_FORCE_V1", "1")
    pl.reload_knobs()
    slow = pl.place(batch)
    assert pl.timing()["packed_path"] == 0
    monkeypatch.delenv("UGP_FORCE_V1")
    assert (_rows(slow) == _rows(fast)).all()
    pl.close()
    ot = capi.OracleTree(st.arrays)
    n_cf = 2048
    e1 = int(q["ent_off"][n_cf])
    cf = capi.ClosedFormC(ot).place_csr(q["ent_off"][:n_cf + 1], q["pos"][:e1], q["ref"][:e1], q["nuc"][:e1], q["is_missing"][:e1])
    assert (_rows(fast)[:n_cf] == _cf_rows(cf)).all()
    threads = os.cpu_count() or 1
    for i in range(7000, 7016):
        w = ot.place_mt(gsynth.csr_sample(q, i), threads)
        assert (w["best"], w["num_best"], w["best_j"]) == (int(fast["best_set_difference"][i]), int(fast["num_best"][i]), int(fast["best_j"][i])), i


def test_place_device_is_stream_ordered():
    """ugp_place_device behaves like a kernel launch on the caller's stream: work queued there before the call (a fill of the
    output buffer) is finished before the call writes, and work queued behind it (a copy of the results) sees them -- 100 times
    in a row with no host synchronisation, on a side stream."""
    import torch
    from usher_amd import synth as gsynth
    st = gsynth.SynthTree(300_000, n_sites=4000, seed=44)
    qs = [st.queries(n, seed=90 + i, max_subst=3, n_lo=0, n_hi=10, iupac_hi=2) for i, n in enumerate((1800, 600))]
    batches = [_csr_batch(q) for q in qs]
    pl = Placer(st.arrays)
    want = [pl.place(b).view(np.int32).reshape(-1, 4).copy() for b in batches]
    handles = [pl.upload(b) for b in batches]
    side = torch.cuda.Stream()
    out = torch.zeros((1800, 4), dtype=torch.int32, device="cuda")
    copies = []
    torch.cuda.synchronize()
    with torch.cuda.stream(side):
        for k in range(100):
            i = k & 1
            out.fill_(-7)                                          # queued on `side` in front of the call
            pl.place_device(handles[i], out.data_ptr(), side.cuda_stream)
            copies.append((i, out[:len(batches[i])].clone()))       # queued behind it
    side.synchronize()
    for i, c in copies:
        assert (c.cpu().numpy() == want[i]).all(), i
    for h in handles:
        pl.free_qset(h)
    pl.close()


def test_two_handles_two_threads_while_the_environment_changes(monkeypatch):
    """The tuning switches are read when a handle is made: two handles driven by two host threads give exact answers on the packed
    path while a third thread keeps calling setenv on switches that would change the path (UGP_FORCE_V1) or the schedule -- no
    placement call reads the environment."""
    import ctypes
    import threading
    from usher_amd import synth as gsynth
    for k in ("UGP_FORCE_V1", "UGP_UNIT_CHUNKS", "UGP_NO_SORT"):
        monkeypatch.delenv(k, raising=False)
    st = gsynth.SynthTree(300_000, n_sites=4000, seed=45)
    q = st.queries(1500, seed=3, max_subst=3, n_lo=0, n_hi=10, iupac_hi=2)
    batch = _csr_batch(q)
    pls = [Placer(st.arrays), Placer(st.arrays)]
    want = pls[0].place(batch).view(np.int32).copy()
    libc = ctypes.CDLL(None)
    stop = threading.Event()
    errs = []

    def churn():
        n = 0
        while not stop.is_set():
            for k in (b"UGP_FORCE_V1", b"UGP_UNIT_CHUNKS", b"UGP_NO_SORT", b"UGP_FILLER_%d" % (n % 64)):
                libc.setenv(k, b"%d" % (1 + n % 7), 1)
            libc.unsetenv(b"UGP_FORCE_V1")
            n += 1

    def work(pl):
        try:
            for _ in range(30):
                got = pl.place(batch).view(np.int32)
                assert (got == want).all()
                assert pl.timing()["packed_path"] == 1
        except BaseException as e:   # noqa: BLE001
            errs.append(e)

    tc = threading.Thread(target=churn)
    tc.start()
    ts = [threading.Thread(target=work, args=(pl,)) for pl in pls]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    stop.set()
    tc.join()
    for k in (b"UGP_FORCE_V1", b"UGP_UNIT_CHUNKS", b"UGP_NO_SORT"):
        libc.unsetenv(k)
    for pl in pls:
        pl.close()
    assert not errs, errs


def test_async_batches_with_bad_rows_are_reported_and_never_placed():
    """ugp_place_batch_async queues the whole pipeline before the row check's verdict is read; a batch that fails the check is
    built as if it had no rows (k_scatter_entries / k_build_tiles look at the verdict), so no kernel sees duplicate positions, a
    bad allele mask or a REF mismatch; ITS ugp_job_wait reports the error, and the batches around it get their exact answers."""
    from usher_amd import synth as gsynth
    from usher_amd.placement import UgpError
    st = gsynth.SynthTree(300_000, n_sites=4000, seed=46)
    qa, qb = st.queries(1500, seed=1, max_subst=3, n_lo=0, n_hi=10), st.queries(900, seed=2, max_subst=3, n_lo=0, n_hi=200)
    good_a, good_b = _csr_batch(qa), _csr_batch(qb)
    pl = Placer(st.arrays)
    want_a, want_b = pl.place(good_a).copy(), pl.place(good_b).copy()

    def corrupt(kind):
        b = QueryBatch.from_csr(good_b.ent_off.copy(), good_b.pos.copy(), good_b.ref.copy(), good_b.nuc.copy(), good_b.is_missing.copy())
        smp = int(np.flatnonzero(np.diff(b.ent_off.astype(np.int64)) >= 3)[5])
        e0 = int(b.ent_off[smp])
        if kind == "duplicate":
            b.pos[e0 + 1] = b.pos[e0]
        elif kind == "mask":
            b.is_missing[e0] = 0; b.nuc[e0] = 0
        elif kind == "ref":
            tree_pos = np.unique(st.arrays["mut_pos"][st.arrays["mut_pos"] > 0])
            k = int(np.flatnonzero(np.isin(b.pos, tree_pos))[0])          # a row at a position the tree mutates: its REF is checked
            b.ref[k] = 1 if b.ref[k] != 1 else 2
        return b
    for kind, code in (("duplicate", -2), ("mask", -1), ("ref", -2)):
        j1 = pl.place_async(good_a)
        j2 = pl.place_async(corrupt(kind))
        assert (pl.job_wait(j1).view(np.int32) == want_a.view(np.int32)).all(), kind
        j3 = pl.place_async(good_b)
        with pytest.raises(UgpError) as ei:
            pl.job_wait(j2)
        assert ei.value.code == code, (kind, str(ei.value))
        assert (pl.job_wait(j3).view(np.int32) == want_b.view(np.int32)).all(), kind
    assert (pl.place(good_a).view(np.int32) == want_a.view(np.int32)).all()
    pl.close()
