"""Oracle checks at the sizes bench.py quotes: BASELINE config 5 and the extended searches of the other
mapper2_body callers on the 10M-node tree of the headline, through the C ABI, against the C closed form and the literal
restatement of the reference (oracle/ugp_oracle.c: orc_place_sample, orc_place_sample_list).  One tree, one oracle for the module."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

from oracle import capi
from usher_amd import Placer, QueryBatch
from usher_amd import synth as gsynth

pytestmark = pytest.mark.gpu

KNOBS = ("UGP_FORCE_V1", "UGP_NO_SORT", "UGP_NO_PRUNE", "UGP_COARSE_MIN_NODES", "UGP_EX_SLOW")


@pytest.fixture(scope="module")
def big():
    for k in KNOBS:
        os.environ.pop(k, None)
    st = gsynth.SynthTree(10_000_000, n_sites=25000, seed=1)        # bench.py's tree
    ot = capi.OracleTree(st.arrays)
    yield st, ot


def _batch(q):
    return QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])


def _rows(res):
    return np.stack([res["best_set_difference"].astype(np.int64), res["num_best"].astype(np.int64), res["best_j"].astype(np.int64),
                     res["best_has_unique"].astype(np.int64)], 1)


def _pool(fn, items):
    # (ctypes calls release the GIL: the literal oracle is one thread per call, a 10M-node search takes seconds)
    with ThreadPoolExecutor(max_workers=max(1, min(len(items), (os.cpu_count() or 2) - 1))) as ex:
        return list(ex.map(fn, items))


def test_config5_at_10m_nodes_results_and_tie_lists(big, monkeypatch):
    """BASELINE config 5's workload at the size bench.py quotes it on: 16,384 queries with 100-5,000 N cells and 0-30 IUPAC cells each on
    the 10M-node MAT.  Results and tie lists (cap 64) of 2,048 samples against the C closed form, 16 against the literal oracle
    (usher_mapper.cpp:209-211, 294-296: missing and ambiguous cells), packed == the 32-bit one-sample-per-lane kernel on a 4,096
    slice, tie counts == num_best for every sample."""
    st, ot = big
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    q = st.queries(16384, seed=1004, n_lo=100, n_hi=5000, iupac_hi=30)     # (bench.py's extra key draws seed 1004 too)
    assert int(np.diff(q["ent_off"].astype(np.int64)).max()) > 3000
    batch = _batch(q)
    pl = Placer(st.arrays)
    res = pl.place(batch)
    assert pl.timing()["packed_path"] == 1
    cap = 64
    ties, ties_hu, tc = pl.tied_nodes(batch, cap)
    assert (tc.astype(np.int64) == res["num_best"].astype(np.int64)).all()
    lo, hi = 6000, 10096
    part = batch.slice(lo, hi)
    monkeypatch.setenv("UGP_FORCE_V1", "1")
    pl.reload_knobs()
    slow = pl.place(part)
    assert pl.timing()["packed_path"] == 0
    monkeypatch.delenv("UGP_FORCE_V1")
    pl.reload_knobs()
    assert (_rows(slow) == _rows(res)[lo:hi]).all()
    pl.close()
    n_cf, base = 2048, 7000
    e0, e1 = int(q["ent_off"][base]), int(q["ent_off"][base + n_cf])
    cf = capi.ClosedFormC(ot).place_csr(q["ent_off"][base:base + n_cf + 1] - q["ent_off"][base], q["pos"][e0:e1], q["ref"][e0:e1], q["nuc"][e0:e1],
                                        q["is_missing"][e0:e1], tie_cap=cap)
    got = _rows(res)[base:base + n_cf]
    want = np.stack([cf["best"].astype(np.int64), cf["num_best"].astype(np.int64), cf["best_j"].astype(np.int64), cf["has_unique"].astype(np.int64)], 1)
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    assert int((cf["num_best"] > 1).sum()) > 200                          # ties are common with this much ambiguity
    for i in range(n_cf):
        if cf["num_best"][i] <= cap:
            assert ties[base + i].tolist() == cf["ties"][i].tolist(), i
            assert ties_hu[base + i].tolist() == cf["ties_has_unique"][i].tolist(), i
    picks = list(range(0, 16384, 1024))
    lit = _pool(lambda i: ot.place(gsynth.csr_sample(q, i), tie_cap=cap), picks)
    for i, w in zip(picks, lit):
        assert (int(res["best_set_difference"][i]), int(res["num_best"][i]), int(res["best_j"][i]), bool(res["best_has_unique"][i])) == \
               (w["best"], w["num_best"], w["best_j"], w["has_unique"]), i
        if w["num_best"] <= cap:
            assert ties[i].tolist() == w["ties"].tolist() and ties_hu[i].tolist() == w["ties_has_unique"].tolist(), i


def test_extended_searches_at_10m_nodes(big, monkeypatch):
    """The four call styles of tools/bench_ex.py on the 10M-node tree, 4,096 samples each -- ripples (ripples/main.cpp:343-377: nodes with
    enough descendant leaves, a per-node distance), annotate (matUtils/annotate.cpp:615-638: depth-first indices), merge
    (matUtils/merge.cpp:253-280: a level-capped subtree) and uncertainty (matUtils/uncertainty.cpp:212-235: the sample's own node left
    out, depth-first indices): the packed pruned path == the one-sample-per-lane kernel (UGP_EX_SLOW=1) for EVERY sample, results and
    tie lists; 32 samples per style against the oracle's restatement of the call sites (orc_place_sample_list)."""
    st, ot = big
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    nq = 4096
    n = int(st.arrays["n"])
    par = np.asarray(st.arrays["parent"]).astype(np.int64)
    leaves = np.zeros(n, np.int64)
    has_child = np.zeros(n, bool)
    has_child[par[1:]] = True
    leaves[~has_child] = 1
    # (a breadth-first numbering: parent[] ascends, a level is an index range; children fold into their parents deepest level first)
    assert (np.diff(par[1:]) >= 0).all()
    levels, hi = [], 1
    while hi < n:
        nxt = int(np.searchsorted(par[1:], hi, "left")) + 1     # first node whose parent is not in the levels so far
        levels.append((hi, nxt))
        hi = nxt
    for b0, b1 in reversed(levels):
        np.add.at(leaves, par[b0:b1], leaves[b0:b1])
    mask = (leaves >= 10).astype(np.uint8)
    mask[0] = 1
    dist = np.random.default_rng(3).integers(0, 4, n).astype(np.uint32)
    q = st.queries(nq, seed=77)
    q_own = st.queries(nq, seed=78, max_subst=0)        # each sample = the mutation set of a tree node (its source)
    own = np.asarray(q_own["source"]).astype(np.int64)
    batch_q, batch_own = _batch(q), _batch(q_own)
    pl = Placer(st.arrays)
    dfs = pl.node_order("dfs").astype(np.int64)
    pos_of = np.empty(n, np.int64); pos_of[dfs] = np.arange(n)
    m_levels = pl.subtree_mask(0, 12)
    allj = np.arange(n, dtype=np.int64)
    root_muts = int(st.arrays["mut_off"][1])
    styles = [("ripples", dict(order="bfs", node_mask=mask, distance=dist), batch_q, q),
              ("annotate", dict(order="dfs"), batch_q, q),
              ("merge", dict(order="bfs", node_mask=m_levels), batch_q, q),
              ("uncertainty", dict(order="dfs", skip_node=pos_of[own].astype(np.uint32)), batch_own, q_own)]
    fast = {}
    for name, kw, batch, _ in styles:
        r = pl.place_ex(batch, **kw)
        assert pl.timing()["packed_path"] == 1, name
        fast[name] = (r.copy(),) + tuple(pl.tied_nodes_ex(batch, 64, **kw))
    # the node-level options prepared once (ugp_ex_prepare): the same answers call after call, and the per-call time of the ripples
    # style without the host's O(N) preparation (43 ms per 4,096 samples -> a few ms)
    import time
    for name, kw, batch, _ in styles:
        node_kw = {k: v for k, v in kw.items() if k != "skip_node"}
        ex = pl.prepare_ex(**node_kw)
        pl.place_prepared(batch, ex, skip_node=kw.get("skip_node"))
        dt = 1e9
        for _ in range(2):   # (the faster of two: a hiccup of the box is not what the limit below is about)
            t0 = time.perf_counter()
            got = pl.place_prepared(batch, ex, skip_node=kw.get("skip_node"))
            dt = min(dt, time.perf_counter() - t0)
        assert pl.timing()["packed_path"] == 1, name
        assert (got.view(np.int32) == fast[name][0].view(np.int32)).all(), name
        print("prepared %-12s %.2f ms per %d samples" % (name, dt * 1e3, nq))
        assert dt < 0.030, (name, dt)                         # (the one-shot ripples-style call is 40+ ms)
        pl.free_ex(ex)
    monkeypatch.setenv("UGP_EX_SLOW", "1")
    pl.reload_knobs()
    for name, kw, batch, _ in styles:
        r = pl.place_ex(batch, **kw)
        assert pl.timing()["packed_path"] == 0, name
        tj, th, tc = pl.tied_nodes_ex(batch, 64, **kw)
        a = fast[name]
        bad = np.flatnonzero((a[0].view(np.int32) != r.view(np.int32)).reshape(nq, -1).any(axis=1))
        assert len(bad) == 0, (name, bad[:10], a[0][bad[:3]], r[bad[:3]])
        assert (a[3] == tc).all(), name
        assert all(x.tolist() == y.tolist() for x, y in zip(a[1], tj)) and all(x.tolist() == y.tolist() for x, y in zip(a[2], th)), name
    monkeypatch.delenv("UGP_EX_SLOW")
    pl.close()
    # ---- 32 per style against the literal restatement of the call sites (one thread per call, all calls at once)
    picks = list(range(5, nq, nq // 32))[:32]
    n_ripples = np.flatnonzero(mask).astype(np.int64)
    n_merge = np.flatnonzero(m_levels).astype(np.int64)
    d_ripples = dist[n_ripples].astype(np.int64)

    def lit(job):
        name, i = job
        if name == "ripples":
            return ot.place_list(gsynth.csr_sample(q, i), n_ripples, jidx=n_ripples, distance=d_ripples, tie_cap=64)
        if name == "annotate":
            return ot.place_list(gsynth.csr_sample(q, i), dfs, jidx=allj, tie_cap=64)
        if name == "merge":
            return ot.place_list(gsynth.csr_sample(q, i), n_merge, jidx=n_merge, tie_cap=64)
        keep = dfs != own[i]
        s = gsynth.csr_sample(q_own, i)
        return ot.place_list(s, dfs[keep], jidx=allj[keep], init_best=len(s["pos"]) + root_muts + 1, tie_cap=64)

    jobs = [(name, i) for name in ("ripples", "annotate", "merge", "uncertainty") for i in picks]
    for (name, i), w in zip(jobs, _pool(lit, jobs)):
        r, tj, th, tc = fast[name]
        if w["num_best"] == 0 or w["best"] >= 10 ** 9:
            assert int(r["num_best"][i]) == 0, (name, i)
            continue
        assert (int(r["best_set_difference"][i]), int(r["num_best"][i]), int(r["best_j"][i]), bool(r["best_has_unique"][i])) == \
               (w["best"], w["num_best"], w["best_j"], w["has_unique"]), (name, i, r[i], w)
        assert int(tc[i]) == w["num_best"], (name, i)
        if w["num_best"] <= 64:
            assert tj[i].tolist() == w["ties"].tolist() and th[i].tolist() == w["ties_has_unique"].tolist(), (name, i)


def test_far_queries_at_10m_nodes(big, monkeypatch):
    """Round 6: queries that are NOT near any node -- a random node's genotype + 50-200 substitutions, every 8th the
    all-reference sample (no rows at all) -- the batch bench.py's `far_queries` key times: bounds decide little, the seeds are loose,
    whole tiles sit at the root.  512 samples against the C closed form (incl. every all-reference one among them), 16 against the
    literal oracle; with and without the phase-2 shortcut and the third bound the same answers."""
    st, ot = big
    for k in KNOBS:
        monkeypatch.delenv(k, raising=False)
    q = st.queries(16384, seed=9001, max_subst=200, min_subst=50, ref_every_8th=True)      # (bench.py far_pruned_frac draws seed 9001 too)
    rows = np.diff(q["ent_off"].astype(np.int64))
    assert (rows[7::8] == 0).all() and rows[rows > 0].min() >= 20
    batch = _batch(q)
    pl = Placer(st.arrays)
    res = pl.place(batch)
    assert pl.timing()["packed_path"] == 1
    for knobs in ({"UGP_NO_UNIQ": "1"}, {"UGP_BOUND3": "0"}, {"UGP_BOUND3": "1", "UGP_FORK": "1"}):
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        pl.reload_knobs()
        again = pl.place(batch)
        for k in knobs:
            monkeypatch.delenv(k)
        assert (_rows(again) == _rows(res)).all(), knobs
    pl.reload_knobs()
    pl.close()
    n_cf, base = 512, 4096
    e0, e1 = int(q["ent_off"][base]), int(q["ent_off"][base + n_cf])
    cf = capi.ClosedFormC(ot).place_csr(q["ent_off"][base:base + n_cf + 1] - q["ent_off"][base], q["pos"][e0:e1], q["ref"][e0:e1], q["nuc"][e0:e1],
                                        q["is_missing"][e0:e1])
    got = _rows(res)[base:base + n_cf]
    want = np.stack([cf["best"].astype(np.int64), cf["num_best"].astype(np.int64), cf["best_j"].astype(np.int64), cf["has_unique"].astype(np.int64)], 1)
    assert (got == want).all(), np.flatnonzero((got != want).any(axis=1))[:10]
    picks = list(range(3, 16384, 1024)) + [7, 8199]          # (7, 8199: all-reference samples)
    lit = _pool(lambda i: ot.place(gsynth.csr_sample(q, i)), picks)
    for i, w in zip(picks, lit):
        assert (int(res["best_set_difference"][i]), int(res["num_best"][i]), int(res["best_j"][i]), bool(res["best_has_unique"][i])) == \
               (w["best"], w["num_best"], w["best_j"], w["has_unique"]), i


def test_second_handle_of_a_process_keeps_its_overlap(big, monkeypatch):
    """Round 6: a process has four hardware queues by default; a handle's three streams and the caller's use them up.  With one more
    stream per workspace set (the side stream of a lone call, now opt-in: UGP_FORK=1) the SECOND handle of a process ran its overlapped
    calls one after the other -- 13.9 -> 9.6 M placements/s on this very workload (tools/probe_context.py).  Two handles, one after the
    other, the same batches through ugp_place_device_overlapped: the second must not be much slower than the first (and both exact)."""
    import time
    import torch
    st, _ = big
    for k in KNOBS + ("UGP_FORK",):
        monkeypatch.delenv(k, raising=False)
    dev = torch.device("cuda", torch.cuda.current_device())
    stream = torch.cuda.current_stream().cuda_stream
    sets = [st.queries(16384, seed=4242 + r) for r in range(3)]
    rates, answers = [], []
    for which in range(2):
        if which == 1:
            # ... and a Fitch-Sankoff call in between, whose upload stream (cells in pieces) is pooled: the library drops it when the next
            # handle is about to make its streams (ugp::fitch_drop_streams) -- an idle stream would shift their hardware queues
            from usher_amd.fitch import fitch_sankoff
            monkeypatch.setenv("UGP_FITCH_PIECES", "2")
            par = np.concatenate([[-1], (np.arange(1, 600) - 1) // 2]).astype(np.int64)
            leaves = np.arange(300, 600, dtype=np.uint32)
            off = np.arange(0, 1201).astype(np.uint64) * 2   # (1 200 sites = three tiles, two cells each)
            cells = np.tile(leaves[:2], 1200)
            site, node, _, _ = fitch_sankoff(par, np.full(1200, 1, np.uint8), off, cells, np.full(len(cells), 2, np.uint8))
            assert len(site) > 0
            monkeypatch.delenv("UGP_FITCH_PIECES")
        pl = Placer(st.arrays)
        hq = [pl.upload(_batch(q)) for q in sets]
        dd = pl.pipeline_depth()
        out = [torch.zeros((16384, 4), dtype=torch.int32, device=dev) for _ in range(dd)]
        for k in range(9):
            pl.place_device_overlapped(hq[k % 3], out[k % dd].data_ptr(), stream)
        torch.cuda.synchronize()
        n, best = 30, 0.0
        for _ in range(3):   # (the best of three windows: a hiccup of the box is not what this test is about)
            t = time.perf_counter()
            for k in range(n):
                pl.place_device_overlapped(hq[k % 3], out[k % dd].data_ptr(), stream)
            torch.cuda.synchronize()
            best = max(best, 16384 * n / (time.perf_counter() - t))
        rates.append(best)
        answers.append(out[(n - 1) % dd].cpu().numpy().copy())
        for h in hq:
            pl.free_qset(h)
        pl.close()
    assert (answers[0] == answers[1]).all()
    assert rates[1] > 0.78 * rates[0], rates   # (the regression this guards against was 0.69)
