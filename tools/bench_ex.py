#!/usr/bin/env python3
"""Timings of the extended searches (the other callers of mapper2_body, include/usher_amd.h ugp_place_batch_ex) at 10M nodes:
ripples-style (nodes with at least k descendant leaves, a per-node distance) and annotate-style (depth-first indices), on the packed
pruned path and on the one-sample-per-lane kernel (UGP_EX_SLOW=1), results compared.
    python tools/bench_ex.py [--nodes 10000000] [--queries 4096]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from usher_amd import Placer, QueryBatch, synth   # noqa: E402

import torch   # (first: torch's lazy device initialisation fails once another library has brought the runtime up)
torch.cuda.init()
ap = argparse.ArgumentParser()
ap.add_argument("--nodes", type=int, default=10_000_000)
ap.add_argument("--queries", type=int, default=4096)
ap.add_argument("--min-leaves", type=int, default=10)
a = ap.parse_args()
st = synth.SynthTree(a.nodes, n_sites=25000 if a.nodes >= 1_000_000 else 1500, seed=1)
q = st.queries(a.queries, seed=77)
batch_q = QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
par = np.asarray(st.arrays["parent"]).astype(np.int64)
n = int(st.arrays["n"])
leaves = np.zeros(n, np.int64)
has_child = np.zeros(n, bool)
has_child[par[1:]] = True
leaves[~has_child] = 1
for j in range(n - 1, 0, -1):          # (parents before children in the numbering: one reverse pass)
    leaves[par[j]] += leaves[j]
mask = (leaves >= a.min_leaves).astype(np.uint8)
mask[0] = 1
dist = np.random.default_rng(3).integers(0, 4, n).astype(np.uint32)
print("tree %d nodes, %d admitted by the mask (>= %d leaves below), %d queries" % (n, int(mask.sum()), a.min_leaves, a.queries), flush=True)
# uncertainty-style: every sample is the mutation set of a tree node, and that node is left out of its search (uncertainty.cpp:216)
q_own = st.queries(a.queries, seed=78, max_subst=0)
batch_own = QueryBatch.from_csr(q_own["ent_off"], q_own["pos"], q_own["ref"], q_own["nuc"], q_own["is_missing"])
own = np.asarray(q_own["source"]).astype(np.int64)
res = {}
for label, env in (("packed", None), ("one sample per lane", "1")):
    if env:
        os.environ["UGP_EX_SLOW"] = env
    else:
        os.environ.pop("UGP_EX_SLOW", None)
    pl = Placer(st.arrays)
    dfs = pl.node_order("dfs").astype(np.int64)
    pos_of = np.empty(n, np.int64); pos_of[dfs] = np.arange(n)
    for name, kw in (("ripples-style (mask + distance)", dict(order="bfs", node_mask=mask, distance=dist)), ("annotate-style (depth-first indices)", dict(order="dfs")),
                     ("merge-style (root subtree, 12 levels)", dict(order="bfs", node_mask=pl.subtree_mask(0, 12))),
                     ("uncertainty-style (own node left out)", dict(order="dfs", skip_node=pos_of[own].astype(np.uint32)))):
        batch = batch_own if "skip_node" in kw else batch_q
        pl.place_ex(batch, **kw)            # warm (allocations, the depth-first order and its rank)
        t0 = time.perf_counter()
        r = pl.place_ex(batch, **kw)
        dt = time.perf_counter() - t0
        t0 = time.perf_counter()
        tj, th, tc = pl.tied_nodes_ex(batch, 64, **kw)
        dt2 = time.perf_counter() - t0
        res.setdefault(name, []).append(r.copy())
        print("%-22s %-40s place_ex %8.1f ms = %9.0f samples/s   tied_nodes_ex %8.1f ms   (packed_path=%d)" % (label, name, dt * 1e3, a.queries / dt, dt2 * 1e3, pl.timing()["packed_path"]), flush=True)
    if not env:
        # the node-level options prepared once (ugp_ex_prepare): what a caller with ONE node vector for a whole run pays per call
        for name, kw in (("ripples-style (mask + distance)", dict(order="bfs", node_mask=mask, distance=dist)), ("annotate-style (depth-first indices)", dict(order="dfs")),
                         ("merge-style (root subtree, 12 levels)", dict(order="bfs", node_mask=pl.subtree_mask(0, 12)))):
            t0 = time.perf_counter(); ex = pl.prepare_ex(**kw); tp = time.perf_counter() - t0
            pl.place_prepared(batch_q, ex)
            t0 = time.perf_counter()
            for _ in range(5):
                rp = pl.place_prepared(batch_q, ex)
            dt = (time.perf_counter() - t0) / 5
            same = bool((rp.view(np.int32) == res[name][0].view(np.int32)).all())
            print("%-22s %-40s prepare %7.1f ms once, then place_prepared %7.2f ms = %9.0f samples/s   (== one-shot call: %s)" % ("prepared", name, tp * 1e3, dt * 1e3, a.queries / dt, same), flush=True)
            if name.startswith("ripples"):   # ... and with the score matrix, into a device buffer: 64 samples x N scores per call
                import torch
                nb = 64
                e1 = int(q["ent_off"][nb])
                some = QueryBatch.from_csr(q["ent_off"][:nb + 1], q["pos"][:e1], q["ref"][:e1], q["nuc"][:e1], q["is_missing"][:e1])
                d = torch.zeros((nb, n), dtype=torch.int32, device="cuda")
                pl.place_prepared(some, ex, d_scores=d.data_ptr())
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5):
                    pl.place_prepared(some, ex, d_scores=d.data_ptr())
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / 5
                print("%-22s %-40s place_prepared + scores on the device %7.2f ms per %d samples = %9.0f samples/s (%.1f GB of scores per s)" % ("prepared", name, dt * 1e3, nb, nb / dt, nb * n * 4 / dt / 1e9), flush=True)
            pl.free_ex(ex)
    # ripples proper: the score of every admitted node for one pruned sample at a time (ripples/main.cpp:343-377) -- 8 samples here
    few = QueryBatch.from_csr(q["ent_off"][:9], q["pos"][:int(q["ent_off"][8])], q["ref"][:int(q["ent_off"][8])], q["nuc"][:int(q["ent_off"][8])], q["is_missing"][:int(q["ent_off"][8])])
    pl.place_ex(few, order="bfs", node_mask=mask, distance=dist, want_scores=True)
    t0 = time.perf_counter()
    r8, sc8 = pl.place_ex(few, order="bfs", node_mask=mask, distance=dist, want_scores=True)
    dt = time.perf_counter() - t0
    res.setdefault("ripples-style with the score matrix (8 samples)", []).append(np.concatenate([r8.view(np.int32).ravel(), sc8.ravel()[::997]]))
    print("%-22s %-40s place_ex %8.1f ms for 8 samples x %d scores (packed_path=%d)" % (label, "ripples-style + scores", dt * 1e3, n, pl.timing()["packed_path"]), flush=True)
    pl.close()
for name, (x, y) in res.items():
    print("%-40s identical results on both paths: %s" % (name, bool((x.view(np.int32) == y.view(np.int32)).all())))
