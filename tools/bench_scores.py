#!/usr/bin/env python3
"""-p / ugp_scores_per_node at scale: samples x nodes int32 matrix in slabs.  python tools/bench_scores.py [nodes] [samples]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402
from usher_amd import Placer, QueryBatch, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 128
st = synth.SynthTree(N, n_sites=25000 if N >= 1_000_000 else 1500, seed=1)
q = st.queries(Q, seed=31)
batch = QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
pl = Placer(st.arrays)
n = pl.info()["n_nodes"]
pl.scores_per_node(batch.slice(0, 2))   # warm
t0 = time.perf_counter()
sc = pl.scores_per_node(batch)
dt = time.perf_counter() - t0
tm = pl.timing()
cells = Q * n
print("scores_per_node: %d samples x %d nodes: %.3f s wall (%.1f samples/s, %.2f GB/s of int32 output incl. PCIe); kernel %.3f ms = %.1f GB/s of output written"
      % (Q, n, dt, Q / dt, cells * 4 / dt / 1e9, tm["place_ms"], cells * 4 / (tm["place_ms"] * 1e-3) / 1e9))
# k_scores_level's algorithmic bytes per (node, sample): 4 out + 2 D read (16-bit D) + 2 D written by the nodes that have children
internal = len(np.unique(st.arrays["parent"][1:]))
algo = cells * 6.0 + internal * Q * 2.0
print("roofline (HBM, level-by-level kernel): %.2f GB algorithmic per call / %.3f ms = %.0f GB/s = %.1f %% of 8 TB/s"
      % (algo / 1e9, tm["place_ms"], algo / (tm["place_ms"] * 1e-3) / 1e9, algo / (tm["place_ms"] * 1e-3) / 8e12 * 100))
ot = capi.OracleTree(st.arrays)
cf = capi.ClosedFormC(ot)
for i in (0, Q // 2, Q - 1):
    assert (cf.scores(synth.csr_sample(q, i)) == sc[i]).all(), i
print("3 rows equal the closed form")
