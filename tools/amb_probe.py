#!/usr/bin/env python3
"""Which part of a high-ambiguity query set (BASELINE config 5) costs the time: the N runs or the IUPAC cells?
10M-node synthetic MAT, 16 384 queries, four query sets; prints ms per call (calls back to back, no overlap)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usher_amd import Placer, QueryBatch, synth

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
st = synth.SynthTree(nodes, n_sites=25000 if nodes >= 1_000_000 else 1500, seed=1)
pl = Placer(st.arrays)
out = torch.zeros((Q, 4), dtype=torch.int32, device="cuda")
for name, kw in (("plain", {}), ("N runs only", dict(n_lo=100, n_hi=5000)), ("IUPAC only", dict(iupac_hi=30)),
                 ("both", dict(n_lo=100, n_hi=5000, iupac_hi=30))):
    q = st.queries(Q, seed=1017, **kw)
    h = pl.upload(QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"]))
    for _ in range(3):
        pl.place_device(h, out.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        pl.place_device(h, out.data_ptr())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    t = pl.timing()
    print("%-12s rows %9d  %.3f ms per call   tables %.3f  coarse pass %.3f  k_best8 %.3f  phase 2 %.3f ms" % (name, len(q["pos"]), dt * 1e3, t["table_ms"], t["coarse_ms"], t["place_ms"], t["merge_ms"]), flush=True)
