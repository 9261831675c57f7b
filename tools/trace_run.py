#!/usr/bin/env python3
"""One traced k_best8 launch of the default workload (10M-node synthetic MAT, 16 384 queries) for tools/analysis/unit_trace.py:
    UGP_STATS=1 UGP_TRACE=gpurun_out/trace.bin python3 tools/trace_run.py [nodes] [queries] [ambiguous]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usher_amd import Placer, QueryBatch, synth

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
kw = dict(n_lo=100, n_hi=5000, iupac_hi=30) if len(sys.argv) > 3 and sys.argv[3] == "ambiguous" else {}
st = synth.SynthTree(nodes, n_sites=25000 if nodes >= 1_000_000 else 1500, seed=1)
pl = Placer(st.arrays, experiments=True)
q = st.queries(Q, seed=1017, **kw)
h = pl.upload(QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"]))
out = torch.zeros((Q, 4), dtype=torch.int32, device="cuda")
for _ in range(3):
    pl.place_device(h, out.data_ptr())
torch.cuda.synchronize()
print(pl.timing())
