#!/usr/bin/env python3
"""How much of a step's idle time can a second, independent step fill?  Two handles on the same device, one stream each,
the same 10M-node workload; aggregate placements/s against a single handle (tuning probe for cross-batch overlap)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usher_amd import Placer, QueryBatch, synth

nodes = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
Q = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
st = synth.SynthTree(nodes, n_sites=25000 if nodes >= 1_000_000 else 1500, seed=1)
qa = st.queries(Q, seed=1017)
qb = st.queries(Q, seed=2017)
ba = QueryBatch.from_csr(qa["ent_off"], qa["pos"], qa["ref"], qa["nuc"], qa["is_missing"])
bb = QueryBatch.from_csr(qb["ent_off"], qb["pos"], qb["ref"], qb["nuc"], qb["is_missing"])
pa, pb = Placer(st.arrays), Placer(st.arrays)
ha, hb = pa.upload(ba), pb.upload(bb)
oa = torch.zeros((Q, 4), dtype=torch.int32, device="cuda")
ob = torch.zeros((Q, 4), dtype=torch.int32, device="cuda")
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
K = 20
def run(n_handles):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        pa.place_device(ha, oa.data_ptr(), sa.cuda_stream)
        if n_handles == 2:
            pb.place_device(hb, ob.data_ptr(), sb.cuda_stream)
    torch.cuda.synchronize()
    return time.perf_counter() - t0
run(2); run(1)
t1 = run(1); t2 = run(2)
print("one handle : %.3f ms per step, %.2f M placements/s" % (t1 / K * 1e3, Q * K / t1 / 1e6))
print("two handles: %.3f ms per pair of steps, %.2f M placements/s aggregate" % (t2 / K * 1e3, 2 * Q * K / t2 / 1e6))
