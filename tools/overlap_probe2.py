#!/usr/bin/env python3
"""One handle, consecutive ugp_place_device calls (pipelined inside the library), no per-step synchronisation."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usher_amd import Placer, QueryBatch, synth
st = synth.SynthTree(10_000_000, n_sites=25000, seed=1)
Q = 16384
qa = st.queries(Q, seed=1017)
ba = QueryBatch.from_csr(qa["ent_off"], qa["pos"], qa["ref"], qa["nuc"], qa["is_missing"])
pa = Placer(st.arrays)
ha = pa.upload(ba)
oa = torch.zeros((Q, 4), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
K = 20
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = 0.0
    for _ in range(K):
        t1 = time.perf_counter()
        pa.place_device(ha, oa.data_ptr(), s)
        th += time.perf_counter() - t1
    torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print("%.3f ms per step, %.2f M placements/s; host time inside the call %.3f ms per step" % (t / K * 1e3, Q * K / t / 1e6, th / K * 1e3))
