#!/usr/bin/env python3
"""Does a handle's rate depend on what the process did before?  (round 6: bench.py's config-3 key dropped from 7.1 to 4.7 M/s while the same
workload run alone gives 6.8.)  The SARS-CoV-2-shaped 15 M-node tree x 10 000 queries, three sets rotating, overlapped calls: first in a fresh
process, then again after a 10 M-node handle has lived and died in between, then a third time.   python tools/probe_context.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from usher_amd import Placer, QueryBatch, synth

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def run(pl, st, nq, steps, warm, n_rot=3, **kw):
    hqs = []
    for r in range(n_rot):
        q = st.queries(nq, seed=1004 + 104729 * r, **kw)
        hqs.append(pl.upload(QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])))
    dd = pl.pipeline_depth()
    oo = [torch.zeros((nq, 4), dtype=torch.int32, device=dev) for _ in range(dd)]
    for k in range(warm):
        pl.place_device_overlapped(hqs[k % n_rot], oo[k % dd].data_ptr(), stream)
    torch.cuda.synchronize()
    pl.timing_sum()
    t = time.perf_counter()
    for k in range(steps):
        pl.place_device_overlapped(hqs[k % n_rot], oo[k % dd].data_ptr(), stream)
    torch.cuda.synchronize()
    t = time.perf_counter() - t
    tm = pl.timing_sum()
    for h in hqs:
        pl.free_qset(h)
    return round(nq * steps / t / 1e6, 3), round(t * 1e3 / steps, 3), round(tm["place_ms"] / max(1, tm["calls"]), 3)


st3 = synth.SynthTree(15_000_000, n_sites=25000, seed=1, shape="sars2")
for label in ("fresh process", "after a 10 M-node handle lived and died", "third time"):
    pl3 = Placer(st3.arrays)
    print(label, "| sars2 15M x 10000:", run(pl3, st3, 10_000, 24, 24, recent=True), "free/total GB", [round(x / 2**30, 1) for x in torch.cuda.mem_get_info(dev)], flush=True)
    pl3.close()
    if label == "fresh process":
        st = synth.SynthTree(10_000_000, n_sites=25000, seed=1)
        pl = Placer(st.arrays)
        print("   headline 10M x 16384:", run(pl, st, 16384, 40, 10, n_rot=4), flush=True)
        print("   1M queries per call:", run(pl, st, 1_000_000, 3, 3, n_rot=2), flush=True)
        pl.close()
        del st
