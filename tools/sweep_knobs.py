#!/usr/bin/env python3
"""Sweep of the speed-only knobs in ONE process (tree and queries generated once):
    python tools/sweep_knobs.py "UGP_UNIT_CHUNKS=16" "UGP_CHUNK_NODES=200 UGP_UNIT_CHUNKS=12" ...
Every line carries the number of samples whose result differs from the default configuration's (must be 0;
the default itself is checked against the closed-form oracle on 256 samples).  Options: --nodes --queries --ambiguous."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("settings", nargs="*")
ap.add_argument("--nodes", type=int, default=10_000_000)
ap.add_argument("--queries", type=int, default=16384)
ap.add_argument("--ambiguous", action="store_true")
ap.add_argument("--shape", default="random")
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
import torch  # noqa: E402
from oracle import capi  # noqa: E402
from usher_amd import Placer, QueryBatch, synth  # noqa: E402

st = synth.SynthTree(args.nodes, n_sites=25000 if args.nodes >= 1_000_000 else 1500, seed=1, shape=args.shape)
kw = dict(n_lo=100, n_hi=5000, iupac_hi=30) if args.ambiguous else {}
if args.shape == "sars2":
    kw["recent"] = True
q = st.queries(args.queries, seed=1017, **kw)
batch = QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
out = torch.zeros((len(batch), 4), dtype=torch.int32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
BASE_ENV = dict(os.environ)   # every UGP_* switch that is not in here belongs to a previous setting and is removed before the next
ref = None
for setting in ["(default)"] + args.settings:
    for k in [k for k in os.environ if k.startswith("UGP_") and k not in BASE_ENV]:
        del os.environ[k]
    for k, v in BASE_ENV.items():
        if k.startswith("UGP_"):
            os.environ[k] = v
    if setting != "(default)":
        for kv in setting.split():
            k, v = kv.split("=")
            os.environ[k] = v
    t0 = time.time()
    pl = Placer(st.arrays)
    t_create = time.time() - t0
    qs = pl.upload(batch)
    pl.place_device(qs, out.data_ptr(), stream)
    torch.cuda.synchronize()
    acc = {"coarse_ms": 0.0, "table_ms": 0.0, "place_ms": 0.0, "merge_ms": 0.0}
    t0 = time.perf_counter()
    for _ in range(args.steps):
        pl.place_device(qs, out.data_ptr(), stream)
        tm = pl.timing()
        for k in acc:
            acc[k] += tm[k] / args.steps
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / args.steps
    res = out.cpu().numpy().copy()
    if ref is None:
        ref = res
        n = min(256, len(batch))
        e1 = int(q["ent_off"][n])
        cf = capi.ClosedFormC(capi.OracleTree(st.arrays)).place_csr(q["ent_off"][:n + 1], q["pos"][:e1], q["ref"][:e1], q["nuc"][:e1], q["is_missing"][:e1])
        bad = int(((res[:n, 0] != cf["best"]) | (res[:n, 1] != cf["num_best"]) | (res[:n, 2] != cf["best_j"])).sum())
    else:
        bad = int((res != ref).any(1).sum())
    print("%-60s %8.3f ms/step  %9.0f placements/s | coarse %.3f table %.3f best8 %.3f phase2 %.3f | create %.1fs | mismatches %d"
          % (setting, ms, len(batch) / ms * 1e3, acc["coarse_ms"], acc["table_ms"], acc["place_ms"], acc["merge_ms"], t_create, bad), flush=True)
    pl.free_qset(qs)
    pl.close()
