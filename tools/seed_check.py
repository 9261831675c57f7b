#!/usr/bin/env python3
"""How good are the walk's seeds?  Places one batch twice on the instrumented build with UGP_STATS=1 UGP_SEED_CHECK=1: the second call
compares, per sample, the seed descent's cost (and the coarse pass's) with the first call's exact answer and prints the histogram of
the excess (stderr of the library).   python tools/seed_check.py [--nodes 10000000] [--queries 16384] [--shape sars2]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--queries", type=int, default=16384)
    ap.add_argument("--shape", default=None)
    a = ap.parse_args()
    os.environ["UGP_STATS"] = "1"
    os.environ["UGP_SEED_CHECK"] = "1"
    import torch   # (its HIP runtime first, as in bench.py)
    from usher_amd import Placer, QueryBatch, synth
    st = synth.SynthTree(a.nodes, n_sites=25000 if a.nodes >= 1_000_000 else 1500, seed=1, **({"shape": a.shape} if a.shape else {}))
    pe = Placer(st.arrays, device=0, experiments=True)
    qq = st.queries(a.queries, seed=9001)
    b = QueryBatch.from_csr(qq["ent_off"], qq["pos"], qq["ref"], qq["nuc"], qq["is_missing"])
    qs = pe.upload(b)   # (a resident set: the check compares calls on ONE set)
    out = torch.empty((a.queries, 4), dtype=torch.int32, device="cuda:0")
    for _ in range(3):
        pe.place_device(qs, out.data_ptr(), 0)
        torch.cuda.synchronize()
    pe.free_qset(qs)
    pe.close()


if __name__ == "__main__":
    main()
