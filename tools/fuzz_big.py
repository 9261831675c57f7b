#!/usr/bin/env python3
"""Large-tree differential fuzz: the pruning bounds, the preamble records and the seed descent only come into play on big
trees, so this runs the packed path on 1-3 M-node synthetic MATs (both shapes, random batch sizes / ambiguity / knobs) and
compares EVERY sample with the C closed form of the oracle (test infrastructure; the literal oracle on a few samples).
    python tools/fuzz_big.py 8      # seeds 0..7, ~1 min each on the GPU box"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402
from usher_amd import Placer, QueryBatch, synth  # noqa: E402

KNOBS = ("UGP_UNIT_CHUNKS", "UGP_HEAVY_CHUNKS", "UGP_UB_EVERY", "UGP_NO_LPT", "UGP_NO_DESCENT", "UGP_NO_BOUND2", "UGP_SPLIT_CYCLES", "UGP_UNIT_GROW", "UGP_UNIT_MAX", "UGP_LDS_BITS", "UGP_BOUND3", "UGP_NO_UNIQ", "UGP_FORK", "UGP_LIGHT_ORDER",
         "UGP_DESCENT_LANES", "UGP_LDS_SLOTS", "UGP_NO_PAD_FIX", "UGP_COARSE_DIV", "UGP_COARSE_PHASE2", "UGP_NMASK", "UGP_PHASE2_PACKED", "UGP_RADIX_SORT", "UGP_SPLIT_MANY")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    bad = 0
    for seed in range(n):
        rng = np.random.default_rng(77_000 + seed)
        shape = "sars2" if seed % 2 else "random"
        nodes = int(rng.choice([1_000_000, 2_000_000, 3_000_000]))
        nq = int(rng.integers(700, 9000))
        for k in KNOBS:
            os.environ.pop(k, None)
        knobs = {}
        if rng.random() < 0.5:
            knobs["UGP_UNIT_CHUNKS"] = str(int(rng.choice([4, 8, 16, 32])))
        if rng.random() < 0.3:
            knobs["UGP_HEAVY_CHUNKS"] = str(int(rng.choice([4, 16, 32])))
        for k, p, v in (("UGP_UB_EVERY", 0.3, "4"), ("UGP_NO_LPT", 0.2, "1"), ("UGP_NO_DESCENT", 0.2, "1"), ("UGP_NO_BOUND2", 0.15, "1"),
                        ("UGP_SPLIT_CYCLES", 0.5, str(int(rng.choice([0, 1, 5000, 100000])))), ("UGP_UNIT_GROW", 0.4, str(int(rng.choice([0, 1, 8])))), ("UGP_UNIT_MAX", 0.3, str(int(rng.choice([32, 100000])))), ("UGP_LDS_BITS", 0.5, str(int(rng.integers(0, 3)))), ("UGP_BOUND3", 0.7, str(int(rng.integers(0, 2)))), ("UGP_NO_UNIQ", 0.25, "1"), ("UGP_FORK", 0.3, "1"), ("UGP_LIGHT_ORDER", 0.4, str(int(rng.integers(0, 2)))), ("UGP_DESCENT_LANES", 0.4, str(int(rng.choice([16, 64])))),
                        ("UGP_LDS_SLOTS", 0.3, str(int(rng.integers(3, 12)))), ("UGP_NO_PAD_FIX", 0.15, "1"), ("UGP_COARSE_DIV", 0.3, str(int(rng.choice([256, 4096])))), ("UGP_COARSE_PHASE2", 0.2, "1"), ("UGP_NMASK", 0.3, str(int(rng.integers(0, 2)))), ("UGP_PHASE2_PACKED", 0.25, "1"), ("UGP_RADIX_SORT", 0.3, "1"),
                        ("UGP_SPLIT_MANY", 0.3, str(int(rng.choice([0, 1, 8]))))):
            if rng.random() < p:
                knobs[k] = v
        os.environ.update(knobs)
        t0 = time.time()
        st = synth.SynthTree(nodes, n_sites=int(rng.choice([8000, 25000])), seed=int(rng.integers(1, 1 << 20)), shape=shape)
        amb = rng.random() < 0.4
        q = st.queries(nq, seed=int(rng.integers(1, 1 << 20)), max_subst=int(rng.integers(0, 6)), n_lo=50 if amb else 0, n_hi=3000 if amb else int(rng.integers(0, 4)),
                       iupac_hi=20 if amb else 0, recent=(shape == "sars2" and rng.random() < 0.7))
        batch = QueryBatch.from_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
        pl = Placer(st.arrays, experiments="UGP_PHASE2_PACKED" in knobs)
        res = pl.place(batch)
        packed = pl.timing()["packed_path"]
        pl.close()
        ot = capi.OracleTree(st.arrays)
        cf = capi.ClosedFormC(ot).place_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
        got = np.stack([res["best_set_difference"].astype(np.int64), res["num_best"].astype(np.int64), res["best_j"].astype(np.int64), res["best_has_unique"].astype(np.int64)], 1)
        want = np.stack([cf["best"].astype(np.int64), cf["num_best"].astype(np.int64), cf["best_j"].astype(np.int64), cf["has_unique"].astype(np.int64)], 1)
        diff = int((got != want).any(axis=1).sum())
        lit = 0
        for i in rng.choice(nq, size=3, replace=False):
            w = ot.place_mt(synth.csr_sample(q, int(i)), os.cpu_count() or 1)
            lit += (w["best"], w["num_best"], w["best_j"]) != (int(res["best_set_difference"][i]), int(res["num_best"][i]), int(res["best_j"][i]))
        print("seed %d %s %d nodes %d queries packed=%d knobs=%s: %d samples differ from the closed form, %d of 3 from the literal oracle (%.0f s)"
              % (seed, shape, nodes, nq, packed, knobs, diff, lit, time.time() - t0), flush=True)
        bad += (diff > 0) + (lit > 0)
    print("fuzz_big: %d cases, %d with mismatches" % (n, bad))
    sys.exit(1 if bad else 0)


main()
