#!/usr/bin/env python3
"""Extended differential fuzz on the GPU box: random trees (random attachment, polytomies, caterpillars) x random
scheduling knobs, every sample compared with the oracle.  `python tools/fuzz_gpu.py 200` runs seeds 0..199."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402
from tests import synth  # noqa: E402
from usher_amd import Placer, QueryBatch  # noqa: E402

KNOBS = ("UGP_COARSE_MIN_NODES", "UGP_UNIT_CHUNKS", "UGP_HEAVY_CHUNKS", "UGP_UB_EVERY", "UGP_PRUNE_MIN_WORDS", "UGP_NO_LPT", "UGP_NO_SEED",
         "UGP_LDS_SLOTS", "UGP_NO_SIB", "UGP_NO_SORT", "UGP_NO_DESCENT", "UGP_NO_BOUND2", "UGP_SPLIT_CYCLES", "UGP_UNIT_GROW", "UGP_UNIT_MAX", "UGP_LDS_BITS", "UGP_BOUND3", "UGP_NO_UNIQ", "UGP_FORK", "UGP_PRE_WEIGHT", "UGP_COARSE_PHASE2", "UGP_NMASK", "UGP_PHASE2_PACKED", "UGP_RADIX_SORT", "UGP_SPLIT_MANY")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    bad = 0
    for seed in range(n):
        rng = np.random.default_rng(50_000 + seed)
        kind = rng.choice(["random", "random", "polytomy", "caterpillar"])
        if kind == "random":
            arrays, queries = synth.make_case(9000 + seed, n_leaves=int(rng.integers(20, 3000)), n_queries=int(rng.integers(1, 1400)),
                                              n_sites=int(rng.integers(20, 500)), p_masked=float(rng.choice([0.0, 0.03])),
                                              root_muts=int(rng.integers(0, 3)), mut_counts=(0, 0, 1, 1, 1, 2, 3, int(rng.choice([3, 20]))),
                                              n_ambig=(0, 0, 2, 5, 30))
        elif kind == "polytomy":
            arrays, queries = synth.polytomy_case(9000 + seed, fanouts=tuple(int(x) for x in rng.integers(2, 60, size=int(rng.integers(2, 4)))),
                                                  n_queries=int(rng.integers(1, 1200)))
        else:
            arrays, queries = synth.caterpillar_case(9000 + seed, depth=int(rng.integers(5, 500)), muts_per_node=int(rng.integers(1, 4)),
                                                     n_queries=int(rng.integers(1, 700)), n_sites=int(rng.integers(50, 2000)))
        for k in KNOBS:
            os.environ.pop(k, None)
        knobs = {"UGP_COARSE_MIN_NODES": "0", "UGP_UNIT_CHUNKS": str(int(rng.integers(1, 12))), "UGP_HEAVY_CHUNKS": str(int(rng.integers(1, 12))),
                 "UGP_UB_EVERY": str(int(rng.choice([1, 2, 7, 1000]))), "UGP_PRUNE_MIN_WORDS": str(int(rng.choice([1, 2, 4, 8, 40]))),
                 "UGP_LDS_SLOTS": str(int(rng.integers(1, 12)))}
        if rng.random() < 0.4:
            knobs["UGP_UNIT_CHUNKS"] = "16"
        if rng.random() < 0.5:
            knobs["UGP_SPLIT_CYCLES"] = str(int(rng.choice([0, 1, 2000, 400000])))
            knobs["UGP_UNIT_GROW"] = str(int(rng.choice([0, 1, 4])))
            knobs["UGP_UNIT_MAX"] = str(int(rng.choice([3, 64, 100000])))
        if rng.random() < 0.5:
            knobs["UGP_LDS_BITS"] = str(int(rng.integers(0, 3)))   # (2: the walk without the active-row bitmap, round 5)
        if rng.random() < 0.7:
            knobs["UGP_BOUND3"] = str(int(rng.integers(0, 2)))     # (third pruning bound pinned on / off; otherwise the library's static rule: off on trees this small)
        for k, p in (("UGP_NO_LPT", 0.3), ("UGP_NO_SEED", 0.2), ("UGP_NO_SIB", 0.2), ("UGP_NO_SORT", 0.1), ("UGP_NO_DESCENT", 0.3), ("UGP_NO_BOUND2", 0.2),
                     ("UGP_PRE_WEIGHT", 0.2), ("UGP_COARSE_PHASE2", 0.2), ("UGP_NMASK", 0.3), ("UGP_PHASE2_PACKED", 0.25), ("UGP_RADIX_SORT", 0.3), ("UGP_NO_UNIQ", 0.25), ("UGP_FORK", 0.3)):
            if rng.random() < p:
                knobs[k] = "1"
        os.environ.update(knobs)
        ot = capi.OracleTree(arrays)
        pl = Placer(arrays, chunk_nodes=int(rng.integers(2, 90)), experiments="UGP_PHASE2_PACKED" in knobs)
        res = pl.place(QueryBatch(queries))
        for i, s in enumerate(queries):
            w = ot.place(s, want_ties=False)
            got = (int(res["best_set_difference"][i]), int(res["num_best"][i]), int(res["best_j"][i]), bool(res["best_has_unique"][i]))
            if got != (w["best"], w["num_best"], w["best_j"], w["has_unique"]):
                bad += 1
                print("MISMATCH seed", seed, kind, knobs, "sample", i, got, w)
                break
        pl.close()
    print("fuzz: %d cases, %d with mismatches" % (n, bad))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
