#!/usr/bin/env python3
"""Random cases of ugp_fitch_sankoff against the oracle's literal Sankoff (tests/test_fitch.py's generators): tree kinds and sizes, site
counts around the 512-site tile boundaries, cell densities from none to every leaf, cells on internal nodes, several passes
(UGP_FITCH_BYTES) and the exact listing pass (UGP_FITCH_EMIT_CAP).     python tools/fuzz_fitch.py --cases 200 --seed 1
Prints one line per mismatch and a summary; exit code 1 on any mismatch.  (A tool: it imports oracle/ the way the tests do.)"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=200)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-nodes", type=int, default=4000)
    a = ap.parse_args()
    import test_fitch as T
    from usher_amd.fitch import fitch_sankoff
    rng = np.random.default_rng(a.seed)
    bad = 0
    t0 = time.time()
    shapes = {}
    for case in range(a.cases):
        kind = ["random", "bushy", "chain", "star"][int(rng.integers(0, 4))]
        n = int(rng.integers(1, a.max_nodes if kind != "chain" else 200))
        n_sites = int(rng.choice([1, 7, 8, 9, 63, 64, 65, 500, 511, 512, 513, 520, 1023, 1025, 1500, 2100]))
        if n * n_sites > 3_000_000:
            n_sites = max(1, 3_000_000 // n)
        p_var = float(rng.choice([0.0, 0.0005, 0.003, 0.02, 0.3]))
        p_int = float(rng.choice([0.0, 0.0, 0.002, 0.05]))
        parent = T.random_bfs_tree(rng, n, kind)
        ref, off, nodes, nucs = T.random_sites(rng, parent, n_sites, p_var=p_var, p_internal=p_int, p_dense=float(rng.choice([2 * p_var, 0.9])))
        want = T.oracle_mutations(parent, ref, off, nodes, nucs)
        mode = int(rng.integers(0, 4))
        os.environ.pop("UGP_FITCH_BYTES", None)
        os.environ.pop("UGP_FITCH_EMIT_CAP", None)
        os.environ.pop("UGP_FITCH_PIECES", None)
        if rng.random() < 0.5:
            os.environ["UGP_FITCH_PIECES"] = str(int(rng.integers(2, 5)))   # (the upload in pieces of whole tiles, as for large inputs)
        if mode == 1:
            os.environ["UGP_FITCH_BYTES"] = str((n + 1) * 4 * int(rng.choice([1, 3, 64, 65, 130])))
        elif mode == 2:
            os.environ["UGP_FITCH_EMIT_CAP"] = str(int(rng.choice([1, 3, 17])))
        site, node, mpar, mnuc = fitch_sankoff(parent, ref, off, nodes, nucs)
        got = list(zip(site.tolist(), node.tolist(), mpar.tolist(), mnuc.tolist()))
        nch = np.bincount(parent[1:], minlength=n) if n > 1 else np.zeros(1, np.int64)
        key = "polytomy>255" if nch.max() > 255 else "<=255" if nch.max() > 31 else "<=31" if nch.max() > 7 else "<=7"
        shapes[key] = shapes.get(key, 0) + 1
        if got != want:
            bad += 1
            print("MISMATCH case %d: %s n=%d sites=%d p_var=%g p_int=%g mode=%d  got %d want %d" % (case, kind, n, n_sites, p_var, p_int, mode, len(got), len(want)))
    print("fuzz_fitch: %d cases, %d mismatches, %.0f s; widest node of the case: %s" % (a.cases, bad, time.time() - t0, shapes))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
