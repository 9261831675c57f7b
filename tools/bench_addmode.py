#!/usr/bin/env python3
"""Secondary benchmark: default `usher` mode (samples inserted one after another) through bin/usher-amd on a
synthetic MAT written as parsimony.proto, with the batched re-derivation scheme and, on a
subset, with a full search per sample (USHER_AMD_MAX_TOUCHED=0 = the reference's loop).  Prints one JSON line."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
NUC = {1: "A", 2: "C", 4: "G", 8: "T"}
IDX = {1: 0, 2: 1, 4: 2, 8: 3}


from tools.time_load import host_lib, write_workload   # the synthetic workload as parsimony.proto / VCF files (C++ writers)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=1_000_000)
    ap.add_argument("--samples", type=int, default=1000)
    ap.add_argument("--research-samples", type=int, default=40)
    a = ap.parse_args()
    from usher_amd import synth
    exe = os.path.join(ROOT, "usher_amd", "bin", "usher-amd")
    d = tempfile.mkdtemp(prefix="addmode_")
    st = synth.SynthTree(a.nodes, n_sites=25000 if a.nodes > 200000 else 1500, seed=1)
    t0 = time.time()
    L = host_lib()
    q = st.queries(a.samples, seed=5, max_subst=3, n_lo=0, n_hi=3, iupac_hi=0)
    write_workload(L, st, q, a.samples, d + "/base.pb", d + "/new.vcf")
    write_workload(L, st, q, a.research_samples, None, d + "/few.vcf")
    prep_s = time.time() - t0

    def run(vcf, env, extra):
        e = dict(os.environ)
        e.pop("USHER_AMD_MAX_TOUCHED", None)
        e.update(env)
        out = tempfile.mkdtemp(prefix="out_", dir=d)
        t = time.time()
        r = subprocess.run([exe, "-i", d + "/base.pb", "-v", vcf, "-d", out] + extra, capture_output=True, text=True, env=e)
        dt = time.time() - t
        if r.returncode != 0:
            sys.stderr.write(r.stderr[-3000:])
            sys.exit(1)
        return dt, out, r.stderr

    t_load, _, _ = run(d + "/few.vcf", {}, ["-n"])                         # load + flatten + one static batch: the fixed cost
    t_few_b, o_fb, _ = run(d + "/few.vcf", {}, [])
    t_few_r, o_fr, _ = run(d + "/few.vcf", {"USHER_AMD_MAX_TOUCHED": "0"}, [])
    same = all(open(os.path.join(o_fb, n)).read() == open(os.path.join(o_fr, n)).read() for n in ("placement_stats.tsv", "final-tree.nh"))
    t_all, o_all, err = run(d + "/new.vcf", {"USHER_AMD_PROFILE": "1", "UGP_FLATTEN_VERBOSE": os.environ.get("UGP_FLATTEN_VERBOSE", "")}, [])
    prof = [line for line in err.splitlines() if line.startswith("[usher-amd profile]") or line.startswith("[ugp flatten]")]
    for line in prof:
        sys.stderr.write(line + "\n")
    ties = sum(1 for l in open(os.path.join(o_all, "placement_stats.tsv")) if l.split("\t")[2] != "1")
    print(json.dumps({"metric": "sequential sample insertions/sec (default usher mode, bin/usher-amd)", "nodes": int(st.arrays["n"]),
                      "samples": a.samples, "value": round(a.samples / max(t_all - t_load, 1e-9), 2), "unit": "samples/s",
                      "wall_s": round(t_all, 2), "fixed_cost_s": round(t_load, 2), "samples_with_ties": ties,
                      "research_mode": {"samples": a.research_samples, "wall_s": round(t_few_r, 2),
                                        "samples_per_s": round(a.research_samples / max(t_few_r - t_load, 1e-9), 2),
                                        "batched_wall_s_same_samples": round(t_few_b, 2), "identical_outputs": bool(same)},
                      "prep_s": round(prep_s, 1), "profile": [l for l in prof if l.startswith("[usher-amd profile]")]}))
    if not same:
        sys.exit(1)


if __name__ == "__main__":
    main()
