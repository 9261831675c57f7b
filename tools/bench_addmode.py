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


def varint(v):
    v &= 0xFFFFFFFFFFFFFFFF
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def field_bytes(fno, payload):
    return varint((fno << 3) | 2) + varint(len(payload)) + payload


def write_pb(arrays, path):
    """parsimony.proto data{newick=1, node_mutations=2 (one list per node, depth-first order)}."""
    n = arrays["n"]
    parent = arrays["parent"]
    kids = [[] for _ in range(n)]
    for j in range(1, n):
        kids[parent[j]].append(j)
    # iterative preorder newick; leaves L<j>, internal nodes labelled in preorder like the loader numbers them
    parts, order = [], []
    stack = [(0, 0)]
    internal = 0
    label = {}
    while stack:
        j, k = stack.pop()
        if k == 0:
            order.append(j)
            if kids[j]:
                internal += 1
                label[j] = "node_%d" % internal
                parts.append("(")
            else:
                parts.append("L%d" % j)
        if k < len(kids[j]):
            if k:
                parts.append(",")
            stack.append((j, k + 1))
            stack.append((kids[j][k], 0))
        elif kids[j]:
            parts.append(")" + label[j])
    newick = "".join(parts) + ";"
    out = bytearray(field_bytes(1, newick.encode()))
    off, pos, ref, par, nuc = arrays["mut_off"], arrays["mut_pos"], arrays["mut_ref"], arrays["mut_par"], arrays["mut_nuc"]
    for j in order:
        ml = bytearray()
        for i in range(off[j], off[j + 1]):
            m = (varint(1 << 3) + varint(int(pos[i])) + varint(2 << 3) + varint(IDX[int(ref[i])]) + varint(3 << 3) + varint(IDX[int(par[i])]) +
                 field_bytes(4, varint(IDX[int(nuc[i])])))
            ml += field_bytes(1, m)
        out += field_bytes(2, bytes(ml))
    with open(path, "wb") as f:
        f.write(out)


def write_vcf(q, n, path):
    off, pos, ref, nuc, mis = q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"]
    rows = {}
    for s in range(n):
        for i in range(int(off[s]), int(off[s + 1])):
            rows.setdefault(int(pos[i]), (int(ref[i]), {}))[1][s] = 15 if mis[i] else int(nuc[i])
    with open(path, "w") as f:
        f.write("##fileformat=VCFv4.2\n#CHROM\tPOS\tID\tREF\tALT\tQUAL\tFILTER\tINFO\tFORMAT\t" + "\t".join("NEW%d" % s for s in range(n)) + "\n")
        for p in sorted(rows):
            r, cells = rows[p]
            alts = sorted({a for a in cells.values() if a != 15 and a != r})
            if not alts:
                alts = [next(a for a in (1, 2, 4, 8) if a != r)]
            line = ["0"] * n
            for s, a in cells.items():
                line[s] = "." if a == 15 else ("0" if a == r else str(alts.index(a) + 1))
            f.write("chr\t%d\t.\t%s\t%s\t.\t.\t.\tGT\t%s\n" % (p, NUC[r], ",".join(NUC[a] for a in alts), "\t".join(line)))


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
    write_pb(st.arrays, d + "/base.pb")
    q = st.queries(a.samples, seed=5, max_subst=3, n_lo=0, n_hi=3, iupac_hi=0)
    write_vcf(q, a.samples, d + "/new.vcf")
    write_vcf(q, a.research_samples, d + "/few.vcf")
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
    for line in err.splitlines():
        if line.startswith("[usher-amd profile]") or line.startswith("[ugp flatten]"):
            sys.stderr.write(line + "\n")
    ties = sum(1 for l in open(os.path.join(o_all, "placement_stats.tsv")) if l.split("\t")[2] != "1")
    print(json.dumps({"metric": "sequential sample insertions/sec (default usher mode, bin/usher-amd)", "nodes": int(st.arrays["n"]),
                      "samples": a.samples, "value": round(a.samples / max(t_all - t_load, 1e-9), 2), "unit": "samples/s",
                      "wall_s": round(t_all, 2), "fixed_cost_s": round(t_load, 2), "samples_with_ties": ties,
                      "research_mode": {"samples": a.research_samples, "wall_s": round(t_few_r, 2),
                                        "samples_per_s": round(a.research_samples / max(t_few_r - t_load, 1e-9), 2),
                                        "batched_wall_s_same_samples": round(t_few_b, 2), "identical_outputs": bool(same)},
                      "prep_s": round(prep_s, 1)}))
    if not same:
        sys.exit(1)


if __name__ == "__main__":
    main()
