#!/usr/bin/env python3
"""Secondary benchmark: Fitch-Sankoff MAT construction (ugp_fitch_sankoff) on a synthetic tree.

Leaf genotypes are derived on the GPU (torch, plumbing only) from a seeded synthetic MAT, turned
into the VCF-cell CSR the C ABI takes, and assigned back onto the bare topology.  Checks: the
parsimony of the result is <= that of the generating MAT at the same sites, and a few sites are
compared with the oracle's literal Sankoff (tests only import oracle/; this is a bench check).
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--sites", type=int, default=2048)
    ap.add_argument("--check-sites", type=int, default=2)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--shape", default=None, help="sars2: the SARS-CoV-2-shaped synthetic tree (large polytomies) instead of random attachment")
    a = ap.parse_args()
    import torch
    from usher_amd import synth
    from usher_amd.fitch import fitch_sankoff
    dev = torch.device("cuda:0")
    st = synth.SynthTree(a.nodes, n_sites=25000 if a.nodes > 200000 else 1500, seed=1, **({"shape": a.shape} if a.shape else {}))
    A = st.arrays
    n = A["n"]
    parent = A["parent"]
    pos_all = np.unique(A["mut_pos"])
    pos_sel = pos_all[:a.sites]
    S = len(pos_sel)
    # reference allele per selected position
    first = np.searchsorted(np.sort(A["mut_pos"]), pos_sel)
    order = np.argsort(A["mut_pos"], kind="stable")
    ref = A["mut_ref"][order[first]].astype(np.uint8)
    # level boundaries (BFS order: parent[] non-decreasing)
    bounds = [0, 1]
    while bounds[-1] < n:
        bounds.append(int(np.searchsorted(parent[1:], bounds[-1], side="left")) + 1)
    mut_node = np.repeat(np.arange(n), np.diff(A["mut_off"]))
    site_of = np.searchsorted(pos_sel, A["mut_pos"])
    keep = (site_of < S) & (pos_sel[np.minimum(site_of, S - 1)] == A["mut_pos"])
    m_node = torch.from_numpy(mut_node[keep]).to(dev)
    m_site = torch.from_numpy(site_of[keep]).to(dev)
    m_nuc = torch.from_numpy(A["mut_nuc"][keep].astype(np.uint8)).to(dev)
    gen_parsimony = int(keep.sum())
    t_par = torch.from_numpy(parent).to(dev)
    state = torch.empty((n, S), dtype=torch.uint8, device=dev)
    state[0] = torch.from_numpy(ref).to(dev)
    m_order = torch.argsort(m_node)
    m_node, m_site, m_nuc = m_node[m_order], m_site[m_order], m_nuc[m_order]
    for L in range(len(bounds) - 1):
        lo, hi = bounds[L], bounds[L + 1]
        if L:
            state[lo:hi] = state[t_par[lo:hi]]
        i0 = int(torch.searchsorted(m_node, lo))
        i1 = int(torch.searchsorted(m_node, hi))
        if i1 > i0:
            state[m_node[i0:i1], m_site[i0:i1]] = m_nuc[i0:i1]
    nch = np.bincount(parent[1:], minlength=n)
    leaves = torch.from_numpy(np.flatnonzero(nch == 0)).to(dev)
    t_ref = torch.from_numpy(ref).to(dev)
    cells_site, cells_node, cells_nuc = [], [], []
    for s0 in range(0, S, 128):
        sub = state[:, s0:s0 + 128].index_select(0, leaves)
        nz = (sub != t_ref[s0:s0 + 128]).nonzero()
        cells_site.append((nz[:, 1] + s0).cpu().numpy())
        cells_node.append(leaves[nz[:, 0]].cpu().numpy())
        cells_nuc.append(sub[nz[:, 0], nz[:, 1]].cpu().numpy())
        del sub, nz
    del state
    torch.cuda.empty_cache()
    cs, cn, cc = np.concatenate(cells_site), np.concatenate(cells_node), np.concatenate(cells_nuc)
    o = np.argsort(cs, kind="stable")
    cs, cn, cc = cs[o], cn[o].astype(np.uint32), cc[o].astype(np.uint8)
    off = np.searchsorted(cs, np.arange(S + 1)).astype(np.uint64)
    best = None
    par32 = np.where(np.asarray(parent).astype(np.int64) < 0, 0xFFFFFFFF, np.asarray(parent).astype(np.int64)).astype(np.uint32)   # (the C ABI's parent array)
    calls = []
    for _ in range(a.reps):
        t0 = time.time()
        site, node, mpar, mnuc = fitch_sankoff(par32, ref, off, cn, cc)
        dt = time.time() - t0
        calls.append(round(fitch_sankoff.last_call_s * 1e3, 2))
        best = fitch_sankoff.last_call_s if best is None else min(best, fitch_sankoff.last_call_s)
    ok = len(site) <= gen_parsimony
    checked = 0
    if a.check_sites:
        from oracle import capi
        for s in np.linspace(0, S - 1, a.check_sites).astype(int):
            lo, hi = int(off[s]), int(off[s + 1])
            _, opar, onuc = capi.fitch_site(parent, int(ref[s]), cn[lo:hi].astype(np.int64), cc[lo:hi].astype(np.int8))
            want = [(int(j), int(opar[j]), int(onuc[j])) for j in np.flatnonzero(onuc)]
            sel = site == s
            got = list(zip(node[sel].tolist(), mpar[sel].tolist(), mnuc[sel].tolist()))
            ok = ok and got == want
            checked += 1
    W = (S + 7) // 8
    print(json.dumps({"metric": "Fitch-Sankoff site assignments/sec (MAT construction)", "value": round(S / best, 2), "unit": "sites/s",
                      "nodes": int(n), "shape": a.shape or "random attachment", "sites": int(S), "levels": len(bounds) - 1, "cells": int(len(cs)), "seconds": round(best, 4),
                      "node_site_per_s": round(n * S / best, 1), "mutations_out": int(len(site)), "generating_mutations": gen_parsimony,
                      "algo_bytes": int(3 * n * W * 4), "oracle_sites_checked": checked, "parity_ok": bool(ok),
                      "seconds_is": "best ugp_fitch_sankoff call (C ABI, host arrays in, result handle out)", "calls_ms": calls}))
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
