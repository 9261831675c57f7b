#!/usr/bin/env python3
"""Round 5: what granularity would a subtree-SITE bound need?  (analysis only; the first half is bound_sim.py)

On top of the two bounds the kernel has (C below), the gain term A(n, s) -- the sample's unmatched true variants, all of which bound (2)
assumes can still be matched below n -- is capped by hA(n) = the largest number of mutations on a path n -> descendant that CAN match a
variant of the samples in question (mutated allele in the sample's set, reference base not in it):
  E1  hA over the whole 512-sample tile        (what a per-tile table of "useful" sites could give)
  E2  hA over the sample's 64-sample group     (8 groups per tile)
  E3  hA for the sample itself                 (a per-sample signature)
  cost(d, s) >= D(n, s) - min(hsub(n), min(A(n, s), hA) + hsec(n))
Exact seeds (slack 0).

--- bound_sim.py's own header follows ---
CPU simulation of the pruning walk of k_best8 for ONE 512-sample tile under two lower bounds (analysis only).

  A: cost(d, s) >= D(n, s) - hsub(n)                          (every mutation lowers D by at most 1)
  B: cost(d, s) >= D(n, s) - min(hsub(n), V_s + hrev(n))      V_s = non-missing rows of the sample, hrev(n) = largest
     number of mutations back to the reference base on a path n -> descendant: a site can lower D along a path only
     if the sample has a row there or the path reverts it to the reference.

A node is walked unless a proper ancestor carrying a pruning record (descendants >= 4 words) had
D - bound > best(s) for all 512 samples (bounds seeded with the exact answers: the best case for both).
usage: bound_sim.py [nodes] [tile index]
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from usher_amd import synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
TILE = int(sys.argv[2]) if len(sys.argv) > 2 else 7
SHAPE = sys.argv[3] if len(sys.argv) > 3 else "random"
Q = 16384
n_sites = 25000 if N >= 1_000_000 else 1500
st = synth.SynthTree(N, n_sites=n_sites, seed=1, shape=SHAPE)
A = st.arrays
n = int(A["n"])
par = np.asarray(A["parent"]).astype(np.int64)
off = np.asarray(A["mut_off"]).astype(np.int64)
mpos = np.asarray(A["mut_pos"]).astype(np.int64)
mref = np.asarray(A["mut_ref"]).astype(np.uint8)
mnuc = np.asarray(A["mut_nuc"]).astype(np.uint8)
M = len(mpos)
nw = np.diff(off)
node_of_mut = np.repeat(np.arange(n), nw)
# levels (BFS order: parent[] ascending)
level = np.zeros(n, np.int32)
starts = [0, 1]
while starts[-1] < n:
    lo, hi = starts[-2], starts[-1]
    kids = np.searchsorted(par[1:], [lo, hi]) + 1          # children of [lo, hi) are nodes [kids[0], kids[1])
    starts.append(int(kids[1]))
    if starts[-1] == starts[-2]:
        break
starts = [s for i, s in enumerate(starts) if i == 0 or s > starts[i - 1]]
if starts[-1] != n:
    starts.append(n)
print("nodes", n, "mutations", M, "levels", len(starts) - 1)

# true parent state of every mutation: walk levels keeping per-node state only for mutated sites is heavy; use the
# generator's own par field (the synthetic tree stores the true parent allele)
mpar = np.asarray(A["mut_par"]).astype(np.uint8)
is_rev = (mnuc == mref).astype(np.int64)
rev_n = np.bincount(node_of_mut, weights=is_rev, minlength=n).astype(np.int64)
sec_n = np.bincount(node_of_mut, weights=(mpar != mref).astype(np.int64), minlength=n).astype(np.int64)   # mutations of a site that is not at the reference base

# bottom-up: hsub, hrev, subtree words
hsub = np.zeros(n, np.int64); hrev = np.zeros(n, np.int64); subw = np.zeros(n, np.int64); hsec = np.zeros(n, np.int64)
for li in range(len(starts) - 2, 0, -1):
    lo, hi = starts[li], starts[li + 1]
    p = par[lo:hi]
    np.maximum.at(hsub, p, nw[lo:hi] + hsub[lo:hi])
    np.maximum.at(hrev, p, rev_n[lo:hi] + hrev[lo:hi])
    np.maximum.at(hsec, p, sec_n[lo:hi] + hsec[lo:hi])
    np.add.at(subw, p, subw[lo:hi] + 1 + nw[lo:hi])
has_rec = (subw >= 4) & (hsub <= 127)
print("hsec max", hsec.max(), "nodes with hsec > 0:", int((hsec > 0).sum()))
print("hsub max", hsub.max(), "hrev max", hrev.max(), "nodes with hrev > 0:", int((hrev > 0).sum()), "records", int(has_rec.sum()))

# queries, sorted by the DFS position of their source node ~ the locality sort
q = st.queries(Q, seed=77)
src = np.asarray(q["source"]).astype(np.int64)
sub = np.ones(n, np.int64)
for li in range(len(starts) - 2, 0, -1):
    lo, hi = starts[li], starts[li + 1]
    np.add.at(sub, par[lo:hi], sub[lo:hi])
# dfs index (children in BFS order; the product reorders them, irrelevant here)
dfs = np.zeros(n, np.int64)
first_child = np.searchsorted(par[1:], np.arange(n)) + 1
for li in range(0, len(starts) - 2):
    lo, hi = starts[li], starts[li + 1]
    klo, khi = starts[li + 1], starts[li + 2] if li + 2 < len(starts) else n
    # exclusive cumsum of sub over the children of each parent
    c = np.arange(klo, khi)
    cs = np.cumsum(sub[c]) - sub[c]
    base = cs[first_child[par[c]] - klo]
    dfs[c] = dfs[par[c]] + 1 + (cs - base)
order = np.argsort(dfs[src], kind="stable")
tile_q = order[TILE * 512:(TILE + 1) * 512]
eo = np.asarray(q["ent_off"]).astype(np.int64)
qpos = np.asarray(q["pos"]).astype(np.int64); qnuc = np.asarray(q["nuc"]).astype(np.uint8); qmiss = np.asarray(q["is_missing"]).astype(np.uint8)
qref = np.asarray(q["ref"]).astype(np.uint8)
L = int(max(mpos.max(), qpos.max())) + 1
ref_at = np.zeros(L, np.uint8)
ref_at[mpos] = mref
ref_at[qpos] = np.where(ref_at[qpos] == 0, qref, ref_at[qpos])
S = np.tile(ref_at, (512, 1))                               # allele set of every sample at every position
V = np.zeros(512, np.int64)
dbot = np.zeros(512, np.int64)
for k, s in enumerate(tile_q):
    a, b = eo[s], eo[s + 1]
    mask = np.where(qmiss[a:b] != 0, 15, qnuc[a:b]).astype(np.uint8)
    S[k, qpos[a:b]] = mask
    V[k] = int((qmiss[a:b] == 0).sum())
    dbot[k] = int(((mask & qref[a:b]) == 0).sum())
t0 = time.time()
# delta of every mutation for every sample, summed per node
Sm = S[:, mpos]                                             # [512, M]
delta = ((Sm & mpar[None, :]) != 0).astype(np.int8) - ((Sm & mnuc[None, :]) != 0).astype(np.int8)
neg = np.minimum(delta, 0)
common = ((Sm & mnuc[None, :]) != 0)
idx = off[:-1][nw > 0]
dn = np.zeros((512, n), np.int16); cn = np.zeros((512, n), np.int16); en = np.zeros((512, n), bool)
dn[:, nw > 0] = np.add.reduceat(delta, idx, axis=1)
# A(n, s) = mismatching sites whose sample set excludes the reference base (the sample's true variants not matched at n)
refin = ((Sm & mref[None, :]) != 0)
an = np.zeros((512, n), np.int16)
an[:, nw > 0] = np.add.reduceat(np.where(refin, 0, delta).astype(np.int8), idx, axis=1)
del refin
cn[:, nw > 0] = np.add.reduceat(neg, idx, axis=1)
en[:, nw > 0] = np.add.reduceat(common, idx, axis=1) > 0
useful = (((Sm & mnuc[None, :]) != 0) & ((Sm & mref[None, :]) == 0))   # [512, M]: this mutation can match a true variant of the sample
del delta, neg, common, Sm
D = np.zeros((512, n), np.int16)
D[:, 0] = dbot + dn[:, 0]
Aout = np.zeros((512, n), np.int16)
Aout[:, 0] = dbot + an[:, 0]                                  # below the root every mismatch is an unmatched true variant
for li in range(1, len(starts) - 1):
    lo, hi = starts[li], starts[li + 1]
    D[:, lo:hi] = D[:, par[lo:hi]] + dn[:, lo:hi]
    Aout[:, lo:hi] = Aout[:, par[lo:hi]] + an[:, lo:hi]
del an
leaf = np.ones(n, bool); leaf[par[1:]] = False
cost = np.empty((512, n), np.int32)
cost[:, 0] = D[:, 0]
cost[:, 1:] = D[:, par[1:]].astype(np.int32) + cn[:, 1:]
elig = en | (~leaf & (nw == 0))[None, :]
elig[:, 0] = True
best = np.where(elig, cost, 1 << 20).min(axis=1)
print("tile", TILE, "best scores: min %d median %d max %d; V: median %d max %d; D tables %.1f s" % (best.min(), np.median(best), best.max(), np.median(V), V.max(), time.time() - t0))


def walk(bound, name):
    far = ((D.astype(np.int64) - bound) > best[:, None]).all(axis=0)
    pruned_at = far & has_rec
    blocked = np.zeros(n, bool)
    for li in range(1, len(starts) - 1):
        lo, hi = starts[li], starts[li + 1]
        blocked[lo:hi] = blocked[par[lo:hi]] | pruned_at[par[lo:hi]]
    visited = ~blocked
    print("%-58s visited nodes %9d (%.2f %%)  jumps %8d" % (name, int(visited.sum()), 100.0 * visited.sum() / n, int((pruned_at & visited).sum())), flush=True)

hit = (np.where(elig, cost, 1 << 20) <= best[:, None]).any(axis=0)
need = np.zeros(n, bool)
for li in range(len(starts) - 2, 0, -1):
    lo, hi = starts[li], starts[li + 1]
    np.logical_or.at(need, par[lo:hi], need[lo:hi] | hit[lo:hi])
vis = np.ones(n, bool); vis[1:] = need[par[1:]]
print("floor (perfect subtree test): visited nodes %d (%.2f %%)" % (int(vis.sum()), 100.0 * vis.sum() / n))

A64 = Aout.astype(np.int64)
base = np.minimum(hsub[None, :], A64 + hsec[None, :])
walk(base, "C (the kernel): D - min(hsub, A + hsec)")

def path_max(cnt_node):   # largest sum of cnt over the nodes of a path n -> descendant (n excluded), bottom-up
    h = np.zeros(cnt_node.shape, np.int32)
    for li in range(len(starts) - 2, 0, -1):
        lo, hi = starts[li], starts[li + 1]
        if h.ndim == 1:
            np.maximum.at(h, par[lo:hi], cnt_node[lo:hi] + h[lo:hi])
        else:
            np.maximum.at(h, par[lo:hi], cnt_node[lo:hi] + h[lo:hi])   # rows = nodes
    return h

def per_node(flags_m):   # [M] bool -> count per node
    return np.bincount(node_of_mut, weights=flags_m.astype(np.float64), minlength=n).astype(np.int32)

t0 = time.time()
u_tile = useful.any(axis=0)
hA1 = path_max(per_node(u_tile)).astype(np.int64)
print("E1: useful for SOME sample of the tile: %.1f %% of the mutations; hA_tile median over recorded nodes %d vs hsub %d" % (100.0 * u_tile.mean(), np.median(hA1[has_rec]), np.median(hsub[has_rec])))
walk(np.minimum(hsub[None, :], np.minimum(A64, hA1[None, :]) + hsec[None, :]), "E1 tile-level useful sites")
# E1s: what the kernel could do WITHOUT a per-tile pass over the tree: for a record whose descendants are at most K stream words away
# -- words the walk's pipeline already holds, with their "useful for this tile" bits -- the number of useful words among them (the
# subtree's TOTAL, an upper bound of the path maximum); hsub for larger subtrees
u_node = per_node(u_tile).astype(np.int64)
u_sub = np.zeros(n, np.int64)
for li in range(len(starts) - 2, 0, -1):
    lo, hi = starts[li], starts[li + 1]
    np.add.at(u_sub, par[lo:hi], u_sub[lo:hi] + u_node[lo:hi])
for K in ((32, 1 << 30) if os.environ.get('SIM_V1_ONLY') else (8, 16, 24, 32, 48, 64, 128, 1 << 30)):
    hk = np.where(subw <= K, np.minimum(hsub, u_sub + hsec), hsub)
    walk(np.minimum(hk[None, :], A64 + hsec[None, :]), "E1s subtree totals within %s words" % (K if K < (1 << 30) else "any number of"))
if os.environ.get('SIM_SKIP_E23'): sys.exit(0)
# V1: an implementable path maximum: the useful events of the tile (4 % of the mutations, found through static posting
# lists per (site, allele)) raise a counter over the DFS interval of their node's subtree, at BLOCK granularity (a block = BLK nodes of
# the DFS order ~ 2 BLK stream words): cum_over[block] >= the number of useful events on the root path of any node of the block.  For a
# record at n:  hU(n) <= max(cum_over over the blocks of n's descendants) - cumU(n), cumU(n) = useful events on root -> n, exact (the
# walk tracks it like D).  Blocks are read 64 at a time, at the coarsest of 64-ary levels whose span of the range is < 64 entries.
cumU = np.zeros(n, np.int64)
cumU[0] = u_node[0]
for li in range(1, len(starts) - 1):
    lo, hi = starts[li], starts[li + 1]
    cumU[lo:hi] = cumU[par[lo:hi]] + u_node[lo:hi]
ev_node = node_of_mut[u_tile]
for BLK in ((8, 16) if os.environ.get('SIM_BLK_BIG') else (1, 2, 4)):
    nb = (n + BLK - 1) // BLK + 2
    b0 = dfs[ev_node] // BLK
    b1 = (dfs[ev_node] + sub[ev_node] - 1) // BLK
    plus = np.bincount(b0, minlength=nb).astype(np.int64)
    minus = np.bincount(b1 + 1, minlength=nb + 1).astype(np.int64)[:nb]
    cum_over = np.cumsum(plus) - np.cumsum(minus)
    levels = [cum_over]
    while len(levels[-1]) > 64:
        a = levels[-1]
        pad = (-len(a)) % 64
        a = np.concatenate([a, np.zeros(pad, np.int64)]).reshape(-1, 64).max(axis=1)
        levels.append(a)
    def sparse(a):
        st = [a]
        k = 1
        while k < 64:
            prev = st[-1]
            nxt = prev.copy()
            if len(prev) > k:
                nxt[:len(prev) - k] = np.maximum(prev[:len(prev) - k], prev[k:])
            st.append(nxt)
            k *= 2
        return st
    sts = [sparse(a) for a in levels]
    q0 = (dfs + 1) // BLK                     # first descendant
    q1 = (dfs + sub - 1) // BLK               # last descendant
    has_desc = sub > 1
    hU = np.zeros(n, np.int64)
    done = ~has_desc
    for L, st in enumerate(sts):
        i0 = q0 >> (6 * L)
        i1 = q1 >> (6 * L)
        sel = (~done) & (i1 - i0 < 64)
        if sel.any():
            ln = (i1 - i0 + 1)[sel]
            k = np.floor(np.log2(ln)).astype(np.int64)
            a0 = i0[sel]; a1 = i1[sel] - (1 << k) + 1
            m = np.zeros(sel.sum(), np.int64)
            for kk in range(7):
                w = k == kk
                if w.any():
                    m[w] = np.maximum(st[kk][a0[w]], st[kk][a1[w]])
            hU[sel] = np.maximum(0, m - cumU[sel])
            done |= sel
    hk = np.minimum(hsub, hU + hsec)
    ok = (hU >= hA1).all()
    walk(np.minimum(hk[None, :], A64 + hsec[None, :]), "V1 block cum maxima, block = %d nodes (valid: %s)" % (BLK, ok))
    # V2: nothing tracked by the walk -- cumU(n) replaced by cum_under[block of n] = events that cover n's whole block (they lie on n's
    # root path), so hU2 = (maximum over the descendants' blocks) - cum_under >= hU
    same = np.bincount(b0[b0 == b1], minlength=nb).astype(np.int64)
    endc = np.bincount(b1, minlength=nb).astype(np.int64)
    startcum = np.cumsum(plus)
    cum_under = np.concatenate([[0], startcum[:-1]]) - np.cumsum(endc) + same
    assert (cum_under >= 0).all() and (cum_under[dfs // BLK] <= cumU).all()
    # the maximum over the range as computed above was m - cumU: redo with cum_under
    hU2 = np.zeros(n, np.int64)
    done = ~has_desc
    for L, st in enumerate(sts):
        i0 = q0 >> (6 * L)
        i1 = q1 >> (6 * L)
        sel = (~done) & (i1 - i0 < 64)
        if sel.any():
            ln = (i1 - i0 + 1)[sel]
            k = np.floor(np.log2(ln)).astype(np.int64)
            a0 = i0[sel]; a1 = i1[sel] - (1 << k) + 1
            m = np.zeros(sel.sum(), np.int64)
            for kk in range(7):
                w = k == kk
                if w.any():
                    m[w] = np.maximum(st[kk][a0[w]], st[kk][a1[w]])
            hU2[sel] = np.maximum(0, m - cum_under[(dfs // BLK)[sel]])
            done |= sel
    hk2 = np.minimum(hsub, hU2 + hsec)
    walk(np.minimum(hk2[None, :], A64 + hsec[None, :]), "V2 no tracking: max - cum_under, block = %d nodes (valid: %s)" % (BLK, (hU2 >= hA1).all()))
if os.environ.get('SIM_V1_ONLY'): sys.exit(0)
bound2 = np.empty((512, n), np.int64)
for g in range(8):
    ug = useful[g * 64:(g + 1) * 64].any(axis=0)
    hg = path_max(per_node(ug)).astype(np.int64)
    bound2[g * 64:(g + 1) * 64] = np.minimum(hsub[None, :], np.minimum(A64[g * 64:(g + 1) * 64], hg[None, :]) + hsec[None, :])
    if g == 0: print("E2: useful for some sample of a 64-sample group: %.1f %% of the mutations" % (100.0 * ug.mean()))
walk(bound2, "E2 64-sample groups")
# E3: per sample (nodes x samples table, int16)
cnt_s = np.zeros((n, 512), np.int16)
idx = off[:-1][nw > 0]
cnt_s[nw > 0] = np.add.reduceat(useful.T.astype(np.int16), idx, axis=0)
hs = np.zeros((n, 512), np.int16)
for li in range(len(starts) - 2, 0, -1):
    lo, hi = starts[li], starts[li + 1]
    np.maximum.at(hs, par[lo:hi], cnt_s[lo:hi] + hs[lo:hi])
print("E3: useful per sample: %.3f %% of the mutations (mean)" % (100.0 * useful.mean()))
walk(np.minimum(hsub[None, :], np.minimum(A64, hs.T.astype(np.int64)) + hsec[None, :]), "E3 per-sample useful sites")
print("tables %.0f s" % (time.time() - t0))
