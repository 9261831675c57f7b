#!/usr/bin/env python3
"""CPU simulation of chunk-level lower bounds (analysis only; uses the oracle's closed form as a D provider).

For chunk c with first node f, every node d of the chunk has a deepest ancestor a(d) among the proper ancestors
of f, and cost(d, s) >= D(a(d), s) - words(a(d) -> d].  With H_i(c) = max words over the nodes hanging off path
node p_i, the chunk holds no candidate for a tile when D(p_i, s) - H_i > ub(s) for every record i and every
sample s of the tile.  This script measures how many (tile, chunk) pairs and (tile, 8-chunk unit) pairs stay
alive under that test.   usage: chunk_bound_sim.py [nodes] [n_tiles_to_simulate]
"""
import ctypes as C
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import capi  # noqa: E402
from usher_amd import FlatTreeView, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
TILES = int(sys.argv[2]) if len(sys.argv) > 2 else 1
Q = 16384
st = synth.SynthTree(N, n_sites=25000 if N >= 1_000_000 else 1500, seed=1)
A = st.arrays
n = A["n"]
fv = FlatTreeView(A)
d2b = fv.dfs2bfs.astype(np.int64)
cno = fv.chunk_node_off.astype(np.int64)
n_chunks = len(cno) - 1
par = A["parent"]
nw = np.diff(A["mut_off"]).astype(np.int64)          # (no masked mutations in the synthetic tree)
# cumulative words root -> node (BFS order: parents first)
pw = np.zeros(n, np.int64)
lvl_par = par.copy()
order = np.arange(n)
pw[0] = nw[0]
# vectorised by BFS levels
level = np.zeros(n, np.int32)
for j0 in range(1, n, 1 << 20):
    j1 = min(n, j0 + (1 << 20))
    # parents of a block may lie inside the block: fall back to a loop only there
    blk = np.arange(j0, j1)
    inside = par[blk] >= j0
    if inside.any():
        for j in blk:
            pw[j] = pw[par[j]] + nw[j]
    else:
        pw[blk] = pw[par[blk]] + nw[blk]
b2d = np.empty(n, np.int64); b2d[d2b] = np.arange(n)
sub = np.ones(n, np.int64)
for j in range(n - 1, 0, -1):
    pass
# subtree sizes by DFS: sub via reverse accumulate (numpy add.at in BFS reverse order would be slow in python; use C-ish trick)
sub = np.ones(n, np.int64)
idx = np.arange(n - 1, 0, -1)
# process in blocks from the end; parents of a block always have smaller index
for j1 in range(n, 1, -(1 << 20)):
    j0 = max(1, j1 - (1 << 20))
    blk = np.arange(j1 - 1, j0 - 1, -1)
    inside = par[blk] >= j0
    if inside.any():
        for j in blk:
            sub[par[j]] += sub[j]
    else:
        np.add.at(sub, par[blk], sub[blk])
t0 = time.time()
rec_node, rec_H, rec_chunk = [], [], []
for c in range(n_chunks):
    f = d2b[cno[c]]
    # proper ancestors of f, deepest first: dfs index and subtree end
    anc = []
    a = par[f]
    while a >= 0:
        anc.append(a); a = par[a]
    anc = np.asarray(anc, np.int64)
    if len(anc) == 0:          # the chunk starts at the root
        rec_node.append(0); rec_H.append(10 ** 6); rec_chunk.append(c); continue
    ends = b2d[anc] + sub[anc]                       # non-decreasing from deep to shallow
    dn = np.arange(cno[c], cno[c + 1])               # dfs indices of the chunk's nodes
    k = np.searchsorted(ends, dn, side="right")      # first ancestor whose subtree still contains d
    nodes = d2b[dn]
    h = pw[nodes] - pw[anc[k]]
    hk = np.zeros(len(anc), np.int64) - 1
    np.maximum.at(hk, k, h)
    for i in np.flatnonzero(hk >= 0):
        rec_node.append(anc[i]); rec_H.append(hk[i]); rec_chunk.append(c)
rec_node = np.asarray(rec_node); rec_H = np.asarray(rec_H); rec_chunk = np.asarray(rec_chunk)
starts = np.flatnonzero(np.r_[True, rec_chunk[1:] != rec_chunk[:-1]])
print("chunks %d  records %d (%.2f per chunk)  H mean %.1f max %d  [%.1fs]" % (n_chunks, len(rec_node), len(rec_node) / n_chunks, rec_H[rec_H < 10**6].mean(), rec_H[rec_H < 10**6].max(), time.time() - t0))

q = st.queries(Q, seed=1017)
ot = capi.OracleTree(A)
cf = capi.ClosedFormC(ot)
res = cf.place_csr(q["ent_off"], q["pos"], q["ref"], q["nuc"], q["is_missing"])
key = b2d[res["best_j"]]
order = np.argsort(key, kind="stable")
L = capi.lib()
L.orc_cf_D.restype = C.c_int
L.orc_cf_D.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 5


def sample_alive(si):
    s = synth.csr_sample(q, int(si))
    D = np.empty(n, np.int32)
    pos = np.ascontiguousarray(s["pos"], np.int32); ref = np.ascontiguousarray(s["ref"], np.int8)
    nuc = np.ascontiguousarray(s["nuc"], np.int8); mis = np.ascontiguousarray(s["is_missing"], np.int8)
    L.orc_cf_D(cf.h, len(pos), pos.ctypes.data, ref.ctypes.data, nuc.ctypes.data, mis.ctypes.data, D.ctypes.data)
    val = D[rec_node].astype(np.int64) - rec_H
    return np.minimum.reduceat(val, starts) <= int(res["best"][si])


for t in range(TILES):
    tile = order[(t * (Q // 512 // max(TILES, 1)) if TILES > 1 else 7) * 512:][:512]
    t0 = time.time()
    with ThreadPoolExecutor(8) as ex:
        alive = np.zeros(n_chunks, bool)
        for a in ex.map(sample_alive, tile):
            alive |= a
    own_lo, own_hi = np.searchsorted(cno, key[tile].min(), "right") - 1, np.searchsorted(cno, key[tile].max(), "right")
    unit = alive[: n_chunks // 8 * 8].reshape(-1, 8).any(1)
    unit64 = alive[: n_chunks // 64 * 64].reshape(-1, 64).any(1)
    print("tile %d: own region chunks [%d, %d) = %.2f%%; alive chunks %.2f%%  alive 8-chunk units %.2f%%  alive 64-chunk units %.2f%%  [%.0fs]"
          % (t, own_lo, own_hi, 100.0 * (own_hi - own_lo) / n_chunks, 100.0 * alive.mean(), 100.0 * unit.mean(), 100.0 * unit64.mean(), time.time() - t0))
