"""Test helper: a python interpreter of the flattened DFS record stream.

It executes exactly the per-record algorithm of k_place (usher_amd/csrc/
ugp_kernels.hip) -- read slot / register / D_bottom, per-mutation delta from the
sample's allele nibble, write slot, score, (min, count, max key) reduction,
preamble replay per group -- so the host flattening (slots, preambles, chunk
cuts, true parent alleles, tie ranks) can be checked on CPU against the oracle
before any GPU run.  Test infrastructure only; never imported by usher_amd.
"""
from __future__ import annotations

import numpy as np

RS_REG, RS_BOTTOM, WS_NONE = 63, 62, 63
F_LEAF, F_NOSCORE, F_ROOT, F_MASKED = 1 << 28, 1 << 29, 1 << 30, 1 << 31
M_AFTER_MASK = 1 << 31
IDX2HOT = [1, 2, 4, 8]


def sample_site_alleles(flat, sample):
    """nibble per site (reference base unless the sample has a row) and D_bottom."""
    nib = flat.site_ref.astype(np.int64).copy()
    dbot = 0
    for p, r, a, mis in zip(sample["pos"], sample["ref"], sample["nuc"], sample["is_missing"]):
        a = 15 if mis else int(a)
        if not mis and (a & int(r)) == 0:
            dbot += 1
        if 0 <= p < len(flat.pos2site) and flat.pos2site[p] >= 0:
            nib[flat.pos2site[p]] = a
    return nib, dbot


def run_group(flat, nib, dbot, c0, c1, want_scores=None):
    """One wave's work: preamble of chunk c0, bodies of chunks [c0, c1)."""
    slots = {}
    best, cnt, key_best = 0x7FFFFFFF, 0, 0
    dcur = 0
    node_idx = int(flat.chunk_node_off[c0])
    for words, lo, hi in ((flat.pre_stream, int(flat.chunk_pre_off[c0]), int(flat.chunk_pre_off[c0 + 1])),
                          (flat.stream, int(flat.chunk_body_off[c0]), int(flat.chunk_body_off[c1]))):
        i = lo
        while i < hi:
            w0 = int(words[i]); key = int(words[i + 1]); i += 2
            nmut = w0 & 0xFFFF
            rslot, wslot = (w0 >> 16) & 63, (w0 >> 22) & 63
            if rslot == RS_REG:
                dpar = dcur
            elif rslot == RS_BOTTOM:
                dpar = dbot
            else:
                dpar = slots[rslot]
            tsum = neg = common = n_before = 0
            for _ in range(nmut):
                w = int(words[i]); i += 1
                site, mi, pi = w & 0x3FFFFF, (w >> 22) & 3, (w >> 24) & 3
                x = int(nib[site])
                c, p = (x >> mi) & 1, (x >> pi) & 1
                d = p - c
                tsum += d
                if not (w & M_AFTER_MASK):
                    n_before += 1
                    common += c
                    neg += min(d, 0)
            dn = dpar + tsum
            if wslot != WS_NONE:
                slots[wslot] = dn
            dcur = dn
            if not (w0 & F_NOSCORE):
                if w0 & F_ROOT:
                    cost, elig, hu = dn, True, 0
                else:
                    cost = dpar + neg
                    masked = bool(w0 & F_MASKED)
                    elig = common > 0 or (not (w0 & F_LEAF) and not masked and nmut == 0)
                    hu = 1 if (masked or common != n_before) else 0
                if want_scores is not None:
                    want_scores[int(flat.dfs2bfs[node_idx])] = cost + (0 if elig else 1)
                node_idx += 1
                if elig:
                    k = key | hu
                    if cost < best:
                        best, cnt, key_best = cost, 1, k
                    elif cost == best:
                        cnt += 1
                        key_best = max(key_best, k)
    return best, cnt, key_best


def place(flat, sample, n_groups=1, want_scores=False):
    nib, dbot = sample_site_alleles(flat, sample)
    n_chunks = len(flat.chunk_body_off) - 1
    n_groups = max(1, min(n_groups, n_chunks))
    scores = np.zeros(len(flat.dfs2bfs), dtype=np.int64) if want_scores else None
    best, cnt, key = 0x7FFFFFFF, 0, 0
    for g in range(n_groups):
        c0, c1 = g * n_chunks // n_groups, (g + 1) * n_chunks // n_groups
        if c0 >= c1:
            continue
        b, c, k = run_group(flat, nib, dbot, c0, c1, scores)
        if c == 0:
            continue
        if b < best:
            best, cnt, key = b, c, k
        elif b == best:
            cnt += c
            key = max(key, k)
    out = {"best": best, "num_best": cnt, "best_j": int(flat.rank2bfs[key >> 1]), "has_unique": bool(key & 1)}
    if want_scores:
        out["scores"] = scores
    return out
