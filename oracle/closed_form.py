"""oracle/closed_form.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Second, independent CPU restatement of the placement hot path: the closed form
the HIP kernels evaluate (DESIGN.md "What the kernel computes"),
written as plain python/numpy loops.  tests/ check it against the literal
oracle (oracle/ugp_oracle.c) and against the recorded reference outputs; the
GPU path is then checked against both.

For a sample s with allele set S_s(p) at position p ({ref} when the VCF has no
row for s at p, all four bases for a missing call) and node state
state_n(p) (most recent non-masked mutation on root->n, else ref):

    D(n,s)    = sum_p [ state_n(p) not in S_s(p) ]
              = D(parent(n),s) + sum_{m in muts(n), not masked} delta(m,s)
    delta     = [prev(m) in S] - [m.mut in S]           (prev = true parent state)
    D(bottom) = #{ non-missing rows e of s : (e.mask & e.ref) == 0 }
    cost(root)= D(root)
    cost(n)   = D(parent) + sum_{m before the first masked one} min(delta, 0)
    common(n) = #{ m before the first masked one : m.mut in S }
    eligible  = root  or  common > 0  or  (internal and num_mut == 0)
    has_unique= masked or common != num_mut

following usher_mapper.cpp:190-270 (branch loop), :275-445 (ancestor state and
the two scans) and :454-503 (eligibility, reduction, +1 for ineligible nodes
when scores are requested).  Preconditions (checked): sample rows sorted by
position with no duplicates; tree alleles one-hot.
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np


class ClosedFormTree:
    def __init__(self, arrays: dict):
        self.n = int(arrays["n"])
        self.parent = np.asarray(arrays["parent"], dtype=np.int64)
        self.mut_off = np.asarray(arrays["mut_off"], dtype=np.int64)
        self.mut_pos = np.asarray(arrays["mut_pos"], dtype=np.int64)
        self.mut_ref = np.asarray(arrays["mut_ref"], dtype=np.int64)
        self.mut_nuc = np.asarray(arrays["mut_nuc"], dtype=np.int64)
        n = self.n
        self.children: List[List[int]] = [[] for _ in range(n)]
        for j in range(1, n):
            self.children[self.parent[j]].append(j)
        self.num_leaves = np.zeros(n, dtype=np.int64)
        for j in range(n - 1, -1, -1):
            if not self.children[j]:
                self.num_leaves[j] = 1
            if j > 0:
                self.num_leaves[self.parent[j]] += self.num_leaves[j]
        # true parent state of every mutation (prev) from a DFS with an undo log
        self.mut_prev = np.zeros(len(self.mut_pos), dtype=np.int64)
        state: Dict[int, int] = {}
        self.ref_at: Dict[int, int] = {}
        stack = [(0, False)]
        undo: List[List] = []
        while stack:
            j, leaving = stack.pop()
            if leaving:
                for p, old in reversed(undo.pop()):
                    if old is None:
                        del state[p]
                    else:
                        state[p] = old
                continue
            log = []
            for i in range(self.mut_off[j], self.mut_off[j + 1]):
                p = int(self.mut_pos[i])
                if p < 0:
                    continue
                nuc = int(self.mut_nuc[i])
                if nuc & (nuc - 1) or nuc == 0:
                    raise ValueError("tree allele is not one-hot")
                self.ref_at.setdefault(p, int(self.mut_ref[i]))
                self.mut_prev[i] = state.get(p, int(self.mut_ref[i]))
                log.append((p, state.get(p)))
                state[p] = nuc
            undo.append(log)
            stack.append((j, True))
            for c in reversed(self.children[j]):
                stack.append((c, False))

    def place(self, sample: dict, compute_scores: bool = False):
        pos = np.asarray(sample["pos"], dtype=np.int64)
        if len(pos) > 1 and not np.all(pos[1:] > pos[:-1]):
            raise ValueError("sample rows must be sorted by position without duplicates")
        S: Dict[int, int] = {}
        d_bottom = 0
        for p, r, nuc, mis in zip(pos, sample["ref"], sample["nuc"], sample["is_missing"]):
            if mis:
                S[int(p)] = 0xF
            else:
                S[int(p)] = int(nuc)
                if (int(nuc) & int(r)) == 0:
                    d_bottom += 1
        n = self.n
        D = np.zeros(n, dtype=np.int64)
        cost = np.zeros(n, dtype=np.int64)
        elig = np.zeros(n, dtype=bool)
        hu = np.zeros(n, dtype=bool)
        for j in range(n):  # BFS order: parents first
            par = self.parent[j]
            d_par = d_bottom if par < 0 else D[par]
            tsum = 0
            neg = 0
            common = 0
            num_mut = 0
            masked = False
            for i in range(self.mut_off[j], self.mut_off[j + 1]):
                p = int(self.mut_pos[i])
                if p < 0:
                    if not masked:
                        num_mut += 1
                    masked = True
                    continue
                s_p = S.get(p, int(self.mut_ref[i]))
                c = 1 if (s_p & int(self.mut_nuc[i])) else 0
                pr = 1 if (s_p & int(self.mut_prev[i])) else 0
                delta = pr - c
                tsum += delta
                if not masked:
                    num_mut += 1
                    common += c
                    neg += min(delta, 0)
            D[j] = d_par + tsum
            if par < 0:
                cost[j] = D[j]
                elig[j] = True
                hu[j] = False
            else:
                cost[j] = d_par + neg
                leaf = not self.children[j]
                elig[j] = common > 0 or (not leaf and num_mut == 0)
                hu[j] = masked or common != num_mut
        best = int(cost[elig].min())
        tied = np.nonzero(elig & (cost == best))[0]
        key = self.num_leaves[tied] * (n + 1) + tied
        best_j = int(tied[np.argmax(key)])
        out = {"best": best, "num_best": int(len(tied)), "best_j": best_j, "has_unique": bool(hu[best_j]),
               "ties": tied, "ties_has_unique": hu[tied]}
        if compute_scores:
            out["scores"] = (cost + (~elig).astype(np.int64)).astype(np.int32)
        return out
