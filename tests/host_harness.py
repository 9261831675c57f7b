"""Test helper: run the product's usher-compatible front end (libusher_host.so,
usher_amd/csrc/host) on CPU with the ORACLE plugged in as the placement
backend.  This checks the host side -- loaders, VCF ingest, tree update, every
output file -- against the recorded reference outputs without a GPU.  On the
GPU box the same front end runs as bin/usher-amd with the HIP backend."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from oracle import capi
from usher_amd import _lib

HOST_LIB = os.path.join(os.path.dirname(_lib.LIB_PATH), "libusher_host.so")

PLACE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_lib.ugp_tree_desc), C.c_uint64, C.POINTER(_lib.ugp_queries), C.c_void_p)
SCORES_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_lib.ugp_tree_desc), C.c_uint64, C.POINTER(_lib.ugp_queries), C.c_void_p)
TIES_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_lib.ugp_tree_desc), C.c_uint64, C.POINTER(_lib.ugp_queries), C.c_uint32,
                      C.c_void_p, C.c_void_p, C.c_void_p)
ERR_FN = C.CFUNCTYPE(C.c_char_p, C.c_void_p)
FITCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(_lib.ugp_sites), C.POINTER(C.c_uint64))
FITCH_GET_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
UPDATE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_lib.ugp_touched), C.c_void_p, C.c_uint64, C.POINTER(C.c_uint32))
T_OPEN_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_lib.ugp_queries))
T_SCORE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, C.c_uint64)
T_RESCORE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64)
T_FETCH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
WARM_FN = C.CFUNCTYPE(None, C.c_void_p)
PREPARE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(_lib.ugp_tree_desc), C.c_uint64)
INT_MAX = 2 ** 31 - 1


class Backend(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("place", PLACE_FN), ("scores", SCORES_FN), ("ties", TIES_FN), ("last_error", ERR_FN),
                ("fitch", FITCH_FN), ("fitch_get", FITCH_GET_FN), ("update", UPDATE_FN), ("touched_open", T_OPEN_FN), ("touched_score", T_SCORE_FN),
                ("touched_rescore", T_RESCORE_FN), ("touched_fetch", T_FETCH_FN), ("warm", WARM_FN), ("prepare", PREPARE_FN)]


def _arr(ptr, n, dt):
    if n == 0 or not ptr:
        return np.zeros(0, dt)
    return np.frombuffer((C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr), dtype=dt).copy()


def _tree_arrays(t):
    n = int(t.n_nodes)
    mut_off = _arr(t.mut_off, n + 1, np.uint64).astype(np.int64)
    m = int(mut_off[-1])
    parent = _arr(t.parent, n, np.uint32).astype(np.int64)
    parent[parent == 0xFFFFFFFF] = -1
    return {"n": n, "parent": parent, "mut_off": mut_off, "mut_pos": _arr(t.mut_pos, m, np.int32),
            "mut_ref": _arr(t.mut_ref, m, np.uint8).astype(np.int8), "mut_par": _arr(t.mut_par, m, np.uint8).astype(np.int8),
            "mut_nuc": _arr(t.mut_nuc, m, np.uint8).astype(np.int8)}


def _samples(q):
    nq = int(q.n_queries)
    off = _arr(q.ent_off, nq + 1, np.uint64).astype(np.int64)
    ne = int(off[-1]) if nq else 0
    pos, ref = _arr(q.pos, ne, np.int32), _arr(q.ref, ne, np.uint8).astype(np.int8)
    nuc, mis = _arr(q.nuc, ne, np.uint8).astype(np.int8), _arr(q.is_missing, ne, np.uint8).astype(np.int8)
    return [{"pos": pos[off[i]:off[i + 1]], "ref": ref[off[i]:off[i + 1]], "nuc": nuc[off[i]:off[i + 1]],
             "is_missing": mis[off[i]:off[i + 1]]} for i in range(nq)]


class OracleBackend:
    """add_mode=True also provides the add-mode callbacks (Backend::update, touched_*) with the semantics of the device library:
    records of touched nodes scored in closed form (numpy), running per-sample results that are merged into and can go stale,
    flattened nodes named by a record excluded from place / ties (the oracle then searches the list of the remaining nodes)."""

    def __init__(self, add_mode: bool = False):
        self.version = None
        self.tree = None
        self.calls = 0
        self.add_mode = add_mode
        self.records = []        # id -> dict(flags, path (pos, allele, ref) arrays, own (pos, mut, prev, ref) arrays), or None once retired
        self.excluded = set()
        self.allowed = None      # cached list of the nodes that are still candidates
        self.batch = None        # dict(dense [Q][P], dbot [Q], best [Q], lists [Q] of (id, hu))
        self.stat = {"update": 0, "open": 0, "score": 0, "rescore": 0, "fetch": 0}

        def tree_for(t, version):
            if self.version != version:
                self.tree = capi.OracleTree(_tree_arrays(t.contents))
                self.version = version
            return self.tree

        def search(ot, s, want_ties):
            if not self.excluded:
                return ot.place(s, want_ties=want_ties)
            if self.allowed is None:
                self.allowed = np.array([j for j in range(ot.n) if j not in self.excluded], np.int64)
            r = ot.place_list(s, self.allowed, tie_cap=1 << 16)
            r["best_j"] = int(self.allowed[r["best_j"]])
            r["ties"] = self.allowed[r["ties"]]
            return r

        def place(ctx, t, version, q, out):
            ot = tree_for(t, version)
            res = np.zeros((int(q.contents.n_queries), 4), np.int32)
            for i, s in enumerate(_samples(q.contents)):
                r = search(ot, s, False)
                res[i] = (r["best"], r["num_best"], r["best_j"], int(r["has_unique"]))
            C.memmove(out, res.ctypes.data, res.nbytes)
            self.calls += 1
            return 0

        def scores(ctx, t, version, q, out):
            ot = tree_for(t, version)
            rows = [ot.place(s, compute_scores=True, want_ties=False)["scores"] for s in _samples(q.contents)]
            arr = np.ascontiguousarray(np.stack(rows).astype(np.int32)) if rows else np.zeros((0, 0), np.int32)
            C.memmove(out, arr.ctypes.data, arr.nbytes)
            return 0

        def ties(ctx, t, version, q, cap, tj, th, tc):
            ot = tree_for(t, version)
            ss = _samples(q.contents)
            a_j = np.zeros((len(ss), max(cap, 1)), np.uint32)
            a_h = np.zeros((len(ss), max(cap, 1)), np.uint8)
            a_c = np.zeros(len(ss), np.uint32)
            for i, s in enumerate(ss):
                r = search(ot, s, True)
                k = min(len(r["ties"]), cap)
                a_j[i, :k] = r["ties"][:k]
                a_h[i, :k] = r["ties_has_unique"][:k]
                a_c[i] = r["num_best"]
            if cap:
                C.memmove(tj, a_j.ctypes.data, len(ss) * cap * 4)
                C.memmove(th, a_h.ctypes.data, len(ss) * cap)
            C.memmove(tc, a_c.ctypes.data, a_c.nbytes)
            return 0

        def fitch(ctx, n_nodes, parent, sites, n_out):
            # oracle restatement of mapper_body, one site at a time
            st = sites.contents
            par = _arr(parent, int(n_nodes), np.uint32).astype(np.int64)
            par[par == 0xFFFFFFFF] = -1
            ns = int(st.n_sites)
            ref = _arr(st.ref, ns, np.uint8)
            off = _arr(st.var_off, ns + 1, np.uint64).astype(np.int64) if ns else np.zeros(1, np.int64)
            vn = _arr(st.var_node, int(off[-1]), np.uint32).astype(np.int64)
            vc = _arr(st.var_nuc, int(off[-1]), np.uint8).astype(np.int8)
            rows = []
            for s in range(ns):
                _, mpar, mnuc = capi.fitch_site(par, int(ref[s]), vn[off[s]:off[s + 1]], vc[off[s]:off[s + 1]])
                for j in np.flatnonzero(mnuc):
                    rows.append((s, int(j), int(mpar[j]), int(mnuc[j])))
            self._fitch = rows
            n_out[0] = len(rows)
            return 0

        def fitch_get(ctx, site, node, mpar, mnuc):
            rows = self._fitch
            for ptr, col, dt in ((site, 0, np.uint32), (node, 1, np.uint32), (mpar, 2, np.uint8), (mnuc, 3, np.uint8)):
                a = np.array([r[col] for r in rows], dt)
                if len(a):
                    C.memmove(ptr, a.ctypes.data, a.nbytes)
            self._fitch = []
            return 0

        # ---- add mode: the device library's semantics restated (include/usher_amd.h "add mode") ---------------------------
        def update(ctx, recs, retired, n_retired, first_id):
            self.stat["update"] += 1
            first_id[0] = len(self.records)
            for rid in _arr(retired, int(n_retired), np.uint32):
                self.records[int(rid)] = None
            if recs:
                t = recs.contents
                n = int(t.n)
                flat_j, flags, n_path = _arr(t.flat_j, n, np.uint32), _arr(t.flags, n, np.uint8), _arr(t.n_path, n, np.uint32)
                off = _arr(t.ent_off, n + 1, np.uint64).astype(np.int64)
                ne = int(off[-1]) if n else 0
                pos, al, pv, rf = _arr(t.pos, ne, np.int32).astype(np.int64), _arr(t.allele, ne, np.uint8), _arr(t.prev, ne, np.uint8), _arr(t.ref, ne, np.uint8)
                for i in range(n):
                    b, m, e = int(off[i]), int(off[i]) + int(n_path[i]), int(off[i + 1])
                    self.records.append({"flags": int(flags[i]), "path": (pos[b:m], al[b:m], rf[b:m]), "own": (pos[m:e], al[m:e], pv[m:e], rf[m:e])})
                    if flat_j[i] != 0xFFFFFFFF:
                        assert flat_j[i] != 0
                        self.excluded.add(int(flat_j[i]))
                        self.allowed = None
            return 0

        def score_records(ids, q0, q1):
            """merge the records `ids` into the running results of samples [q0, q1): minimum first, then the lists"""
            B = self.batch
            dense, dbot = B["dense"][q0:q1], B["dbot"][q0:q1]
            P = dense.shape[1]
            costs = {}
            for rid in ids:
                rec = self.records[rid]
                if rec is None:
                    continue

                def S(pos, ref):                      # allele sets of the samples at the entry positions: [q][k]
                    inside = pos < P
                    row = np.where(inside[None, :], dense[:, np.where(inside, pos, 0)], 0)
                    return np.where(row != 0, row, ref[None, :])
                pp, pa, pr = rec["path"]
                sp = S(pp, pr)
                D = dbot + ((sp & pa[None, :]) == 0).sum(1) - ((sp & pr[None, :]) == 0).sum(1)
                op, om, ov, orf = rec["own"]
                so = S(op, orf)
                c = (so & om[None, :]) != 0
                p = (so & ov[None, :]) != 0
                common = c.sum(1)
                neg = np.minimum(p.astype(np.int64) - c.astype(np.int64), 0).sum(1)
                masked, leaf = bool(rec["flags"] & 2), bool(rec["flags"] & 1)
                num_mut = len(op) + (1 if masked else 0)
                elig = (common > 0) | ((not leaf) and num_mut == 0)
                hu = np.full(len(D), masked) | (common != num_mut)
                costs[rid] = (np.where(elig, D + neg, INT_MAX), hu)
            for q in range(q0, q1):
                best = min([B["best"][q]] + [int(c[q - q0]) for c, _ in costs.values()])
                if best < B["best"][q]:
                    B["best"][q] = best
                    B["lists"][q] = []
                if best != INT_MAX:
                    for rid, (c, hu) in costs.items():
                        if int(c[q - q0]) == best:
                            B["lists"][q].append((rid, bool(hu[q - q0])))

        def touched_open(ctx, q):
            self.stat["open"] += 1
            ss = _samples(q.contents)
            P = 1 + max([int(s["pos"].max()) for s in ss if len(s["pos"])] + [0])
            dense = np.zeros((len(ss), P), np.uint8)
            dbot = np.zeros(len(ss), np.int64)
            for i, s in enumerate(ss):
                a = np.where(s["is_missing"] != 0, 15, s["nuc"].astype(np.int64) & 15).astype(np.uint8)
                dense[i, s["pos"]] = a
                dbot[i] = int(((s["is_missing"] == 0) & ((a & s["ref"].astype(np.uint8)) == 0)).sum())
            self.batch = {"dense": dense, "dbot": dbot, "best": [INT_MAX] * len(ss), "lists": [[] for _ in ss]}
            score_records(range(len(self.records)), 0, len(ss))
            return 0

        def touched_score(ctx, first_id, first_sample):
            self.stat["score"] += 1
            score_records(range(int(first_id), len(self.records)), int(first_sample), len(self.batch["best"]))
            return 0

        def touched_rescore(ctx, sample):
            self.stat["rescore"] += 1
            q = int(sample)
            self.batch["best"][q] = INT_MAX
            self.batch["lists"][q] = []
            score_records(range(len(self.records)), q, q + 1)
            return 0

        def touched_fetch(ctx, first_sample, n, cap, best, count, ids, hu):
            self.stat["fetch"] += 1
            q0, n, cap = int(first_sample), int(n), int(cap)
            a_best = np.array(self.batch["best"][q0:q0 + n], np.int32)
            a_cnt = np.array([len(l) for l in self.batch["lists"][q0:q0 + n]], np.uint32)
            a_ids = np.zeros((n, max(cap, 1)), np.uint32)
            a_hu = np.zeros((n, max(cap, 1)), np.uint8)
            for i, l in enumerate(self.batch["lists"][q0:q0 + n]):
                for k, (rid, h) in enumerate(l[:cap]):
                    a_ids[i, k], a_hu[i, k] = rid, h
            C.memmove(best, a_best.ctypes.data, a_best.nbytes)
            C.memmove(count, a_cnt.ctypes.data, a_cnt.nbytes)
            if cap:
                C.memmove(ids, a_ids.ctypes.data, n * cap * 4)
                C.memmove(hu, a_hu.ctypes.data, n * cap)
            return 0

        # Backend::warm / ::prepare (round 5): the front end hands the tree over on a thread of its own while the VCF is read; the
        # oracle keeps it -- the first `place` with the same version must find it -- and counts the calls
        self.prepared = []
        self.warmed = 0

        def warm(ctx):
            self.warmed += 1

        def prepare(ctx, t, version):
            self.prepared.append(int(version))
            tree_for(t, int(version))
            return 0
        self._pre = (WARM_FN(warm), PREPARE_FN(prepare))
        self._fitch = []
        self._cbs = (PLACE_FN(place), SCORES_FN(scores), TIES_FN(ties), ERR_FN(lambda ctx: b"oracle backend"),
                     FITCH_FN(fitch), FITCH_GET_FN(fitch_get))
        self._add = (UPDATE_FN(update), T_OPEN_FN(touched_open), T_SCORE_FN(touched_score), T_RESCORE_FN(touched_rescore), T_FETCH_FN(touched_fetch))
        if add_mode:
            self.struct = Backend(None, *self._cbs, *self._add, *self._pre)
        else:
            self.struct = Backend(None, *self._cbs, UPDATE_FN(), T_OPEN_FN(), T_SCORE_FN(), T_RESCORE_FN(), T_FETCH_FN(), *self._pre)


def run_usher(args, backend=None):
    """Run the front end in-process with the oracle backend; returns the exit code."""
    L = C.CDLL(HOST_LIB)
    L.uh_usher_main.restype = C.c_int
    L.uh_usher_main.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(Backend)]
    be = backend or OracleBackend()
    argv = [b"usher"] + [a.encode() for a in args]
    arr = (C.c_char_p * len(argv))(*argv)
    return L.uh_usher_main(len(argv), arr, C.byref(be.struct))
