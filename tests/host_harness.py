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


class Backend(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("place", PLACE_FN), ("scores", SCORES_FN), ("ties", TIES_FN), ("last_error", ERR_FN),
                ("fitch", FITCH_FN), ("fitch_get", FITCH_GET_FN)]


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
    def __init__(self):
        self.version = None
        self.tree = None
        self.calls = 0

        def tree_for(t, version):
            if self.version != version:
                self.tree = capi.OracleTree(_tree_arrays(t.contents))
                self.version = version
            return self.tree

        def place(ctx, t, version, q, out):
            ot = tree_for(t, version)
            res = np.zeros((int(q.contents.n_queries), 4), np.int32)
            for i, s in enumerate(_samples(q.contents)):
                r = ot.place(s, want_ties=False)
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
                r = ot.place(s)
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

        self._fitch = []
        self._cbs = (PLACE_FN(place), SCORES_FN(scores), TIES_FN(ties), ERR_FN(lambda ctx: b"oracle backend"),
                     FITCH_FN(fitch), FITCH_GET_FN(fitch_get))
        self.struct = Backend(None, *self._cbs)


def run_usher(args, backend=None):
    """Run the front end in-process with the oracle backend; returns the exit code."""
    L = C.CDLL(HOST_LIB)
    L.uh_usher_main.restype = C.c_int
    L.uh_usher_main.argtypes = [C.c_int, C.POINTER(C.c_char_p), C.POINTER(Backend)]
    be = backend or OracleBackend()
    argv = [b"usher"] + [a.encode() for a in args]
    arr = (C.c_char_p * len(argv))(*argv)
    return L.uh_usher_main(len(argv), arr, C.byref(be.struct))
