"""ctypes binding of oracle/libugp_oracle.so (TEST INFRASTRUCTURE ONLY)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libugp_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "ugp_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libugp_oracle.so"] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        P = C.c_void_p
        L.orc_tree_create.restype = P
        L.orc_tree_create.argtypes = [C.c_int64, P, P, P, P, P, P]
        L.orc_tree_destroy.argtypes = [P]
        L.orc_tree_num_leaves.restype = C.c_int64
        L.orc_tree_num_leaves.argtypes = [P, C.c_int64]
        L.orc_place_sample.restype = C.c_int
        L.orc_place_sample.argtypes = [P, C.c_int64, P, P, P, P, C.c_int, P, P, P, P, P, P, C.c_int64, P]
        L.orc_place_sample_mt.restype = C.c_int
        L.orc_place_sample_mt.argtypes = [P, C.c_int64, P, P, P, P, C.c_int, P, P, P]
        L.orc_place_sample_list.restype = C.c_int
        L.orc_place_sample_list.argtypes = [P, C.c_int64, P, P, P, P, C.c_int64, P, P, P, C.c_int, C.c_int32, C.c_int64, P, P, P, P, P, P, C.c_int64, P]
        L.orc_node_has_unique_prefix.restype = C.c_int
        L.orc_node_has_unique_prefix.argtypes = [P, C.c_int64, P, P, P, P, C.c_int64, P]
        L.orc_cf_create.restype = P
        L.orc_cf_create.argtypes = [P]
        L.orc_cf_destroy.argtypes = [P]
        L.orc_cf_place_batch.restype = C.c_int
        L.orc_cf_place_batch.argtypes = [P, C.c_int64, P, P, P, P, P, C.c_int, P, P, P, P, P, P, C.c_int64]
        L.orc_cf_scores.restype = C.c_int
        L.orc_cf_scores.argtypes = [P, C.c_int64, P, P, P, P, P]
        L.orc_node_vecs.restype = C.c_int
        L.orc_node_vecs.argtypes = [P, C.c_int64, P, P, P, P, C.c_int64, C.c_int64] + [P] * 12
        L.orc_fitch_site.restype = C.c_int
        L.orc_fitch_site.argtypes = [C.c_int64, P, C.c_int8, C.c_int64, P, P, P, P, P]
        L.orc_add_mutations.restype = C.c_int
        L.orc_add_mutations.argtypes = [C.c_int64, P, P, P, P, C.c_int64, P, P, P, P, P]
        _lib = L
    return _lib


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleTree:
    """Holds an orc_tree built from BFS-order flat arrays (oracle/refio.tree_to_bfs_arrays)."""

    def __init__(self, arrays: dict):
        self.arrays = arrays
        self.n = int(arrays["n"])
        self._keep = [np.ascontiguousarray(arrays[k]) for k in
                      ("parent", "mut_off", "mut_pos", "mut_ref", "mut_par", "mut_nuc")]
        assert self._keep[0].dtype == np.int64 and self._keep[1].dtype == np.int64
        assert self._keep[2].dtype == np.int32
        self.h = lib().orc_tree_create(self.n, *[_p(a) for a in self._keep])
        if not self.h:
            raise ValueError("orc_tree_create failed (parent[] must be BFS ordered)")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_tree_destroy(self.h)
            self.h = None

    def num_leaves(self, j: int) -> int:
        return int(lib().orc_tree_num_leaves(self.h, j))

    def place(self, sample: dict, compute_scores: bool = False, want_ties: bool = True, tie_cap: int = 1 << 20):
        pos = np.ascontiguousarray(sample["pos"], dtype=np.int32)
        ref = np.ascontiguousarray(sample["ref"], dtype=np.int8)
        nuc = np.ascontiguousarray(sample["nuc"], dtype=np.int8)
        mis = np.ascontiguousarray(sample["is_missing"], dtype=np.int8)
        best = np.zeros(1, np.int32)
        nb = np.zeros(1, np.int64)
        bj = np.zeros(1, np.int64)
        hu = np.zeros(1, np.int8)
        scores = np.zeros(self.n, np.int32) if compute_scores else None
        cap = min(tie_cap, self.n) if want_ties else 0
        ties = np.zeros(max(cap, 1), np.int64)
        thu = np.zeros(max(cap, 1), np.int8)
        rc = lib().orc_place_sample(self.h, len(pos), _p(pos), _p(ref), _p(nuc), _p(mis), int(compute_scores),
                                    _p(best), _p(nb), _p(bj), _p(hu), _p(scores),
                                    _p(ties) if want_ties else None, cap, _p(thu) if want_ties else None)
        assert rc == 0
        k = min(int(nb[0]), cap)
        return {
            "best": int(best[0]), "num_best": int(nb[0]), "best_j": int(bj[0]), "has_unique": bool(hu[0]),
            "scores": scores, "ties": ties[:k].copy(), "ties_has_unique": thu[:k].astype(bool),
        }

    def place_list(self, sample: dict, nodes, jidx=None, distance=None, compute_scores: bool = False, init_best: int = 10 ** 9,
                   init_best_distance: int = 10 ** 9, tie_cap: int = 1 << 16):
        """mapper2_body over a caller-supplied node vector (the matUtils / ripples call sites): `nodes` = BFS indices,
        jidx = the index j handed to mapper2_body for each (default: the position), distance per entry."""
        pos = np.ascontiguousarray(sample["pos"], dtype=np.int32)
        ref = np.ascontiguousarray(sample["ref"], dtype=np.int8)
        nuc = np.ascontiguousarray(sample["nuc"], dtype=np.int8)
        mis = np.ascontiguousarray(sample["is_missing"], dtype=np.int8)
        nodes = np.ascontiguousarray(nodes, dtype=np.int64)
        jidx = None if jidx is None else np.ascontiguousarray(jidx, dtype=np.int64)
        distance = None if distance is None else np.ascontiguousarray(distance, dtype=np.int64)
        best = np.zeros(1, np.int32); nb = np.zeros(1, np.int64); bj = np.zeros(1, np.int64); hu = np.zeros(1, np.int8)
        scores = np.zeros(len(nodes), np.int32) if compute_scores else None
        ties = np.zeros(tie_cap, np.int64); thu = np.zeros(tie_cap, np.int8)
        rc = lib().orc_place_sample_list(self.h, len(pos), _p(pos), _p(ref), _p(nuc), _p(mis), len(nodes), _p(nodes), _p(jidx), _p(distance),
                                         int(compute_scores), int(init_best), int(init_best_distance), _p(best), _p(nb), _p(bj), _p(hu),
                                         _p(scores), _p(ties), tie_cap, _p(thu))
        assert rc == 0
        k = min(int(nb[0]), tie_cap)
        return {"best": int(best[0]), "num_best": int(nb[0]), "best_j": int(bj[0]), "has_unique": bool(hu[0]), "scores": scores,
                "ties": ties[:k].copy(), "ties_has_unique": thu[:k].astype(bool)}

    def node_has_unique_prefix(self, sample: dict, k: int) -> np.ndarray:
        pos = np.ascontiguousarray(sample["pos"], dtype=np.int32)
        ref = np.ascontiguousarray(sample["ref"], dtype=np.int8)
        nuc = np.ascontiguousarray(sample["nuc"], dtype=np.int8)
        mis = np.ascontiguousarray(sample["is_missing"], dtype=np.int8)
        out = np.zeros(max(k, 1), np.int8)
        assert lib().orc_node_has_unique_prefix(self.h, len(pos), _p(pos), _p(ref), _p(nuc), _p(mis), k, _p(out)) == 0
        return out[:k].astype(bool)

    def place_mt(self, sample: dict, nthreads: int):
        pos = np.ascontiguousarray(sample["pos"], dtype=np.int32)
        ref = np.ascontiguousarray(sample["ref"], dtype=np.int8)
        nuc = np.ascontiguousarray(sample["nuc"], dtype=np.int8)
        mis = np.ascontiguousarray(sample["is_missing"], dtype=np.int8)
        best = np.zeros(1, np.int32)
        nb = np.zeros(1, np.int64)
        bj = np.zeros(1, np.int64)
        rc = lib().orc_place_sample_mt(self.h, len(pos), _p(pos), _p(ref), _p(nuc), _p(mis), int(nthreads),
                                       _p(best), _p(nb), _p(bj))
        assert rc == 0
        return {"best": int(best[0]), "num_best": int(nb[0]), "best_j": int(bj[0])}

    def node_vecs(self, sample: dict, j: int, cap: int = 65536):
        pos = np.ascontiguousarray(sample["pos"], dtype=np.int32)
        ref = np.ascontiguousarray(sample["ref"], dtype=np.int8)
        nuc = np.ascontiguousarray(sample["nuc"], dtype=np.int8)
        mis = np.ascontiguousarray(sample["is_missing"], dtype=np.int8)
        ex = [np.zeros(cap, np.int32)] + [np.zeros(cap, np.int8) for _ in range(3)]
        im = [np.zeros(cap, np.int32)] + [np.zeros(cap, np.int8) for _ in range(3)]
        nex = np.zeros(1, np.int64)
        nim = np.zeros(1, np.int64)
        sd = np.zeros(1, np.int32)
        hu = np.zeros(1, np.int8)
        rc = lib().orc_node_vecs(self.h, len(pos), _p(pos), _p(ref), _p(nuc), _p(mis), j, cap,
                                 *[_p(a) for a in ex], _p(nex), *[_p(a) for a in im], _p(nim), _p(sd), _p(hu))
        assert rc == 0

        def pack(arrs, n):
            n = min(int(n), cap)
            return [(int(arrs[0][i]), int(arrs[1][i]), int(arrs[2][i]), int(arrs[3][i])) for i in range(n)]

        return {"excess": pack(ex, nex[0]), "imputed": pack(im, nim[0]), "set_difference": int(sd[0]),
                "has_unique": bool(hu[0])}


class ClosedFormC:
    """The closed form of mapper2_body (oracle/closed_form.py) as an O(N + M) C sweep per sample, sample-parallel:
    the checker for EVERY sample of a full-size batch.  Built on an OracleTree (shares its arrays)."""

    def __init__(self, ot: OracleTree):
        self.ot = ot
        self.h = lib().orc_cf_create(ot.h)
        if not self.h:
            raise ValueError("orc_cf_create failed (tree alleles must be one-hot)")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_cf_destroy(self.h)
            self.h = None

    def place_csr(self, ent_off, pos, ref, nuc, is_missing, nthreads: int = 0, tie_cap: int = 0):
        """Structured result per sample: best, num_best, best_j, has_unique (+ ties lists when tie_cap > 0)."""
        ent_off = np.ascontiguousarray(ent_off, dtype=np.int64)
        pos = np.ascontiguousarray(pos, dtype=np.int32)
        ref = np.ascontiguousarray(ref).astype(np.int8)
        nuc = np.ascontiguousarray(nuc).astype(np.int8)
        mis = np.ascontiguousarray(is_missing).astype(np.int8)
        n = len(ent_off) - 1
        best = np.zeros(n, np.int32)
        nb = np.zeros(n, np.int64)
        bj = np.zeros(n, np.int64)
        hu = np.zeros(n, np.int8)
        ties = np.zeros((n, tie_cap), np.int64) if tie_cap else None
        thu = np.zeros((n, tie_cap), np.int8) if tie_cap else None
        rc = lib().orc_cf_place_batch(self.h, n, _p(ent_off), _p(pos), _p(ref), _p(nuc), _p(mis), int(nthreads or os.cpu_count() or 1),
                                      _p(best), _p(nb), _p(bj), _p(hu), _p(ties), _p(thu), tie_cap)
        if rc != 0:
            raise ValueError("orc_cf_place_batch: sample rows must be sorted by position without duplicates")
        out = {"best": best, "num_best": nb, "best_j": bj, "has_unique": hu.astype(bool)}
        if tie_cap:
            out["ties"] = [ties[i, :min(int(nb[i]), tie_cap)].copy() for i in range(n)]
            out["ties_has_unique"] = [thu[i, :min(int(nb[i]), tie_cap)].astype(bool) for i in range(n)]
        return out

    def place(self, sample: dict, tie_cap: int = 0):
        off = np.array([0, len(sample["pos"])], np.int64)
        r = self.place_csr(off, sample["pos"], sample["ref"], sample["nuc"], sample["is_missing"], 1, tie_cap)
        out = {"best": int(r["best"][0]), "num_best": int(r["num_best"][0]), "best_j": int(r["best_j"][0]), "has_unique": bool(r["has_unique"][0])}
        if tie_cap:
            out["ties"] = r["ties"][0]
            out["ties_has_unique"] = r["ties_has_unique"][0]
        return out

    def scores(self, sample: dict) -> np.ndarray:
        pos = np.ascontiguousarray(sample["pos"], dtype=np.int32)
        ref = np.ascontiguousarray(sample["ref"]).astype(np.int8)
        nuc = np.ascontiguousarray(sample["nuc"]).astype(np.int8)
        mis = np.ascontiguousarray(sample["is_missing"]).astype(np.int8)
        out = np.zeros(self.ot.n, np.int32)
        rc = lib().orc_cf_scores(self.h, len(pos), _p(pos), _p(ref), _p(nuc), _p(mis), _p(out))
        if rc != 0:
            raise ValueError("orc_cf_scores failed")
        return out


def fitch_site(parent: np.ndarray, ref_nuc: int, var_node: np.ndarray, var_nuc: np.ndarray):
    n = len(parent)
    parent = np.ascontiguousarray(parent, dtype=np.int64)
    var_node = np.ascontiguousarray(var_node, dtype=np.int64)
    var_nuc = np.ascontiguousarray(var_nuc, dtype=np.int8)
    state = np.zeros(n, np.int8)
    mpar = np.zeros(n, np.int8)
    mnuc = np.zeros(n, np.int8)
    rc = lib().orc_fitch_site(n, _p(parent), ref_nuc, len(var_node), _p(var_node), _p(var_nuc),
                              _p(state), _p(mpar), _p(mnuc))
    if rc != 0:
        raise ValueError("orc_fitch_site failed")
    return state, mpar, mnuc
