"""Seeded synthetic MAT / query generator (bench and test input; C++ in
csrc/host/ugs_synth.cpp).  Returns the BFS-order flat arrays the C ABI takes."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "libugp_synth.so")
_lib = None


def _L():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            raise RuntimeError("libugp_synth.so is not built; run `make -C usher_amd/csrc`")
        L = C.CDLL(_PATH)
        P = C.c_void_p
        L.ugs_tree_create2.restype = P
        L.ugs_tree_create2.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
        L.ugs_tree_adopt.restype = P
        L.ugs_tree_adopt.argtypes = [C.c_uint64, P, P, P, P, P, P, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
        L.ugs_tree_destroy.argtypes = [P]
        for n in ("nodes", "muts"):
            getattr(L, "ugs_tree_" + n).restype = C.c_uint64
            getattr(L, "ugs_tree_" + n).argtypes = [P]
        for n in ("parent", "mut_off", "mut_pos", "mut_ref", "mut_par", "mut_nuc"):
            getattr(L, "ugs_tree_" + n).restype = P
            getattr(L, "ugs_tree_" + n).argtypes = [P]
        L.ugs_queries_create2.restype = P
        L.ugs_queries_create2.argtypes = [P, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
        L.ugs_queries_destroy.argtypes = [P]
        for n in ("count", "entries"):
            getattr(L, "ugs_queries_" + n).restype = C.c_uint64
            getattr(L, "ugs_queries_" + n).argtypes = [P]
        for n in ("ent_off", "pos", "ref", "nuc", "missing", "source"):
            getattr(L, "ugs_queries_" + n).restype = P
            getattr(L, "ugs_queries_" + n).argtypes = [P]
        _lib = L
    return _lib


def _arr(ptr, n, dt):
    if n == 0:
        return np.zeros(0, dt)
    buf = (C.c_char * (n * np.dtype(dt).itemsize)).from_address(ptr)
    return np.frombuffer(buf, dtype=dt).copy()


SHAPES = {"random": 0, "sars2": 1}


class SynthTree:
    def __init__(self, target_nodes: int, genome_len: int = 29903, n_sites: int = 1500, seed: int = 1, shape: str = "random"):
        L = _L()
        self._h = L.ugs_tree_create2(target_nodes, genome_len, n_sites, seed, SHAPES[shape])
        if not self._h:
            raise ValueError("ugs_tree_create failed")
        self._load(genome_len)

    @classmethod
    def from_arrays(cls, arrays: dict, genome_len: int = 29903, n_sites: int = 1500, seed: int = 1, shape: str = "random") -> "SynthTree":
        """Adopt tree arrays another process generated with the same (seed, genome_len, n_sites): keeps the query generator usable."""
        L = _L()
        self = cls.__new__(cls)
        par = np.asarray(arrays["parent"]).astype(np.int64)
        par32 = np.ascontiguousarray(np.where(par < 0, 0xFFFFFFFF, par).astype(np.uint32))
        keep = [par32, np.ascontiguousarray(arrays["mut_off"], dtype=np.uint64), np.ascontiguousarray(arrays["mut_pos"], dtype=np.int32)] + \
               [np.ascontiguousarray(arrays[k]).astype(np.uint8) for k in ("mut_ref", "mut_par", "mut_nuc")]
        self._h = L.ugs_tree_adopt(len(par32), *[a.ctypes.data_as(C.c_void_p) for a in keep], genome_len, n_sites, seed, SHAPES[shape])
        if not self._h:
            raise MemoryError("ugs_tree_adopt failed")
        self._load(genome_len)
        return self

    def _load(self, genome_len: int):
        L = _L()
        n, m = L.ugs_tree_nodes(self._h), L.ugs_tree_muts(self._h)
        par = _arr(L.ugs_tree_parent(self._h), n, np.uint32).astype(np.int64)
        par[par == 0xFFFFFFFF] = -1
        self.arrays = {
            "n": int(n), "parent": par, "mut_off": _arr(L.ugs_tree_mut_off(self._h), n + 1, np.uint64).astype(np.int64),
            "mut_pos": _arr(L.ugs_tree_mut_pos(self._h), m, np.int32), "mut_ref": _arr(L.ugs_tree_mut_ref(self._h), m, np.uint8).astype(np.int8),
            "mut_par": _arr(L.ugs_tree_mut_par(self._h), m, np.uint8).astype(np.int8), "mut_nuc": _arr(L.ugs_tree_mut_nuc(self._h), m, np.uint8).astype(np.int8),
        }
        self.genome_len = genome_len

    def queries(self, n_queries: int, seed: int = 1, max_subst: int = 3, n_lo: int = 0, n_hi: int = 0, iupac_hi: int = 0, recent: bool = False,
                iupac_true: bool = False, min_subst: int = 0, ref_every_8th: bool = False):
        """CSR query arrays: (ent_off, pos, ref, nuc, is_missing, source_node).  iupac_true: every ambiguity code holds the
        sample's own base (default: any set of 2-3 bases, i.e. most such cells are mismatches).  min_subst .. max_subst
        substitutions per sample (uniform); ref_every_8th: every 8th sample is the all-reference sample (no rows) -- together the
        "far" queries: samples that are not in the neighbourhood of any node."""
        L = _L()
        q = L.ugs_queries_create2(self._h, n_queries, seed, max_subst, n_lo, n_hi, iupac_hi,
                                  (1 if recent else 0) | (2 if iupac_true else 0) | (4 if ref_every_8th else 0) | (int(min_subst) << 8))
        if not q:
            raise MemoryError("ugs_queries_create failed")
        try:
            nq, ne = L.ugs_queries_count(q), L.ugs_queries_entries(q)
            return {
                "ent_off": _arr(L.ugs_queries_ent_off(q), nq + 1, np.uint64), "pos": _arr(L.ugs_queries_pos(q), ne, np.int32),
                "ref": _arr(L.ugs_queries_ref(q), ne, np.uint8), "nuc": _arr(L.ugs_queries_nuc(q), ne, np.uint8),
                "is_missing": _arr(L.ugs_queries_missing(q), ne, np.uint8), "source": _arr(L.ugs_queries_source(q), nq, np.uint32),
            }
        finally:
            L.ugs_queries_destroy(q)

    def close(self):
        if getattr(self, "_h", None):
            _L().ugs_tree_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def csr_sample(q: dict, i: int) -> dict:
    b, e = int(q["ent_off"][i]), int(q["ent_off"][i + 1])
    return {"name": "Q%d" % i, "pos": q["pos"][b:e], "ref": q["ref"][b:e].astype(np.int8), "nuc": q["nuc"][b:e].astype(np.int8),
            "is_missing": q["is_missing"][b:e].astype(np.int8)}
