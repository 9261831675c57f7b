"""Python mirror of the placement C ABI (include/usher_amd.h).

Mirrors, for a batch of samples on a static tree, the reference's per-sample
block usher_common.cpp:342-449 (`mapper2_body` over every BFS node, then the
tie pass): `Placer.place` returns per sample the reference's
`best_set_difference`, `num_best`, `best_j` (BFS index) and `has_unique`
(usher_graph.hpp:79-92).  All compute happens in libusher_amd.so on the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import build_library  # noqa: F401

RESULT_DTYPE = np.dtype([("best_set_difference", np.int32), ("num_best", np.uint32), ("best_j", np.uint32),
                         ("best_has_unique", np.uint32)])


class UgpError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__("ugp error %d: %s" % (code, msg))
        self.code = code


def _check(rc: int) -> None:
    if rc != 0:
        raise UgpError(rc, (_lib.lib().ugp_last_error() or b"").decode())


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class _TreeArrays:
    """Keeps numpy arrays alive and typed for a ugp_tree_desc."""

    def __init__(self, arrays: Dict):
        n = int(arrays["n"])
        parent = np.asarray(arrays["parent"]).astype(np.int64)
        par32 = np.where(parent < 0, 0xFFFFFFFF, parent).astype(np.uint32)
        self.n = n
        self.parent = np.ascontiguousarray(par32)
        self.mut_off = np.ascontiguousarray(arrays["mut_off"], dtype=np.uint64)
        self.mut_pos = np.ascontiguousarray(arrays["mut_pos"], dtype=np.int32)
        self.mut_ref = np.ascontiguousarray(arrays["mut_ref"]).astype(np.uint8)
        self.mut_par = np.ascontiguousarray(arrays["mut_par"]).astype(np.uint8)
        self.mut_nuc = np.ascontiguousarray(arrays["mut_nuc"]).astype(np.uint8)
        self.desc = _lib.ugp_tree_desc(n, _ptr(self.parent), _ptr(self.mut_off), _ptr(self.mut_pos),
                                       _ptr(self.mut_ref), _ptr(self.mut_par), _ptr(self.mut_nuc))


class QueryBatch:
    """CSR batch of query samples (the rows read_vcf() puts in Missing_Sample::mutations)."""

    def __init__(self, samples: Sequence[Dict]):
        self.names = [s.get("name", "q%d" % i) for i, s in enumerate(samples)]
        lens = [len(s["pos"]) for s in samples]
        self.ent_off = np.zeros(len(samples) + 1, dtype=np.uint64)
        if lens:
            self.ent_off[1:] = np.cumsum(lens)
        cat = lambda k, dt: (np.concatenate([np.asarray(s[k]) for s in samples]).astype(dt) if lens and sum(lens)
                             else np.zeros(0, dt))
        self.pos = np.ascontiguousarray(cat("pos", np.int32))
        self.ref = np.ascontiguousarray(cat("ref", np.uint8))
        self.nuc = np.ascontiguousarray(cat("nuc", np.uint8))
        self.is_missing = np.ascontiguousarray(cat("is_missing", np.uint8))
        self.desc = _lib.ugp_queries(len(samples), _ptr(self.ent_off), _ptr(self.pos), _ptr(self.ref), _ptr(self.nuc),
                                     _ptr(self.is_missing))

    @classmethod
    def from_csr(cls, ent_off, pos, ref, nuc, is_missing, names=None) -> "QueryBatch":
        self = cls.__new__(cls)
        self.ent_off = np.ascontiguousarray(ent_off, dtype=np.uint64)
        self.pos = np.ascontiguousarray(pos, dtype=np.int32)
        self.ref = np.ascontiguousarray(ref, dtype=np.uint8)
        self.nuc = np.ascontiguousarray(nuc, dtype=np.uint8)
        self.is_missing = np.ascontiguousarray(is_missing, dtype=np.uint8)
        n = len(self.ent_off) - 1
        self.names = list(names) if names is not None else ["q%d" % i for i in range(n)]
        self.desc = _lib.ugp_queries(n, _ptr(self.ent_off), _ptr(self.pos), _ptr(self.ref), _ptr(self.nuc),
                                     _ptr(self.is_missing))
        return self

    def __len__(self) -> int:
        return len(self.ent_off) - 1

    def slice(self, lo: int, hi: int) -> "QueryBatch":
        e0, e1 = int(self.ent_off[lo]), int(self.ent_off[hi])
        return QueryBatch.from_csr(self.ent_off[lo:hi + 1] - self.ent_off[lo], self.pos[e0:e1], self.ref[e0:e1],
                                   self.nuc[e0:e1], self.is_missing[e0:e1], self.names[lo:hi])


class FlatTreeView:
    """Host-only view of the flattened tree (no GPU needed): what ugp_mat_create uploads."""

    IDS = {"stream": (0, np.uint32), "pre_stream": (1, np.uint32), "chunk_body_off": (2, np.uint32),
           "chunk_pre_off": (3, np.uint32), "chunk_node_off": (4, np.uint32), "pos2site": (5, np.int32),
           "site_ref": (6, np.uint8), "rank2bfs": (7, np.uint32), "dfs2bfs": (8, np.uint32),
           "stream8": (10, np.uint32), "pre8_stream": (11, np.uint32), "chunk8_body_off": (12, np.uint32),
           "chunk8_pre_off": (13, np.uint32), "stream_t": (15, np.uint32), "chunk_t_off": (16, np.uint32),
           "b3_group_off": (18, np.uint32), "b3_events": (19, np.uint32)}

    def __init__(self, arrays: Dict, chunk_nodes: int = 0):
        L = _lib.lib()
        self._t = _TreeArrays(arrays)
        h = C.c_void_p()
        _check(L.ugp_flat_create(C.byref(self._t.desc), chunk_nodes, C.byref(h)))
        try:
            for name, (which, dt) in self.IDS.items():
                p = C.c_void_p()
                n = C.c_uint64()
                _check(L.ugp_flat_get(h, which, C.byref(p), C.byref(n)))
                if n.value:
                    buf = (C.c_char * (n.value * np.dtype(dt).itemsize)).from_address(p.value)
                    setattr(self, name, np.frombuffer(buf, dtype=dt).copy())
                else:
                    setattr(self, name, np.zeros(0, dt))
            p = C.c_void_p()
            n = C.c_uint64()
            _check(L.ugp_flat_get(h, 9, C.byref(p), C.byref(n)))
            self.max_slots = int(n.value)
            _check(L.ugp_flat_get(h, 14, C.byref(p), C.byref(n)))
            self.max_path_muts = int(n.value)
            _check(L.ugp_flat_get(h, 17, C.byref(p), C.byref(n)))
            self.lds_slots = int(n.value)
        finally:
            L.ugp_flat_destroy(h)


class Placer:
    """A flattened MAT resident on one GPU (ugp_mat) plus the batch entry points."""

    def __init__(self, arrays: Dict, device: int = 0, chunk_nodes: Optional[int] = None, experiments: bool = False, flat_file: Optional[str] = None):
        """experiments=True binds libusher_amd_exp.so (built with -DUGP_EXPERIMENTS: UGP_STATS, UGP_TRACE, UGP_SEED_*,
        UGP_PHASE2_PACKED, UGP_KBEST_EXCLUSIVE); the release library ignores those variables.  The tuning switches are read
        from the environment here, once; reload_knobs() reads them again.  flat_file: a flattening of these arrays written by
        save_flat() (ugp_flat_save) -- uploaded as it is, no flattening in this process (the other ranks of a multi-GPU launch)."""
        L = self._L = _lib.lib(experiments)
        self._t = _TreeArrays(arrays)
        self.n_nodes = self._t.n
        self._h = C.c_void_p()
        if flat_file is not None:
            self._ck(L.ugp_mat_create_from_flat(flat_file.encode(), device, C.byref(self._h)))
        elif chunk_nodes is None:
            self._ck(L.ugp_mat_create(C.byref(self._t.desc), device, C.byref(self._h)))
        else:
            self._ck(L.ugp_mat_create_chunked(C.byref(self._t.desc), device, int(chunk_nodes), C.byref(self._h)))
        self.device = device

    def _ck(self, rc: int) -> None:
        if rc != 0:
            raise UgpError(rc, (self._L.ugp_last_error() or b"").decode())

    @staticmethod
    def save_flat(arrays: Dict, path: str, experiments: bool = False) -> None:
        """ugp_flat_save: flatten `arrays` once and write the result to `path` for Placer(..., flat_file=path) in other processes."""
        L = _lib.lib(experiments)
        t = _TreeArrays(arrays)
        rc = L.ugp_flat_save(C.byref(t.desc), path.encode())
        if rc != 0:
            raise UgpError(rc, (L.ugp_last_error() or b"").decode())

    def reload_knobs(self) -> None:
        """ugp_mat_reload_knobs: the per-call tuning switches from the environment again (test / tuning hook)."""
        self._ck(self._L.ugp_mat_reload_knobs(self._h))

    def close(self) -> None:
        if getattr(self, "_h", None) and self._h.value:
            self._L.ugp_mat_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self) -> Dict:
        out = _lib.ugp_info()
        self._ck(self._L.ugp_mat_info(self._h, C.byref(out)))
        return {k: getattr(out, k) for k, _ in out._fields_}

    def place(self, batch: QueryBatch) -> np.ndarray:
        """ugp_place_batch: structured array (best_set_difference, num_best, best_j, best_has_unique)."""
        out = np.zeros(len(batch), dtype=RESULT_DTYPE)
        self._ck(self._L.ugp_place_batch(self._h, C.byref(batch.desc), _ptr(out)))
        return out

    def place_async(self, batch: QueryBatch):
        """ugp_place_batch_async: starts the batch and returns a job; job_wait(job) gives the same array as place().  At most
        pipeline_depth() jobs may be outstanding (two for batches of more than 32,768 samples or of 128 rows per sample and more); the
        batch's arrays may be reused as soon as this returns."""
        out = np.zeros(len(batch), dtype=RESULT_DTYPE)
        job = C.c_void_p()
        self._ck(self._L.ugp_place_batch_async(self._h, C.byref(batch.desc), _ptr(out), C.byref(job)))
        return (job, out)

    def job_wait(self, job) -> np.ndarray:
        h, out = job
        self._ck(self._L.ugp_job_wait(h))
        return out

    def scores_per_node(self, batch: QueryBatch) -> np.ndarray:
        out = np.zeros((len(batch), self.n_nodes), dtype=np.int32)
        self._ck(self._L.ugp_scores_per_node(self._h, C.byref(batch.desc), _ptr(out)))
        return out

    def tied_nodes(self, batch: QueryBatch, cap: int):
        tj = np.zeros((len(batch), max(cap, 1)), dtype=np.uint32)
        th = np.zeros((len(batch), max(cap, 1)), dtype=np.uint8)
        tc = np.zeros(len(batch), dtype=np.uint32)
        self._ck(self._L.ugp_tied_nodes(self._h, C.byref(batch.desc), cap, _ptr(tj), _ptr(th), _ptr(tc)))
        return [tj[i, :min(int(tc[i]), cap)].copy() for i in range(len(batch))], \
               [th[i, :min(int(tc[i]), cap)].astype(bool) for i in range(len(batch))], tc

    # ---- the other callers of mapper2_body (matUtils uncertainty / annotate / merge, ripples) -------------
    def _opts(self, batch, order, node_mask, skip_node, distance, scores):
        keep = [None if a is None else np.ascontiguousarray(a, dtype=dt) for a, dt in
                ((node_mask, np.uint8), (skip_node, np.uint32), (distance, np.uint32))]
        if keep[0] is not None and len(keep[0]) != self.n_nodes or keep[2] is not None and len(keep[2]) != self.n_nodes:
            raise ValueError("node_mask / distance need one entry per node")
        if keep[1] is not None and len(keep[1]) != len(batch):
            raise ValueError("skip_node needs one entry per sample")
        o = _lib.ugp_place_opts({"bfs": 0, "dfs": 1}[order], _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(scores))
        return o, keep

    def place_ex(self, batch: QueryBatch, order: str = "bfs", node_mask=None, skip_node=None, distance=None, want_scores: bool = False):
        """ugp_place_batch_ex: every node index (mask, skip_node, best_j, score columns) is a position in `order`."""
        out = np.zeros(len(batch), dtype=RESULT_DTYPE)
        scores = np.zeros((len(batch), self.n_nodes), dtype=np.int32) if want_scores else None
        o, keep = self._opts(batch, order, node_mask, skip_node, distance, scores)
        self._ck(self._L.ugp_place_batch_ex(self._h, C.byref(batch.desc), C.byref(o), _ptr(out)))
        return (out, scores) if want_scores else out

    def tied_nodes_ex(self, batch: QueryBatch, cap: int, order: str = "bfs", node_mask=None, skip_node=None, distance=None):
        tj = np.zeros((len(batch), max(cap, 1)), dtype=np.uint32)
        th = np.zeros((len(batch), max(cap, 1)), dtype=np.uint8)
        tc = np.zeros(len(batch), dtype=np.uint32)
        o, keep = self._opts(batch, order, node_mask, skip_node, distance, None)
        self._ck(self._L.ugp_tied_nodes_ex(self._h, C.byref(batch.desc), C.byref(o), cap, _ptr(tj), _ptr(th), _ptr(tc)))
        return [tj[i, :min(int(tc[i]), cap)].copy() for i in range(len(batch))], \
               [th[i, :min(int(tc[i]), cap)].astype(bool) for i in range(len(batch))], tc

    def prepare_ex(self, order: str = "bfs", node_mask=None, distance=None):
        """ugp_ex_prepare: the node-level options of an extended search (order, mask, distance) made ready once; returns a handle for
        place_prepared / free_ex."""
        o, keep = self._opts(None, order, node_mask, None, distance, None)
        h = C.c_void_p()
        self._ck(self._L.ugp_ex_prepare(self._h, C.byref(o), C.byref(h)))
        return h

    def free_ex(self, h) -> None:
        if h:
            self._L.ugp_ex_destroy(h)

    def place_prepared(self, batch: QueryBatch, ex, skip_node=None, d_scores: int = 0):
        """ugp_place_batch_prepared; d_scores = device pointer to len(batch) * n_nodes int32 (e.g. torch tensor .data_ptr()), or 0."""
        out = np.zeros(len(batch), dtype=RESULT_DTYPE)
        sk = None if skip_node is None else np.ascontiguousarray(skip_node, dtype=np.uint32)
        if sk is not None and len(sk) != len(batch):
            raise ValueError("skip_node must have one entry per query")
        self._ck(self._L.ugp_place_batch_prepared(self._h, C.byref(batch.desc), ex, None if sk is None else _ptr(sk), _ptr(out), C.c_void_p(d_scores) if d_scores else None))
        return out

    def node_order(self, order: str) -> np.ndarray:
        out = np.zeros(self.n_nodes, dtype=np.uint32)
        self._ck(self._L.ugp_node_order(self._h, {"bfs": 0, "dfs": 1}[order], _ptr(out)))
        return out

    def subtree_mask(self, root_j: int, max_levels: int, order: str = "bfs") -> np.ndarray:
        out = np.zeros(self.n_nodes, dtype=np.uint8)
        self._ck(self._L.ugp_subtree_mask(self._h, {"bfs": 0, "dfs": 1}[order], int(root_j), int(max_levels), _ptr(out)))
        return out

    # ---- add mode: records of the nodes created or rewritten since the tree was flattened -----------------------------
    def update(self, records: Sequence[Dict] = (), retired: Sequence[int] = ()) -> int:
        """ugp_mat_update.  A record is a dict: flat_j (index in the flattened tree, or None for a node created since), leaf, masked,
        path = [(pos, state, ref)] -- the parent's state wherever it is not the reference base --, own = [(pos, mut, prev, ref)].
        Returns the id of the first new record."""
        n = len(records)
        flat_j = np.array([0xFFFFFFFF if r.get("flat_j") is None else r["flat_j"] for r in records], np.uint32)
        flags = np.array([(1 if r.get("leaf") else 0) | (2 if r.get("masked") else 0) for r in records], np.uint8)
        n_path = np.array([len(r["path"]) for r in records], np.uint32)
        off = np.zeros(n + 1, np.uint64)
        pos, al, pv, rf = [], [], [], []
        for i, r in enumerate(records):
            for (p, a, f) in r["path"]:
                pos.append(p); al.append(a); pv.append(0); rf.append(f)
            for (p, m, pr, f) in r["own"]:
                pos.append(p); al.append(m); pv.append(pr); rf.append(f)
            off[i + 1] = len(pos)
        pos = np.array(pos, np.int32); al = np.array(al, np.uint8); pv = np.array(pv, np.uint8); rf = np.array(rf, np.uint8)
        t = _lib.ugp_touched(n, _ptr(flat_j), _ptr(flags), _ptr(n_path), _ptr(off), _ptr(pos), _ptr(al), _ptr(pv), _ptr(rf))
        ret = np.ascontiguousarray(list(retired), dtype=np.uint32)
        first = C.c_uint32()
        self._ck(self._L.ugp_mat_update(self._h, C.byref(t), _ptr(ret), len(ret), C.byref(first)))
        return int(first.value)

    def touched_open(self, batch: QueryBatch) -> None:
        self._ck(self._L.ugp_touched_open(self._h, C.byref(batch.desc)))

    def touched_score(self, first_id: int, first_sample: int) -> None:
        self._ck(self._L.ugp_touched_score(self._h, first_id, first_sample))

    def touched_rescore(self, sample: int) -> None:
        self._ck(self._L.ugp_touched_rescore(self._h, sample))

    def touched_fetch(self, first_sample: int, n: int, cap: int = 64):
        """(best [n] int32 -- INT32_MAX when no record is eligible --, count [n], ids [n][cap], has_unique [n][cap])"""
        best = np.zeros(n, np.int32); cnt = np.zeros(n, np.uint32)
        ids = np.zeros((n, max(cap, 1)), np.uint32); hu = np.zeros((n, max(cap, 1)), np.uint8)
        self._ck(self._L.ugp_touched_fetch(self._h, first_sample, n, cap, _ptr(best), _ptr(cnt), _ptr(ids), _ptr(hu)))
        return best, cnt, ids, hu

    # ---- device-resident path (bench / multi-GPU) ---------------------------
    def upload(self, batch: QueryBatch):
        h = C.c_void_p()
        self._ck(self._L.ugp_qset_upload(self._h, C.byref(batch.desc), C.byref(h)))
        return h

    def free_qset(self, h) -> None:
        self._L.ugp_qset_destroy(h)

    def place_device(self, qset, d_out_ptr: int, stream: int = 0) -> None:
        """ugp_place_device: stream-ordered on `stream`, like a kernel launch."""
        self._ck(self._L.ugp_place_device(self._h, qset, C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def place_device_overlapped(self, qset, d_out_ptr: int, stream: int = 0) -> None:
        """ugp_place_device_overlapped: up to pipeline_depth() consecutive calls share the device; `stream` gets each call's completion;
        a call is ordered behind what was on `stream` pipeline_depth() - 1 calls ago (cycle through that many output buffers)."""
        self._ck(self._L.ugp_place_device_overlapped(self._h, qset, C.c_void_p(d_out_ptr), C.c_void_p(stream)))

    def pipeline_depth(self) -> int:
        """ugp_pipeline_depth: overlapped calls kept on the device at a time = output buffers to cycle through."""
        return int(self._L.ugp_pipeline_depth(self._h))

    def timing(self) -> Dict:
        out = _lib.ugp_timing()
        self._ck(self._L.ugp_get_timing(self._h, C.byref(out)))
        return {k: getattr(out, k) for k, _ in out._fields_}

    def timing_sum(self) -> Dict:
        """Durations summed over every call since the previous timing_sum() (waits for calls in flight); 'calls' = how many."""
        out = _lib.ugp_timing()
        n = C.c_uint32()
        self._ck(self._L.ugp_get_timing_sum(self._h, C.byref(out), C.byref(n)))
        d = {k: getattr(out, k) for k, _ in out._fields_}
        d["calls"] = int(n.value)
        return d


class MultiPlacer:
    """The same tree on several devices of one node (ugp_mat_create_multi: flattened once, uploaded n times).
    `place` shards a batch into contiguous blocks, one per device, each placed by its own host thread (ctypes
    releases the GIL for the duration of the call); the gather is a host-memory write.  This is the in-process
    form of the multi-GPU path (the reference simply loops over samples, usher_common.cpp:310); usher_amd.dist is
    the one-process-per-GPU form with an RCCL all-gather."""

    def __init__(self, arrays: Dict, devices: Sequence[int]):
        L = _lib.lib()
        self._t = _TreeArrays(arrays)
        self.n_nodes = self._t.n
        self.devices = list(devices)
        n = len(self.devices)
        devs = (C.c_int * n)(*self.devices)
        hs = (C.c_void_p * n)()
        _check(L.ugp_mat_create_multi(C.byref(self._t.desc), devs, n, hs))
        self._hs = [C.c_void_p(h) for h in hs]

    def close(self) -> None:
        for h in getattr(self, "_hs", []):
            if h.value:
                _lib.lib().ugp_mat_destroy(h)
        self._hs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def place(self, batch: QueryBatch) -> np.ndarray:
        import threading
        from .dist import shard_bounds
        n = len(self._hs)
        out = np.zeros(len(batch), dtype=RESULT_DTYPE)
        errs: List[Optional[BaseException]] = [None] * n

        def work(d: int):
            try:
                lo, hi = shard_bounds(len(batch), n, d)
                if hi > lo:
                    part = batch.slice(lo, hi)
                    res = np.zeros(hi - lo, dtype=RESULT_DTYPE)
                    _check(_lib.lib().ugp_place_batch(self._hs[d], C.byref(part.desc), _ptr(res)))
                    out[lo:hi] = res
            except BaseException as e:   # noqa: BLE001 -- re-raised on the calling thread
                errs[d] = e

        ts = [threading.Thread(target=work, args=(d,)) for d in range(n)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        return out
