"""Fitch-Sankoff assignment of VCF sites onto a tree (MAT construction, `usher -t`):
python mirror of ugp_fitch_sankoff (include/usher_amd.h), which replaces
mapper_body::operator() (src/usher_mapper.cpp:6-161)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def fitch_sankoff(parent, ref, var_off, var_node, var_nuc, device: int = 0):
    """parent: breadth-first parent indices (root = -1 or UINT32_MAX); ref[s]: one-hot REF allele of site s;
    (var_off, var_node, var_nuc): CSR of the non-REF genotype cells of tree nodes per site.
    Returns (site, node, par_nuc, mut_nuc) arrays, ordered by site then node."""
    L = _lib.lib()
    if isinstance(parent, np.ndarray) and parent.dtype == np.uint32 and parent.flags.c_contiguous:
        par = parent            # (already what the C ABI takes: root = UINT32_MAX)
    else:
        par = np.asarray(parent).astype(np.int64)
        par = np.where(par < 0, 0xFFFFFFFF, par).astype(np.uint32)
    ref = np.ascontiguousarray(ref, dtype=np.uint8)
    var_off = np.ascontiguousarray(var_off, dtype=np.uint64)
    var_node = np.ascontiguousarray(var_node, dtype=np.uint32)
    var_nuc = np.ascontiguousarray(var_nuc, dtype=np.uint8)
    if len(var_off) != len(ref) + 1:
        raise ValueError("var_off must have n_sites + 1 entries")
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    sites = _lib.ugp_sites(len(ref), p(ref), p(var_off), p(var_node), p(var_nuc))
    h = C.c_void_p()
    import time
    t0 = time.perf_counter()
    rc = L.ugp_fitch_sankoff(device, len(par), p(par), C.byref(sites), C.byref(h))
    fitch_sankoff.last_call_s = time.perf_counter() - t0   # (the C call alone: what bench_fitch.py reports)
    if rc != 0:
        raise RuntimeError("ugp_fitch_sankoff failed (%d): %s" % (rc, L.ugp_last_error().decode()))
    try:
        n = int(L.ugp_fitch_count(h))
        site, node = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
        mpar, mnuc = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        rc = L.ugp_fitch_get(h, p(site), p(node), p(mpar), p(mnuc))
        if rc != 0:
            raise RuntimeError("ugp_fitch_get failed (%d): %s" % (rc, L.ugp_last_error().decode()))
    finally:
        L.ugp_fitch_destroy(h)
    return site, node, mpar, mnuc


def release_pool(device: int = 0) -> None:
    """ugp_fitch_release: hand the pooled device buffers of fitch_sankoff (up to 4 GiB of row storage) back."""
    _lib.lib().ugp_fitch_release(device)
