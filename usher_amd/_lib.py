"""ctypes binding of libusher_amd.so (include/usher_amd.h).  No fallback: if the
shared library is missing this raises, it never substitutes a CPU path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libusher_amd.so")
EXP_LIB_PATH = os.path.join(_HERE, "libusher_amd_exp.so")   # the same sources with -DUGP_EXPERIMENTS (diagnostics, experiments)
_lib = None
_lib_exp = None


class ugp_tree_desc(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("parent", C.c_void_p), ("mut_off", C.c_void_p), ("mut_pos", C.c_void_p),
                ("mut_ref", C.c_void_p), ("mut_par", C.c_void_p), ("mut_nuc", C.c_void_p)]


class ugp_queries(C.Structure):
    _fields_ = [("n_queries", C.c_uint64), ("ent_off", C.c_void_p), ("pos", C.c_void_p), ("ref", C.c_void_p),
                ("nuc", C.c_void_p), ("is_missing", C.c_void_p)]


class ugp_result(C.Structure):
    _fields_ = [("best_set_difference", C.c_int32), ("num_best", C.c_uint32), ("best_j", C.c_uint32),
                ("best_has_unique", C.c_uint32)]


class ugp_info(C.Structure):
    _fields_ = [("n_nodes", C.c_uint64), ("n_muts", C.c_uint64), ("n_sites", C.c_uint64), ("stream_bytes", C.c_uint64),
                ("algo_tree_bytes", C.c_uint64), ("algo_tile_bytes", C.c_uint64), ("n_chunks", C.c_uint32),
                ("max_slots", C.c_uint32), ("max_position", C.c_uint32), ("device", C.c_uint32)]


class ugp_timing(C.Structure):
    _fields_ = [("table_ms", C.c_float), ("place_ms", C.c_float), ("merge_ms", C.c_float),
                ("place_launches", C.c_uint32), ("n_tiles", C.c_uint32), ("n_groups", C.c_uint32), ("packed_path", C.c_uint32),
                ("reserved", C.c_uint32), ("words_total", C.c_uint64), ("words_skipped", C.c_uint64), ("coarse_ms", C.c_float), ("bound3", C.c_uint32)]


class ugp_place_opts(C.Structure):
    _fields_ = [("order", C.c_uint32), ("node_mask", C.c_void_p), ("skip_node", C.c_void_p), ("distance", C.c_void_p), ("scores", C.c_void_p)]


class ugp_touched(C.Structure):
    _fields_ = [("n", C.c_uint64), ("flat_j", C.c_void_p), ("flags", C.c_void_p), ("n_path", C.c_void_p), ("ent_off", C.c_void_p), ("pos", C.c_void_p),
                ("allele", C.c_void_p), ("prev", C.c_void_p), ("ref", C.c_void_p)]


class ugp_sites(C.Structure):
    _fields_ = [("n_sites", C.c_uint64), ("ref", C.c_void_p), ("var_off", C.c_void_p), ("var_node", C.c_void_p),
                ("var_nuc", C.c_void_p)]


# every symbol include/usher_amd.h declares: name -> (restype, argtypes)
P = C.c_void_p
SYMBOLS = {
    "ugp_mat_create": (C.c_int, [C.POINTER(ugp_tree_desc), C.c_int, C.POINTER(P)]),
    "ugp_mat_create_multi": (C.c_int, [C.POINTER(ugp_tree_desc), C.POINTER(C.c_int), C.c_int, C.POINTER(P)]),
    "ugp_flat_save": (C.c_int, [C.POINTER(ugp_tree_desc), C.c_char_p]),
    "ugp_mat_create_from_flat": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(P)]),
    "ugp_mat_destroy": (None, [P]),
    "ugp_mat_info": (C.c_int, [P, C.POINTER(ugp_info)]),
    "ugp_place_batch": (C.c_int, [P, C.POINTER(ugp_queries), P]),
    "ugp_scores_per_node": (C.c_int, [P, C.POINTER(ugp_queries), P]),
    "ugp_tied_nodes": (C.c_int, [P, C.POINTER(ugp_queries), C.c_uint32, P, P, P]),
    "ugp_place_batch_ex": (C.c_int, [P, C.POINTER(ugp_queries), C.POINTER(ugp_place_opts), P]),
    "ugp_tied_nodes_ex": (C.c_int, [P, C.POINTER(ugp_queries), C.POINTER(ugp_place_opts), C.c_uint32, P, P, P]),
    "ugp_ex_prepare": (C.c_int, [P, C.POINTER(ugp_place_opts), C.POINTER(P)]),
    "ugp_ex_destroy": (None, [P]),
    "ugp_place_batch_prepared": (C.c_int, [P, C.POINTER(ugp_queries), P, P, P, P]),
    "ugp_device_warmup": (C.c_int, [C.c_int]),
    "ugp_node_order": (C.c_int, [P, C.c_uint32, P]),
    "ugp_subtree_mask": (C.c_int, [P, C.c_uint32, C.c_uint32, C.c_uint32, P]),
    "ugp_qset_upload": (C.c_int, [P, C.POINTER(ugp_queries), C.POINTER(P)]),
    "ugp_qset_destroy": (None, [P]),
    "ugp_qset_size": (C.c_uint64, [P]),
    "ugp_place_device": (C.c_int, [P, P, P, P]),
    "ugp_place_device_overlapped": (C.c_int, [P, P, P, P]),
    "ugp_pipeline_depth": (C.c_int, [P]),
    "ugp_mat_update": (C.c_int, [P, C.POINTER(ugp_touched), P, C.c_uint64, C.POINTER(C.c_uint32)]),
    "ugp_touched_open": (C.c_int, [P, C.POINTER(ugp_queries)]),
    "ugp_touched_score": (C.c_int, [P, C.c_uint32, C.c_uint64]),
    "ugp_touched_rescore": (C.c_int, [P, C.c_uint64]),
    "ugp_touched_fetch": (C.c_int, [P, C.c_uint64, C.c_uint64, C.c_uint32, P, P, P, P]),
    "ugp_mat_reload_knobs": (C.c_int, [P]),
    "ugp_has_experiments": (C.c_int, []),
    "ugp_place_batch_async": (C.c_int, [P, C.POINTER(ugp_queries), P, C.POINTER(P)]),
    "ugp_job_wait": (C.c_int, [P]),
    "ugp_get_timing": (C.c_int, [P, C.POINTER(ugp_timing)]),
    "ugp_get_timing_sum": (C.c_int, [P, C.POINTER(ugp_timing), C.POINTER(C.c_uint32)]),
    "ugp_debug_bound3_tables": (C.c_int, [P, C.c_uint32, P, P, C.c_uint64, C.POINTER(C.c_uint64)]),
    "ugp_last_error": (C.c_char_p, []),
    "ugp_fitch_sankoff": (C.c_int, [C.c_int, C.c_uint64, P, C.POINTER(ugp_sites), C.POINTER(P)]),
    "ugp_fitch_count": (C.c_uint64, [P]),
    "ugp_fitch_get": (C.c_int, [P, P, P, P, P]),
    "ugp_fitch_destroy": (None, [P]),
    "ugp_fitch_release": (None, [C.c_int]),
    "ugp_mat_create_chunked": (C.c_int, [C.POINTER(ugp_tree_desc), C.c_int, C.c_uint32, C.POINTER(P)]),
    "ugp_flat_create": (C.c_int, [C.POINTER(ugp_tree_desc), C.c_uint32, C.POINTER(P)]),
    "ugp_flat_destroy": (None, [P]),
    "ugp_flat_get": (C.c_int, [P, C.c_int, C.POINTER(P), C.POINTER(C.c_uint64)]),
}


def build_library(force: bool = False) -> str:
    """Compile libusher_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-s", "-C", csrc]
    if force:
        cmd.append("-B")
    subprocess.check_call(cmd)
    return LIB_PATH


def _load(path):
    if not os.path.exists(path):
        raise RuntimeError(
            "%s is not built (%s); run `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C usher_amd/csrc`. There is no CPU fallback." % (os.path.basename(path), path))
    L = C.CDLL(path)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    return L


def lib(experiments: bool = False):
    """The release library, or (experiments=True) the build with -DUGP_EXPERIMENTS; each is loaded once per process."""
    global _lib, _lib_exp
    if experiments:
        if _lib_exp is None:
            _lib_exp = _load(EXP_LIB_PATH)
        return _lib_exp
    if _lib is None:
        # (tuning hook: USHER_AMD_LIB names another build of the same sources -- e.g. one compiled with -DUGP_GRP=8 -- for an A/B on one box)
        _lib = _load(os.environ.get("USHER_AMD_LIB") or LIB_PATH)
    return _lib
