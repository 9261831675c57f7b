"""usher_amd -- MI355X-native parsimony placement engine (UShER-compatible).

The compute path is libusher_amd.so (hand-written HIP kernels for gfx950 behind
the C ABI of include/usher_amd.h).  This package is the thin python mirror of
that ABI; it has no CPU fallback and raises if the library is missing.
"""
from .placement import FlatTreeView, MultiPlacer, Placer, QueryBatch, UgpError, build_library  # noqa: F401

__all__ = ["Placer", "MultiPlacer", "QueryBatch", "FlatTreeView", "UgpError", "build_library"]
