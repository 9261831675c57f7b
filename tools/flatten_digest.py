"""sha256 of every array of the flattened MAT for a fixed set of trees (to check that a change of ugp_flatten.cpp
keeps its output bit-identical): python tools/flatten_digest.py [--big] > before.txt; ...; diff before.txt after.txt"""
import hashlib
import sys
import time

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from usher_amd import synth  # noqa: E402
from usher_amd.placement import FlatTreeView  # noqa: E402
from tests import synth as tsynth  # noqa: E402


def digest(tag, arrays, **kw):
    t0 = time.time()
    v = FlatTreeView(arrays, **kw)
    dt = time.time() - t0
    h = hashlib.sha256()
    parts = []
    for name in sorted(n for n in dir(v) if not n.startswith("_")):
        a = getattr(v, name)
        if isinstance(a, np.ndarray):
            d = hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:12]
        elif isinstance(a, (int, bool)):
            d = str(int(a))
        else:
            continue
        parts.append("%s=%s" % (name, d))
        h.update(("%s=%s;" % (name, d)).encode())
    print(tag, h.hexdigest()[:16], " ".join(parts))
    print("# %s flatten %.3f s" % (tag, dt), file=sys.stderr)


def main():
    for seed in range(6):
        arrays, _ = tsynth.make_case(seed, n_leaves=50 + 97 * seed, n_queries=1, n_sites=40 + 10 * seed)
        for cn in (0, 16, 40):
            digest("small%d/cn%d" % (seed, cn), arrays, chunk_nodes=cn)
    for seed in range(3):
        arrays, _ = tsynth.make_case(40 + seed, n_leaves=300, n_queries=1, n_sites=60, p_masked=0.15, root_muts=3)
        digest("masked%d" % seed, arrays, chunk_nodes=24)
    arrays, _ = tsynth.caterpillar_case(3, depth=700, muts_per_node=2, n_queries=1)
    digest("caterpillar", arrays, chunk_nodes=64)
    arrays, _ = tsynth.polytomy_case(4, n_queries=1)
    digest("polytomy", arrays, chunk_nodes=32)
    single = {k: (np.asarray(v)[:1] if k == "parent" else v) for k, v in arrays.items()}
    for shape in ("random", "sars2"):
        st = synth.SynthTree(300_000, n_sites=4000, seed=5, shape=shape)
        digest("300k/" + shape, st.arrays)
    if "--big" in sys.argv:
        for shape in ("random", "sars2"):
            st = synth.SynthTree(10_000_000, n_sites=25000, seed=1, shape=shape)
            digest("10M/" + shape, st.arrays)


main()
