#!/usr/bin/env python3
"""Times of the front end's two loaders (parsimony.proto, VCF) on the synthetic workload written as files; no GPU needed.
    python tools/time_load.py [--nodes 10000000] [--samples 20000] [--keep DIR]"""
import argparse
import ctypes as C
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def host_lib():
    L = C.CDLL(os.path.join(ROOT, "usher_amd", "libusher_host.so"))
    P = C.c_void_p
    L.uh_write_pb_arrays.argtypes = [C.c_uint64, P, P, P, P, P, P, C.c_char_p]
    L.uh_write_vcf_csr.argtypes = [C.c_uint64, P, P, P, P, P, C.c_char_p, C.c_char_p]
    L.uh_time_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double)]
    return L


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def write_workload(L, st, q, n_samples, pb, vcf, prefix="NEW"):
    a = st.arrays
    par = np.where(np.asarray(a["parent"]).astype(np.int64) < 0, 0xFFFFFFFF, np.asarray(a["parent"]).astype(np.int64)).astype(np.uint32)
    keep = [np.ascontiguousarray(par), np.ascontiguousarray(a["mut_off"], dtype=np.uint64), np.ascontiguousarray(a["mut_pos"], dtype=np.int32),
            np.ascontiguousarray(a["mut_ref"]).astype(np.uint8), np.ascontiguousarray(a["mut_par"]).astype(np.uint8), np.ascontiguousarray(a["mut_nuc"]).astype(np.uint8)]
    if pb:
        assert L.uh_write_pb_arrays(int(a["n"]), *[ptr(k) for k in keep], pb.encode()) == 0
    if vcf:
        e1 = int(q["ent_off"][n_samples])
        kq = [np.ascontiguousarray(q["ent_off"][:n_samples + 1], dtype=np.uint64), np.ascontiguousarray(q["pos"][:e1], dtype=np.int32), np.ascontiguousarray(q["ref"][:e1], dtype=np.uint8),
              np.ascontiguousarray(q["nuc"][:e1], dtype=np.uint8), np.ascontiguousarray(q["is_missing"][:e1], dtype=np.uint8)]
        assert L.uh_write_vcf_csr(n_samples, *[ptr(k) for k in kq], prefix.encode(), vcf.encode()) == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=10_000_000)
    ap.add_argument("--samples", type=int, default=20000)
    ap.add_argument("--keep", default="")
    a = ap.parse_args()
    from usher_amd import synth
    L = host_lib()
    d = a.keep or tempfile.mkdtemp(prefix="load_")
    os.makedirs(d, exist_ok=True)
    pb, vcf = os.path.join(d, "base.pb"), os.path.join(d, "new.vcf")
    t0 = time.time()
    if not (os.path.exists(pb) and os.path.exists(vcf)):
        st = synth.SynthTree(a.nodes, n_sites=25000 if a.nodes > 200000 else 1500, seed=1)
        q = st.queries(a.samples, seed=5, max_subst=3, n_lo=0, n_hi=3, iupac_hi=0)
        write_workload(L, st, q, a.samples, pb, vcf)
    prep = time.time() - t0
    out = (C.c_double * 6)()
    assert L.uh_time_load(pb.encode(), vcf.encode(), out) == 0
    print(json.dumps({"nodes": int(out[2]), "samples": int(out[3]), "sample_rows": int(out[4]), "tree_mutations": int(out[5]), "load_mat_s": round(out[0], 3), "read_vcf_s": round(out[1], 3),
                      "pb_bytes": os.path.getsize(pb), "vcf_bytes": os.path.getsize(vcf), "prep_s": round(prep, 1), "threads": os.environ.get("USHER_AMD_THREADS", "default")}))
    if not a.keep:
        os.remove(pb); os.remove(vcf); os.rmdir(d)


if __name__ == "__main__":
    main()
