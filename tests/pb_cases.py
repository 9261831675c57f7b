"""The parsimony.proto files whose wire format is pinned to the reference's own generated module (test helper).

`build(outdir)` produces, deterministically and on CPU, every kind of `.pb` the host writer (usher_amd/csrc/host/mat.cpp,
save_mat / write_pb_from_arrays; reference: mutation_annotated_tree.cpp:614-681) emits in the test suite:
a `-t` build of the reference's fixtures (Fitch-Sankoff + condensed leaves), the `-l` build of scripts/testBranchLen2,
an add-mode output (new internal nodes, new leaves), a collapsed output tree (`-c`), trees with masked mutations
(`position = -1`: ref_nuc = par_nuc = -1, no mut_nuc, mutation_annotated_tree.cpp:632-634), a 1 M-node synthetic tree written
from arrays -- plus the reference-written fixtures (tests/golden/survey_ref/*.pb) and `ref_written_annotated.pb`, a file
that tools/pin_pb_with_reference.py wrote WITH the reference's module (clade annotations, a chromosome name, a multi-allelic
mut_nuc, a masked mutation).

tools/pin_pb_with_reference.py (build container only: it imports /root/reference/parsimony_pb2.py) parses each file with the
reference's module, asserts a byte-identical re-serialisation, and stores the field dump (or its digest) under
tests/golden/pb_pinned/; tests/test_pb_pinned.py rebuilds the files anywhere, checks the bytes are the pinned ones and compares
what the product's loader decodes (uh_pb_dump) with the reference's dump."""
from __future__ import annotations

import ctypes as C
import gzip
import hashlib
import json
import os
import shutil

import numpy as np

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SURVEY = os.path.join(GOLD, "survey_ref")
FIX = os.path.join(GOLD, "ref_fixtures")
PINNED = os.path.join(GOLD, "pb_pinned")

# cases whose dump is too large to commit: the pinned JSON holds the sha256 of the canonical dump instead
DIGEST_ONLY = {"synth_1m_from_arrays", "ref_big_tree"}


def sha256_file(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 22), b""):
            h.update(blk)
    return h.hexdigest()


def canonical(dump) -> bytes:
    return json.dumps(dump, sort_keys=True, separators=(",", ":"), ensure_ascii=True).encode()


def _host_lib():
    from tests.host_harness import HOST_LIB
    L = C.CDLL(HOST_LIB)
    L.uh_pb_dump.argtypes = [C.c_char_p, C.c_char_p]
    L.uh_pb_resave.argtypes = [C.c_char_p, C.c_char_p]
    P = C.c_void_p
    L.uh_write_pb_arrays.argtypes = [C.c_uint64, P, P, P, P, P, P, C.c_char_p]
    return L


def loader_dump(path, scratch):
    """What the product's load_mat() decoded, in the message's own shape (uh_pb_dump)."""
    out = os.path.join(scratch, "dump.json")
    assert _host_lib().uh_pb_dump(path.encode(), out.encode()) == 0
    with open(out) as f:
        return json.load(f)


def resave(path, out):
    assert _host_lib().uh_pb_resave(path.encode(), out.encode()) == 0
    return out


def write_pb_arrays(arrays, path):
    L = _host_lib()
    n = int(arrays["n"])
    par = np.asarray(arrays["parent"]).astype(np.int64)
    keep = [np.where(par < 0, 0xFFFFFFFF, par).astype(np.uint32), np.asarray(arrays["mut_off"]).astype(np.uint64),
            np.asarray(arrays["mut_pos"]).astype(np.int32), np.asarray(arrays["mut_ref"]).astype(np.uint8),
            np.asarray(arrays["mut_par"]).astype(np.uint8), np.asarray(arrays["mut_nuc"]).astype(np.uint8)]
    keep = [np.ascontiguousarray(a) for a in keep]
    assert L.uh_write_pb_arrays(n, *[a.ctypes.data_as(C.c_void_p) for a in keep], path.encode()) == 0


def build(outdir, big=True):
    """name -> path of every pinned case, (re)built under `outdir`.  Deterministic: same bytes on every machine."""
    from tests import synth
    from tests.host_harness import run_usher
    os.makedirs(outdir, exist_ok=True)
    cases = {}
    # reference-written fixtures (recorded at survey time) and the file written with the reference's module by the pin tool
    cases["ref_global_assignments"] = os.path.join(SURVEY, "global", "global_assignments.pb")
    cases["ref_syn_tree"] = os.path.join(SURVEY, "syn", "tree.pb")
    cases["ref_branchlen2"] = os.path.join(SURVEY, "branchlen2", "tbl2.pb")
    big_pb = os.path.join(outdir, "ref_big_tree.pb")
    with gzip.open(os.path.join(SURVEY, "big", "tree.pb.gz"), "rb") as f, open(big_pb, "wb") as o:
        shutil.copyfileobj(f, o)
    cases["ref_big_tree"] = big_pb
    annotated = os.path.join(PINNED, "ref_written_annotated.pb")
    if os.path.exists(annotated):
        cases["ref_written_annotated"] = annotated
    # -t builds of the reference's fixtures
    vcf = os.path.join(outdir, "global_samples.vcf")
    with gzip.open(os.path.join(FIX, "global_samples.vcf.gz"), "rb") as f, open(vcf, "wb") as o:
        shutil.copyfileobj(f, o)
    d = os.path.join(outdir, "t_global"); os.makedirs(d, exist_ok=True)
    pb = os.path.join(outdir, "t_global.pb")
    assert run_usher(["-t", os.path.join(FIX, "global_phylo.nh"), "-v", vcf, "-o", pb, "-d", d]) == 0
    cases["t_global"] = pb
    d = os.path.join(outdir, "t_branchlen2"); os.makedirs(d, exist_ok=True)
    pb2 = os.path.join(outdir, "t_branchlen2.pb")
    assert run_usher(["-t", os.path.join(FIX, "testBranchLen2.nwk"), "-v", os.path.join(FIX, "testBranchLen2.vcf"), "-o", pb2, "-l", "-d", d]) == 0
    cases["t_branchlen2"] = pb2
    # add mode (default): new leaves, new internal nodes, re-condensed on save (usher_common.cpp:1037-1040)
    d = os.path.join(outdir, "add_global"); os.makedirs(d, exist_ok=True)
    out = os.path.join(outdir, "add_global.pb")
    assert run_usher(["-i", pb, "-v", os.path.join(FIX, "new_samples.vcf"), "-o", out, "-d", d]) == 0
    cases["add_global"] = out
    # the same with the input and output trees collapsed (-c / -C)
    d = os.path.join(outdir, "add_global_collapsed"); os.makedirs(d, exist_ok=True)
    out = os.path.join(outdir, "add_global_collapsed.pb")
    assert run_usher(["-i", pb, "-v", os.path.join(FIX, "new_samples.vcf"), "-c", "-o", out, "-d", d]) == 0
    cases["add_global_collapsed"] = out
    # add mode on the larger recorded tree
    d = os.path.join(outdir, "add_syn"); os.makedirs(d, exist_ok=True)
    out = os.path.join(outdir, "add_syn.pb")
    assert run_usher(["-i", cases["ref_syn_tree"], "-v", os.path.join(SURVEY, "syn", "query.vcf"), "-o", out, "-d", d]) == 0
    cases["add_syn"] = out
    # masked mutations (position = -1) and root mutations, written from arrays
    arrays, _ = synth.make_case(4242, n_leaves=400, n_queries=1, n_sites=200, p_masked=0.2, root_muts=3)
    out = os.path.join(outdir, "masked_from_arrays.pb")
    write_pb_arrays(arrays, out)
    cases["masked_from_arrays"] = out
    # ... and such a tree through load_mat() + save_mat()
    out2 = os.path.join(outdir, "masked_resaved.pb")
    resave(out, out2)
    cases["masked_resaved"] = out2
    if big:
        from usher_amd.synth import SynthTree
        t = SynthTree(1_000_000, n_sites=5000, seed=3)
        out = os.path.join(outdir, "synth_1m_from_arrays.pb")
        write_pb_arrays(t.arrays, out)
        t.close()
        cases["synth_1m_from_arrays"] = out
    return cases
