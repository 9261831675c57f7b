"""The parsimony.proto reader / writer of the host front end against dumps made by the REFERENCE'S OWN generated module.

tools/pin_pb_with_reference.py (build container: imports /root/reference/parsimony_pb2.py, which is never copied) parsed every
file of tests/pb_cases.py with the reference's module, checked that the module re-serialises it byte-identically, and committed
the field dumps under tests/golden/pb_pinned/.  Here, anywhere, without the reference: the files are rebuilt with the product's
writer and must be the pinned bytes; what the product's loader decodes (uh_pb_dump: load_mat, mutation_annotated_tree.cpp:522-612)
must be the reference's dump; load + save (save_mat, :614-681) must give the bytes back; and the breadth-first arrays the
placement tests take (uh_pb_to_arrays) must hold the same mutations node by node."""
import ctypes as C
import hashlib
import json
import os

import numpy as np
import pytest

from tests import pb_cases


@pytest.fixture(scope="module")
def cases(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("pb_cases"))
    return pb_cases.build(d), d


def _pinned(name):
    with open(os.path.join(pb_cases.PINNED, name + ".json")) as f:
        return json.load(f)


def test_every_pinned_case_is_rebuilt_here(cases):
    built, _ = cases
    pinned = sorted(f[:-5] for f in os.listdir(pb_cases.PINNED) if f.endswith(".json"))
    assert sorted(built) == pinned and len(pinned) >= 13
    for name in pinned:
        rec = _pinned(name)
        assert rec["reference_module_reserialises_identically"] and rec["product_load_save_identical"], name
        assert rec["pinned_with"].startswith("/root/reference/parsimony_pb2.py")


def test_writer_produces_the_pinned_bytes(cases):
    """The bytes the reference's module parsed and re-serialised identically are the bytes the writer produces here."""
    built, _ = cases
    for name, path in built.items():
        assert pb_cases.sha256_file(path) == _pinned(name)["sha256_file"], name
    # the -t build of the reference's fixture IS the file the reference recorded: byte for byte where there are no condensed nodes,
    # up to the order of the condensed-node table otherwise (the reference iterates a tbb::concurrent_unordered_map there,
    # mutation_annotated_tree.cpp:650-657: unspecified order)
    assert _pinned("t_branchlen2")["sha256_file"] == _pinned("ref_branchlen2")["sha256_file"]
    a, b = _pinned("t_global"), _pinned("ref_global_assignments")
    assert a["bytes"] == b["bytes"]
    for k in ("newick", "node_mutations", "metadata"):
        assert a["dump"][k] == b["dump"][k], k
    assert sorted(a["dump"]["condensed_nodes"]) == sorted(b["dump"]["condensed_nodes"])


def test_loader_decodes_what_the_reference_module_decodes(cases):
    built, scratch = cases
    for name, path in built.items():
        rec = _pinned(name)
        got = pb_cases.loader_dump(path, scratch)
        assert len(got["node_mutations"]) == rec["nodes"] and sum(len(x) for x in got["node_mutations"]) == rec["mutations"], name
        if "dump" in rec:
            for k in ("newick", "node_mutations", "metadata", "condensed_nodes"):
                assert got[k] == rec["dump"][k], (name, k)
        else:
            assert hashlib.sha256(pb_cases.canonical(got)).hexdigest() == rec["dump_sha256"], name
    # what the annotated file (written by the reference's module) exercises
    d = _pinned("ref_written_annotated")["dump"]
    muts = [m for ml in d["node_mutations"] for m in ml]
    assert any(m[0] < 0 and m[1:4] == [-1, -1, []] for m in muts) and any(len(m[3]) == 2 for m in muts) and any(m[4] == "NC_045512v2" for m in muts)
    assert any(md for md in d["metadata"]) and any(not md for md in d["metadata"])
    masked = _pinned("masked_resaved")["dump"]
    assert sum(m[0] < 0 for ml in masked["node_mutations"] for m in ml) > 50


def test_load_then_save_gives_the_bytes_back(cases):
    built, scratch = cases
    for name, path in built.items():
        out = pb_cases.resave(path, os.path.join(scratch, "again.pb"))
        assert pb_cases.sha256_file(out) == _pinned(name)["sha256_file"], name


@pytest.mark.parametrize("name", ["masked_resaved", "t_branchlen2", "synth_1m_from_arrays"])
def test_breadth_first_arrays_hold_the_pinned_mutations(cases, name):
    """uh_pb_to_arrays (what the placement tests and the oracle take) against the dump, node by node in depth-first preorder;
    files without condensed nodes, so that the arrays are the file's tree as it is."""
    built, scratch = cases
    from tests.host_harness import HOST_LIB
    L = C.CDLL(HOST_LIB)
    L.uh_pb_to_arrays.argtypes = [C.c_char_p] + [C.c_void_p] * 8
    counts = (C.c_uint64 * 2)()
    path = built[name].encode()
    assert L.uh_pb_to_arrays(path, counts, None, None, None, None, None, None, None) == 0
    n, m = int(counts[0]), int(counts[1])
    parent, mut_off = np.zeros(n, np.int64), np.zeros(n + 1, np.int64)
    pos, ref, par, nuc = np.zeros(m, np.int32), np.zeros(m, np.int8), np.zeros(m, np.int8), np.zeros(m, np.int8)
    ptr = lambda a: a.ctypes.data_as(C.c_void_p)
    assert L.uh_pb_to_arrays(path, counts, ptr(parent), ptr(mut_off), ptr(pos), ptr(ref), ptr(par), ptr(nuc), None) == 0
    rec = _pinned(name)
    assert n == rec["nodes"] and m == rec["mutations"] and rec["condensed_nodes"] == 0
    # depth-first preorder from the breadth-first parent array (children of a node are consecutive and in order there)
    first = np.full(n + 1, 0, np.int64)
    np.add.at(first, parent[1:] + 1, 1)
    first = np.cumsum(first)
    kids = np.argsort(parent[1:], kind="stable") + 1
    order = np.zeros(n, np.int64)
    stack, k = [0], 0
    while stack:
        j = stack.pop()
        order[k] = j; k += 1
        stack.extend(kids[first[j]:first[j + 1]][::-1].tolist())
    assert k == n
    idx = {1: 0, 2: 1, 4: 2, 8: 3}
    flat = []
    for j in order:
        for i in range(mut_off[j], mut_off[j + 1]):
            flat.append((int(pos[i]), -1, -1, ()) if pos[i] < 0 else
                        (int(pos[i]), idx[int(ref[i])], idx[int(par[i])], tuple(b for b in range(4) if nuc[i] & (1 << b))))
    if "dump" in rec:
        want = [(mu[0], mu[1], mu[2], tuple(mu[3])) for ml in rec["dump"]["node_mutations"] for mu in ml]
        assert flat == want
        cnt = [int(mut_off[j + 1] - mut_off[j]) for j in order]
        assert cnt == [len(ml) for ml in rec["dump"]["node_mutations"]]
    else:
        got = pb_cases.loader_dump(built[name], scratch)     # (pinned by digest in the test above)
        want = [(mu[0], mu[1], mu[2], tuple(mu[3])) for ml in got["node_mutations"] for mu in ml]
        assert flat == want


def test_oracle_side_parser_reads_the_pinned_fields(cases):
    """oracle/refio.py (the python restatement of the loader that feeds the oracle in the parity tests) against the dumps of the
    reference's module: raw fields + the loader's two content rules."""
    from oracle import refio
    built, _ = cases
    for name, path in built.items():
        rec = _pinned(name)
        if "dump" not in rec:
            continue
        with open(path, "rb") as f:
            nwk, muts, cond, meta = refio.parse_parsimony_pb(f.read())
        assert nwk == rec["dump"]["newick"], name
        got = []
        for ml in muts:
            node = []
            for m in ml:
                if m["position"] < 0:
                    node.append([m["position"], -1, -1, [], m["chromosome"]])
                elif sorted(set(m["mut_nuc"])) != [m["par_nuc"]]:
                    node.append([m["position"], m["ref_nuc"], m["par_nuc"], sorted(set(m["mut_nuc"])), m["chromosome"]])
            got.append(node)
        assert got == rec["dump"]["node_mutations"], name
        assert [[a, list(b)] for a, b in cond] == rec["dump"]["condensed_nodes"], name
        assert [[x for x in md if x] for md in meta] == rec["dump"]["metadata"], name
