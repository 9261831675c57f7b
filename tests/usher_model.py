"""Test helper: a python restatement of the per-sample block of the reference's driver (usher_common.cpp:310-792) for the
default add mode and for --multiple-placements, on oracle/refio's tree model with the oracle doing every search.  Used
to check the C++ front end's multi-tree path (host/driver.cpp run_multi) where no recorded reference output exists.
Files produced: placement_stats.tsv, final-tree[-N].nh, mutation-paths[-N].txt.  Clade annotations are not modelled
(the test trees carry none).  Test infrastructure only."""
from __future__ import annotations

from typing import Dict, List

from oracle import capi, refio


def copy_tree(T: refio.Tree) -> refio.Tree:
    """get_tree_copy, mutation_annotated_tree.cpp:1493-1549."""
    C = refio.create_tree_from_newick_string(refio.get_newick_string(T, T.root, True, True))
    d1, d2 = T.depth_first_expansion(), C.depth_first_expansion()
    assert len(d1) == len(d2)
    for a, b in zip(d1, d2):
        b.clade_annotations = list(a.clade_annotations)
        for m in a.mutations:
            b.add_mutation(m.copy())
    C.condensed_nodes = {k: list(v) for k, v in T.condensed_nodes.items()}
    C.condensed_leaves = set(T.condensed_leaves)
    return C


def _move_plain(T: refio.Tree, src: refio.Node, dst: refio.Node) -> None:
    src.parent.children.remove(src)
    src.parent = dst
    src.branch_length = -1.0
    dst.children.append(src)
    stack = [src]
    while stack:
        n = stack.pop()
        n.level = n.parent.level + 1
        stack.extend(n.children)


def insert(T: refio.Tree, best: refio.Node, as_sibling: bool, name: str, excess: List[refio.Mutation]) -> None:
    """usher_common.cpp:652-765."""
    same = lambda a, b: a.position == b.position and a.mut_nuc == b.mut_nuc
    if as_sibling:
        nid = T.new_internal_node_id()
        mid = T.create_node(nid, best.parent)
        leaf = T.create_node(name, mid)
        _move_plain(T, best, mid)
        branch = [m.copy() for m in best.mutations]
        best.mutations = []
        l1 = [m for m in branch if m.is_masked() or not any(same(m, e) for e in excess)]
        common = [e for e in excess if not e.is_masked() and any(same(e, m) for m in branch)]
        l2 = [e for e in excess if e.is_masked() or not any(same(e, m) for m in branch)]
        for m in common:
            mid.add_mutation(m.copy())
        for m in l1:
            best.add_mutation(m.copy())
        for m in l2:
            leaf.add_mutation(m.copy())
    else:
        leaf = T.create_node(name, best)
        for e in excess:
            if e.is_masked() or not any(same(e, m) for m in best.mutations):
                leaf.add_mutation(e.copy())


def _paths(T: refio.Tree, names: List[str]) -> str:
    """get_sample_mutation_paths, mutation_annotated_tree.cpp:1991-2050."""
    out = ""
    for name in names:
        n = T.get_node(name)
        if n is None:
            continue
        parts = []
        a = n
        while a is not None:
            if a.mutations:
                parts.append(a.identifier + ":" + ",".join(m.get_string() for m in a.mutations) + " ")
            a = a.parent
        out += name + "\t" + "".join(reversed(parts)) + "\n"
    return out


def run(T0: refio.Tree, missing: List[refio.MissingSample], max_trees: int = 1, max_uncertainty: int = 10 ** 6,
        max_parsimony: int = 10 ** 6) -> Dict[str, str]:
    trees = [T0]
    stats = ""
    for ms in missing:
        num_trees = len(trees)
        for t_idx in range(num_trees):
            T = trees[t_idx]
            if T.get_node(ms.name) is not None:
                continue
            arrays = refio.tree_to_bfs_arrays(T)
            bfs = T.breadth_first_expansion()
            ot = capi.OracleTree(arrays)
            s = refio.sample_to_arrays(ms)
            r = ot.place(s)
            best, num_best = r["best"], r["num_best"]
            stats += "%s\t%d\t%d\t" % (ms.name, best, num_best)
            curr = copy_tree(T) if (max_trees > 1 and num_best > 1 and num_trees < max_trees) else None
            if num_best <= max_uncertainty and best <= max_parsimony:
                if num_best > 1 and len(trees) <= max_trees and num_best + len(trees) > max_trees:
                    num_best = 1 + max_trees - len(trees)
                nhu = ot.node_has_unique_prefix(s, num_best)
                for k in range(num_best):
                    if max_trees > 1 and num_best > 1:
                        if k > 0:
                            T = copy_tree(curr)
                            trees.append(T)
                            bfs = T.breadth_first_expansion()
                        best_j, hu = int(r["ties"][k]), bool(nhu[k])
                    else:
                        best_j, hu = r["best_j"], r["has_unique"]
                    node = bfs[best_j]
                    nv = ot.node_vecs(s, best_j)
                    excess = [refio.Mutation(p, rf, pa, mu) for (p, rf, pa, mu) in nv["excess"]]
                    if T.get_node(ms.name) is None:
                        insert(T, node, node.is_leaf() or hu, ms.name, excess)
                    if nv["imputed"]:
                        stats += ";".join("%d:%s" % (p, refio.get_nuc(mu)) for (p, rf, pa, mu) in nv["imputed"])
                    if max_trees == 1:
                        break
            stats += "\n"
    out = {"placement_stats.tsv": stats}
    names = [m.name for m in missing]
    for t, T in enumerate(trees):
        suffix = "-%d" % (t + 1) if len(trees) > 1 else ""
        out["final-tree%s.nh" % suffix] = refio.get_newick_string(T, T.root, True, True)
        out["mutation-paths%s.txt" % suffix] = _paths(T, names)
    return out


# ---------------------------------------------------------------------------------------------------------------------
# Tree editing used by --collapse-tree / --collapse-output-tree and the subtree writers (restated from
# mutation_annotated_tree.cpp: remove_node :960-1049, move_node :1135-1223, collapse_tree :1384-1424,
# condense_leaves :1287-1332, get_leaves :818-840, get_subtree :1575-1681, rotate_for_display :1426-1453,
# get_random_single_subtree :1693-1786, get_random_sample_subtrees :1788-1989).
# ---------------------------------------------------------------------------------------------------------------------
def _relevel(n):
    stack = [n]
    while stack:
        x = stack.pop()
        x.level = x.parent.level + 1 if x.parent else 1
        stack.extend(x.children)


def remove_node(T, source, move_level):
    par = source.parent
    if par is not None:
        par.children.remove(source)
        if not par.children:
            if par is not T.root:
                remove_node(T, par, move_level)
        elif move_level and len(par.children) == 1:
            child = par.children[0]
            if par.parent is not None:
                for k in range(min(len(par.clade_annotations), len(child.clade_annotations))):
                    if child.clade_annotations[k] == "":
                        child.clade_annotations[k] = par.clade_annotations[k]
                child.parent = par.parent
                child.branch_length += par.branch_length
                own = child.mutations
                child.mutations = []
                for m in par.mutations:
                    child.add_mutation(m.copy())
                for m in own:
                    child.add_mutation(m.copy())
                par.parent.children.append(child)
                par.parent.children.remove(par)
                _relevel(child)
                del T.all_nodes[par.identifier]
    queue = [source]
    for n in queue:
        queue.extend(n.children)
    for n in queue:
        del T.all_nodes[n.identifier]


def _same_muts(a, b):
    return len(a) == len(b) and all((x.position, x.is_missing, x.chrom, x.par_nuc, x.mut_nuc) == (y.position, y.is_missing, y.chrom, y.par_nuc, y.mut_nuc)
                                    for x, y in zip(a, b))


def move_node(T, source, dest, move_level=True):
    cur = source.parent
    assert cur is not dest

    def link(p, c):
        c.parent = p
        c.branch_length = -1.0
        p.children.append(c)

    def unlink(p, c):
        p.children.remove(c)
        if not p.children:
            remove_node(T, p, move_level)

    existing = next((c for c in dest.children if _same_muts(c.mutations, source.mutations)), None)
    if existing is cur or not source.mutations:
        existing = None
    relevel = []
    if existing is None:
        link(dest, source); unlink(cur, source); relevel.append(source)
    elif existing.is_leaf():
        if source.is_leaf():
            mid = T.create_node(T.new_internal_node_id(), dest, -1.0)
            for m in source.mutations:
                mid.add_mutation(m.copy())
            source.mutations = []
            existing.mutations = []
            link(mid, source); link(mid, existing); unlink(dest, existing); unlink(cur, source); relevel.append(mid)
        else:
            existing.mutations = []
            link(source, existing); link(dest, source); unlink(dest, existing); unlink(cur, source); relevel.append(source)
    elif source.is_leaf():
        source.mutations = []
        link(existing, source); unlink(cur, source); relevel.append(source)
    else:
        for k in list(source.children):
            move_node(T, k, existing, move_level)
    for n in relevel:
        _relevel(n)


def collapse_tree(T):
    def rec(node):
        if not node.children:
            return
        for c in list(node.children):
            rec(c)
        parent = node.parent
        if parent is None:
            return
        if not node.mutations:
            for c in list(node.children):
                move_node(T, c, parent, False)
        elif len(node.children) == 1:
            child = node.children[0]
            for m in child.mutations:
                node.add_mutation(m.copy())
            child.mutations = [m.copy() for m in node.mutations]
            move_node(T, child, parent, False)
    import sys
    sys.setrecursionlimit(100000)
    rec(T.root)


def condense_leaves(T):
    if T.condensed_nodes:
        uncondense_leaves(T)
    for lid in [n.identifier for n in T.breadth_first_expansion() if n.is_leaf()]:
        l1 = T.get_node(lid)
        if l1 is None or l1.mutations or l1.parent is None:
            continue
        poly = [l2 for l2 in l1.parent.children if l2.is_leaf() and not l2.mutations]
        if len(poly) > 1:
            name = "node_%d_condensed_%d_leaves" % (1 + len(T.condensed_nodes), len(poly))
            T.create_node(name, l1.parent, l1.branch_length)
            T.condensed_nodes[name] = [p.identifier for p in poly]
            for p in poly:
                p.parent.children.remove(p)
                del T.all_nodes[p.identifier]


def uncondense_leaves(T):
    for name, ids in T.condensed_nodes.items():
        n = T.get_node(name)
        if n is None:
            continue
        par = n.parent if n.parent is not None else n
        k = len(ids)
        n_ann = len(T.root.clade_annotations)

        def rename(node, new):
            del T.all_nodes[node.identifier]
            node.identifier = new
            T.all_nodes[new] = node
        if k > 1 and n.mutations:
            rename(n, T.new_internal_node_id())
            for s in ids:
                c = refio.Node(s, n, -1.0)
                c.clade_annotations = [""] * n_ann
                T.all_nodes[s] = c
                n.children.append(c)
        elif k > 1:
            rename(n, ids[0])
            for s in ids[1:]:
                c = refio.Node(s, par, n.branch_length)
                c.clade_annotations = [""] * n_ann
                T.all_nodes[s] = c
                par.children.append(c)
        elif k == 1:
            rename(n, ids[0])
    T.condensed_nodes = {}
    T.condensed_leaves = set()


def get_leaves(T, node=None):
    out, queue = [], [node or T.root]
    for n in queue:
        if not n.children:
            out.append(n)
        queue.extend(n.children)
    return out


def _rsearch(n, include_self):
    out = [n] if include_self else []
    a = n.parent
    while a is not None:
        out.append(a)
        a = a.parent
    return out


def _is_ancestor(a, n):
    return a in _rsearch(n, False)


def get_subtree(T, samples):
    keep = set()
    anc = []
    for s in samples:
        n = T.get_node(s)
        keep.add(n)
        anc.append(set(_rsearch(n, True)))
    for i in range(len(samples)):
        for j in range(i + 1, len(samples)):
            for a in _rsearch(T.get_node(samples[i]), True):
                if a in anc[j]:
                    keep.add(a)
                    break
    S = refio.Tree()
    last = []
    for n in T.depth_first_expansion():
        if n not in keep:
            continue
        while last and not _is_ancestor(last[-1], n):
            last.pop()
        sp = last[-1] if last else None
        nn = S.create_node(n.identifier, S.get_node(sp.identifier) if sp else None, -1.0)
        path = list(reversed(_rsearch(n, True)))
        if sp is not None:
            path = path[path.index(sp) + 1:]
        for c in path:
            for m in c.mutations:
                nn.add_mutation(m.copy())
        last.append(n)
    S.curr_internal_node = T.curr_internal_node
    return S


def rotate_for_display(T):
    order = T.depth_first_expansion()
    nd = {}
    for n in reversed(order):
        nd[id(n)] = 1 + sum(nd[id(c)] for c in n.children)
    for n in order:
        n.children.sort(key=lambda c: -nd[id(c)])     # (stable)


def _subtree_files(T, S, stem):
    rotate_for_display(S)
    out = {stem + ".nh": refio.get_newick_string(S, S.root, True, True)}
    out[stem + "-mutations.txt"] = "".join("%s: %s\n" % (n.identifier, ",".join(m.get_string() for m in n.mutations)) for n in S.depth_first_expansion())
    exp = "".join("%s: %s\n" % (l.identifier, "".join(c + " " for c in T.condensed_nodes[l.identifier])) for l in get_leaves(S) if l.identifier in T.condensed_nodes)
    if exp:
        out[stem + "-expanded.txt"] = exp
    return out


class _LibcRand:
    def __init__(self):
        import ctypes
        self.libc = ctypes.CDLL("libc.so.6")

    def srand(self, s):
        self.libc.srand(s)

    def rand(self):
        return self.libc.rand()


def single_subtree(T, names, subtree_size, rng):
    keep, order = set(), []
    for s in names:
        n = T.get_node(s)
        if n is not None and n not in keep:
            keep.add(n); order.append(n)
    leaves = get_leaves(T)
    for _ in range(len(leaves)):
        l = leaves[rng.rand() % len(leaves)]
        if l not in keep:
            keep.add(l); order.append(l)
        if len(keep) >= subtree_size + len(names):
            break
    return _subtree_files(T, get_subtree(T, [n.identifier for n in order]), "single-subtree")


def sample_subtrees(T, names, subtree_size, rng):
    """Only exact while subtree_size < 5 (no randomly chosen fifth, so std::shuffle is never reached)."""
    assert subtree_size // 5 == 0
    rng.srand(0)
    leaves = get_leaves(T)
    seen = set()
    for _ in range(len(leaves)):
        seen.add(leaves[rng.rand() % len(leaves)])
        if len(seen) >= subtree_size:
            break
    displayed = [T.get_node(s) is None for s in names]
    out = {}
    num = 0
    for i, s in enumerate(names):
        if displayed[i]:
            continue
        last_anc = T.get_node(s)
        keep = []
        for anc in _rsearch(last_anc, True):
            nl = len(get_leaves(T, anc))
            if nl < subtree_size:
                last_anc = anc
                continue
            if nl > subtree_size:
                keep.extend(l.identifier for l in get_leaves(T, last_anc))
                dist = []
                for l in get_leaves(T, anc):
                    if _is_ancestor(last_anc, l):
                        continue
                    d = 0
                    for a in _rsearch(l, True):
                        if a is anc:
                            break
                        d += len(a.mutations)
                    dist.append((d, l))
                dist.sort(key=lambda t: t[0])       # (std::sort on equal keys: the test trees avoid ties mattering -- see the test)
                for d, l in dist:
                    if len(keep) >= subtree_size:
                        break
                    keep.append(l.identifier)
            else:
                for l in get_leaves(T, anc):
                    if len(keep) == subtree_size:
                        break
                    keep.append(l.identifier)
            S = get_subtree(T, keep)
            for j in range(i + 1, len(names)):
                if not displayed[j] and S.get_node(names[j]) is not None:
                    displayed[j] = True
            num += 1
            out.update(_subtree_files(T, S, "subtree-%d" % num))
            break
    return out
