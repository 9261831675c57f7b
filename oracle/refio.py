"""oracle/refio.py -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Pure-python restatement of the reference's data model and file readers, used
by tests/ to drive the C oracle (oracle/ugp_oracle.c) and to cross-check the
product's C++ host code.  Small inputs only (python loops).

Restated (paths relative to /root/reference):
  * nucleotide codec           src/mutation_annotated_tree.cpp:19-139
  * newick reader              src/mutation_annotated_tree.cpp:383-508
  * newick writer              src/mutation_annotated_tree.cpp:215-346
  * Node::add_mutation         src/mutation_annotated_tree.cpp:720-752
  * parsimony.proto reader     src/mutation_annotated_tree.cpp:522-612 + parsimony.proto
  * read_vcf (existing MAT)    src/mutation_annotated_tree.cpp:2180-2277
  * BFS / DFS expansion        src/mutation_annotated_tree.cpp:1225-1273
"""
from __future__ import annotations

import gzip
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

# --------------------------------------------------------------- nucleotides


def get_nuc_id(c: str) -> int:
    """mutation_annotated_tree.cpp:19-74.  Note 'V' falls through to N (:65-70)
    and only a/c/g/t/n are accepted in lower case."""
    table = {
        "a": 1, "A": 1, "c": 2, "C": 2, "g": 4, "G": 4, "t": 8, "T": 8,
        "R": 0b101, "Y": 0b1010, "S": 0b110, "W": 0b1001, "K": 0b1100, "M": 0b11,
        "B": 0b1110, "D": 0b1101, "H": 0b1011,
    }
    return table.get(c, 0b1111)


_NUC_CHARS = {1: "A", 2: "C", 3: "M", 4: "G", 5: "R", 6: "S", 7: "V", 8: "T",
              9: "W", 10: "Y", 11: "H", 12: "K", 13: "D", 14: "B"}


def get_nuc(nuc_id: int) -> str:
    """mutation_annotated_tree.cpp:88-139."""
    return _NUC_CHARS.get(nuc_id, "N")


# ---------------------------------------------------------------- data model


@dataclass
class Mutation:
    position: int
    ref_nuc: int
    par_nuc: int
    mut_nuc: int
    is_missing: bool = False
    chrom: str = ""

    def is_masked(self) -> bool:
        return self.position < 0

    def get_string(self) -> str:  # mutation_annotated_tree.hpp:79-85
        if self.is_masked():
            return "MASKED"
        return get_nuc(self.par_nuc) + str(self.position) + get_nuc(self.mut_nuc)

    def copy(self) -> "Mutation":
        return Mutation(self.position, self.ref_nuc, self.par_nuc, self.mut_nuc, self.is_missing, self.chrom)


class Node:
    def __init__(self, identifier: str, parent: Optional["Node"], branch_length: float = -1.0):
        self.identifier = identifier
        self.parent = parent
        self.children: List[Node] = []
        self.mutations: List[Mutation] = []
        self.branch_length = branch_length
        self.clade_annotations: List[str] = []
        self.level = 1 if parent is None else parent.level + 1

    def is_leaf(self) -> bool:
        return len(self.children) == 0

    def is_root(self) -> bool:
        return self.parent is None

    def add_mutation(self, mut: Mutation) -> None:
        """mutation_annotated_tree.cpp:720-752."""
        it = 0
        while it < len(self.mutations) and self.mutations[it].position < mut.position:
            it += 1
        if it < len(self.mutations) and self.mutations[it].position == mut.position:
            cur = self.mutations[it]
            if cur.par_nuc != mut.mut_nuc:
                cur.mut_nuc = mut.mut_nuc
            else:
                if cur.mut_nuc != mut.par_nuc:
                    raise ValueError("add_mutation: consecutive mutations at same position disagree")
                p = cur.position
                self.mutations = [m for m in self.mutations if m.position != p]
        else:
            self.mutations.insert(it, mut)


class Tree:
    def __init__(self) -> None:
        self.root: Optional[Node] = None
        self.all_nodes: Dict[str, Node] = {}
        self.curr_internal_node = 0
        self.condensed_nodes: Dict[str, List[str]] = {}
        self.condensed_leaves: set = set()

    def new_internal_node_id(self) -> str:
        self.curr_internal_node += 1
        return "node_" + str(self.curr_internal_node)

    def create_node(self, identifier: str, parent: Optional[Node], branch_length: float = -1.0) -> Node:
        if parent is None:
            n = Node(identifier, None, branch_length)
            self.all_nodes = {identifier: n}
            self.root = n
            return n
        if identifier in self.all_nodes:
            raise ValueError("%s already in the tree" % identifier)
        n = Node(identifier, parent, branch_length)
        n.clade_annotations = [""] * (len(self.root.clade_annotations) if self.root else 0)
        self.all_nodes[identifier] = n
        parent.children.append(n)
        return n

    def get_node(self, nid: str) -> Optional[Node]:
        return self.all_nodes.get(nid)

    def breadth_first_expansion(self) -> List[Node]:
        out: List[Node] = []
        if self.root is None:
            return out
        queue = [self.root]
        head = 0
        while head < len(queue):
            n = queue[head]
            head += 1
            out.append(n)
            queue.extend(n.children)
        return out

    def depth_first_expansion(self, node: Optional[Node] = None) -> List[Node]:
        out: List[Node] = []
        start = node or self.root
        if start is None:
            return out
        stack = [start]
        while stack:
            n = stack.pop()
            out.append(n)
            stack.extend(reversed(n.children))
        return out

    def get_num_leaves(self, node: Optional[Node] = None) -> int:
        node = node or self.root
        return sum(1 for n in self.depth_first_expansion(node) if n.is_leaf())

    def get_parsimony_score(self) -> int:
        return sum(len(n.mutations) for n in self.depth_first_expansion())


# -------------------------------------------------------------------- newick


def _string_split(s: str, delim: str) -> List[str]:
    """mutation_annotated_tree.cpp:383-398 (keeps empty interior words, drops an
    empty trailing one)."""
    words = s.split(delim)
    if words and words[-1] == "":
        words.pop()
    return words


def create_tree_from_newick_string(newick: str) -> Tree:
    """mutation_annotated_tree.cpp:415-508.  Internal ids are node_1.. in the
    order '(' are met (:484)."""
    T = Tree()
    leaves: List[str] = []
    num_open: List[int] = []
    num_close: List[int] = []
    branch_len: Dict[int, List[float]] = {}
    level = 0
    for s in _string_split(newick, ","):
        no = nc = 0
        stop = False
        branch_start = False
        leaf = ""
        branch = ""
        for c in s:
            if c == ":":
                stop = True
                branch = ""
                branch_start = True
            elif c == "(":
                no += 1
                level += 1
            elif c == ")":
                stop = True
                nc += 1
                branch_len.setdefault(level, []).append(float(branch) if branch else -1.0)
                level -= 1
                branch_start = False
            elif not stop:
                leaf += c
                branch_start = False
            elif branch_start:
                if c.isdigit() or c in ".eE-+":
                    branch += c
        leaves.append(leaf)
        num_open.append(no)
        num_close.append(nc)
        branch_len.setdefault(level, []).append(float(branch) if branch else -1.0)
    if level != 0:
        raise ValueError("incorrect Newick format")
    heads = {k: 0 for k in branch_len}

    def pop_len(lv: int) -> float:
        v = branch_len[lv][heads[lv]]
        heads[lv] += 1
        return v

    parent_stack: List[Node] = []
    for leaf, no, nc in zip(leaves, num_open, num_close):
        for _ in range(no):
            nid = T.new_internal_node_id()
            par = parent_stack[-1] if parent_stack else None
            node = T.create_node(nid, par, pop_len(level))
            level += 1
            parent_stack.append(node)
        T.create_node(leaf, parent_stack[-1], pop_len(level))
        for _ in range(nc):
            parent_stack.pop()
            level -= 1
    return T


def create_tree_from_newick(path: str) -> Tree:
    with open(path) as f:
        return create_tree_from_newick_string(f.readline().rstrip("\n"))


def _fmt_len(x: float) -> str:
    # operator<<(float) with default precision 6 ("%g")
    return "%g" % x


def get_newick_string(T: Tree, node: Optional[Node] = None, print_internal: bool = True,
                      print_branch_len: bool = True, uncondense_leaves: bool = False) -> str:
    """mutation_annotated_tree.cpp:215-346 (branch length = #mutations, the
    'band-aid' at :230)."""
    node = node or T.root
    traversal = T.depth_first_expansion(node)
    level_offset = node.level - 1
    curr_level = 0
    prev_open = True
    out: List[str] = []
    node_stack: List[str] = []
    len_stack: List[float] = []

    def leaf_text(n: Node, comma: bool) -> None:
        if uncondense_leaves and n.identifier in T.condensed_nodes:
            if comma:
                out.append(",")
            out.append(",".join(T.condensed_nodes[n.identifier]))
        else:
            if comma:
                out.append(",")
            out.append(n.identifier)

    for n in traversal:
        level = n.level - level_offset
        bl = float(len(n.mutations))
        if curr_level < level:
            if not prev_open:
                out.append(",")
            l = level - 1
            if curr_level > 1:
                l = level - curr_level
            for _ in range(l):
                out.append("(")
                prev_open = True
            if n.is_leaf():
                leaf_text(n, False)
                if print_branch_len and bl >= 0:
                    out.append(":" + _fmt_len(bl))
                prev_open = False
            else:
                node_stack.append(n.identifier)
                len_stack.append(bl)
        elif curr_level > level:
            prev_open = False
            for _ in range(level, curr_level):
                out.append(")")
                if print_internal:
                    out.append(node_stack[-1])
                if print_branch_len and len_stack[-1] >= 0:
                    out.append(":" + _fmt_len(len_stack[-1]))
                node_stack.pop()
                len_stack.pop()
            if n.is_leaf():
                leaf_text(n, True)
                if print_branch_len and bl >= 0:
                    out.append(":" + _fmt_len(bl))
            else:
                node_stack.append(n.identifier)
                len_stack.append(bl)
        else:
            prev_open = False
            if n.is_leaf():
                leaf_text(n, True)
                if print_branch_len and bl >= 0:
                    out.append(":" + _fmt_len(bl))
            else:
                node_stack.append(n.identifier)
                len_stack.append(bl)
        curr_level = level
    while node_stack:
        out.append(")")
        if print_internal:
            out.append(node_stack[-1])
        if print_branch_len and len_stack[-1] >= 0:
            out.append(":" + _fmt_len(len_stack[-1]))
        node_stack.pop()
        len_stack.pop()
    out.append(";")
    return "".join(out)


# ------------------------------------------------------------ parsimony.proto


def _varint(buf: bytes, i: int) -> Tuple[int, int]:
    shift = 0
    val = 0
    while True:
        b = buf[i]
        i += 1
        val |= (b & 0x7F) << shift
        if not b & 0x80:
            return val, i
        shift += 7


def _fields(buf: bytes):
    """Yield (field_number, wire_type, value) for one proto3 message."""
    i = 0
    n = len(buf)
    while i < n:
        key, i = _varint(buf, i)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, i = _varint(buf, i)
            yield fno, wt, v
        elif wt == 2:
            ln, i = _varint(buf, i)
            yield fno, wt, buf[i:i + ln]
            i += ln
        elif wt == 1:
            yield fno, wt, buf[i:i + 8]
            i += 8
        elif wt == 5:
            yield fno, wt, buf[i:i + 4]
            i += 4
        else:
            raise ValueError("unsupported wire type %d" % wt)


def _i32(v: int) -> int:
    """int32 fields are sign-extended to 64 bits on the wire (position = -1 is
    a 10-byte varint)."""
    v &= 0xFFFFFFFFFFFFFFFF
    if v >= 1 << 63:
        v -= 1 << 64
    return v


def parse_parsimony_pb(buf: bytes):
    """parsimony.proto: data{newick=1, node_mutations=2, condensed_nodes=3, metadata=4}."""
    newick = ""
    node_mutations: List[List[dict]] = []
    condensed: List[Tuple[str, List[str]]] = []
    metadata: List[List[str]] = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            newick = v.decode()
        elif fno == 2:
            muts = []
            for f2, _, v2 in _fields(v):
                if f2 != 1:
                    continue
                m = {"position": 0, "ref_nuc": 0, "par_nuc": 0, "mut_nuc": [], "chromosome": ""}
                for f3, w3, v3 in _fields(v2):
                    if f3 == 1:
                        m["position"] = _i32(v3)
                    elif f3 == 2:
                        m["ref_nuc"] = _i32(v3)
                    elif f3 == 3:
                        m["par_nuc"] = _i32(v3)
                    elif f3 == 4:
                        if w3 == 2:  # packed
                            j = 0
                            while j < len(v3):
                                x, j = _varint(v3, j)
                                m["mut_nuc"].append(_i32(x))
                        else:
                            m["mut_nuc"].append(_i32(v3))
                    elif f3 == 5:
                        m["chromosome"] = v3.decode()
                muts.append(m)
            node_mutations.append(muts)
        elif fno == 3:
            name = ""
            leaves = []
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    name = v2.decode()
                elif f2 == 2:
                    leaves.append(v2.decode())
            condensed.append((name, leaves))
        elif fno == 4:
            metadata.append([v2.decode() for f2, _, v2 in _fields(v) if f2 == 1])
    return newick, node_mutations, condensed, metadata


def load_mutation_annotated_tree(path: str) -> Tree:
    """mutation_annotated_tree.cpp:522-612."""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        buf = f.read()
    newick, node_mutations, condensed, metadata = parse_parsimony_pb(buf)
    T = create_tree_from_newick_string(newick)
    dfs = T.depth_first_expansion()
    for idx, node in enumerate(dfs):
        if metadata:
            node.clade_annotations = list(metadata[idx])
        for mut in node_mutations[idx]:
            pos = mut["position"]
            if pos >= 0:
                nuc = 0
                for b in mut["mut_nuc"]:
                    nuc += 1 << b
                m = Mutation(pos, 1 << mut["ref_nuc"], 1 << mut["par_nuc"], nuc, False, mut["chromosome"])
                if m.mut_nuc != m.par_nuc:
                    node.add_mutation(m)
            else:
                node.add_mutation(Mutation(pos, 0, 0, 0, False, mut["chromosome"]))
        node.mutations.sort(key=lambda m: m.position)
    for name, leaves in condensed:
        T.condensed_nodes[name] = list(leaves)
        T.condensed_leaves.update(leaves)
    return T


# ----------------------------------------------------------------------- VCF


@dataclass
class MissingSample:
    name: str
    mutations: List[Mutation] = field(default_factory=list)
    num_ambiguous: int = 0


def _stoi(s: str) -> int:
    """std::stoi: leading digits only ("1:x" -> 1, "0/1" -> 0)."""
    j = 0
    while j < len(s) and s[j].isdigit():
        j += 1
    return int(s[:j])


def read_vcf(T: Tree, path: str) -> List[MissingSample]:
    """mutation_annotated_tree.cpp:2180-2277 (existing-MAT branch)."""
    opener = gzip.open if path.endswith(".gz") else open
    missing: List[MissingSample] = []
    missing_idx: List[int] = []
    header_found = False
    n_ids = 0
    with opener(path, "rt") as f:
        for line in f:
            words = line.split()
            if not header_found and len(words) > 1:
                if words[1] == "POS":
                    for j in range(9, len(words)):
                        n_ids += 1
                        if T.get_node(words[j]) is None and words[j] not in T.condensed_leaves:
                            missing.append(MissingSample(words[j]))
                            missing_idx.append(j)
                    header_found = True
            elif header_found:
                if len(words) != 9 + n_ids:
                    raise ValueError("Incorrect VCF format")
                alleles = _string_split(words[4], ",")
                for k, j in enumerate(missing_idx):
                    pos = _stoi(words[1])
                    ref = get_nuc_id(words[3][0])
                    cell = words[j]
                    if cell[0].isdigit():
                        allele_id = _stoi(cell)
                        if allele_id > 0:
                            allele = alleles[allele_id - 1]
                            nuc = get_nuc_id(allele[0])
                            is_missing = allele[0] == "N" or nuc == 0b1111
                            if allele[0] == "N":
                                nuc = 0b1111
                            missing[k].mutations.append(Mutation(pos, ref, ref, nuc, is_missing, words[0]))
                            if nuc & (nuc - 1):
                                missing[k].num_ambiguous += 1
                    else:
                        missing[k].mutations.append(Mutation(pos, ref, ref, 0b1111, True, words[0]))
                        missing[k].num_ambiguous += 1
    return missing


# --------------------------------------------------------------- flat arrays


def tree_to_bfs_arrays(T: Tree):
    """Flatten to the arrays the C oracle and the product C-ABI take: nodes in
    the reference's BFS order (usher_common.cpp:342)."""
    bfs = T.breadth_first_expansion()
    index = {id(n): j for j, n in enumerate(bfs)}
    n = len(bfs)
    parent = np.full(n, -1, dtype=np.int64)
    mut_off = np.zeros(n + 1, dtype=np.int64)
    pos: List[int] = []
    ref: List[int] = []
    par: List[int] = []
    nuc: List[int] = []
    for j, node in enumerate(bfs):
        if node.parent is not None:
            parent[j] = index[id(node.parent)]
        for m in node.mutations:
            pos.append(m.position)
            ref.append(m.ref_nuc)
            par.append(m.par_nuc)
            nuc.append(m.mut_nuc)
        mut_off[j + 1] = len(pos)
    return {
        "n": n,
        "parent": parent,
        "mut_off": mut_off,
        "mut_pos": np.asarray(pos, dtype=np.int32),
        "mut_ref": np.asarray(ref, dtype=np.int8),
        "mut_par": np.asarray(par, dtype=np.int8),
        "mut_nuc": np.asarray(nuc, dtype=np.int8),
        "names": [nd.identifier for nd in bfs],
    }


def sample_to_arrays(s: MissingSample):
    return {
        "name": s.name,
        "pos": np.asarray([m.position for m in s.mutations], dtype=np.int32),
        "ref": np.asarray([m.ref_nuc for m in s.mutations], dtype=np.int8),
        "nuc": np.asarray([m.mut_nuc for m in s.mutations], dtype=np.int8),
        "is_missing": np.asarray([1 if m.is_missing else 0 for m in s.mutations], dtype=np.int8),
    }
