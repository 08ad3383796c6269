"""Host side of counting: the four text formats and the choice of counted pairs.

Formats (reference cherryml/io/_tree.py:214-266, _msa.py:51-77,
_site_rates.py:5-26, _contact_map.py:6-31).  Pairing rules (reference
_count_transitions.py:65-125,128-175): "edge" = every (parent, child); "cherry" = nodes whose
two children are both leaves; "cherry++" = post-order greedy pairing of the unmatched leaves
under every node, in child (= file) order, path lengths accumulated leaf-to-node.
"""
from typing import Dict, List, Tuple

import numpy as np


# ------------------------------------------------------------------------ formats
_strtof = None


def parse_float32(text: str) -> float:
    """The decimal string rounded to float32 ONCE, as `std::stof` does (libc `strtof`), widened to float: what the
    reference's C++ counters keep of a branch length (counting/_count_transitions.cpp:247, _count_co_transitions.cpp:245).
    (`np.float32(float(text))` would round twice -- to double, then to float.)"""
    global _strtof
    if _strtof is None:
        import ctypes
        fn = ctypes.CDLL(None).strtof
        fn.restype = ctypes.c_float
        fn.argtypes = [ctypes.c_char_p, ctypes.c_void_p]
        _strtof = fn
    float(text)   # the same ValueError as the double path for a malformed field
    return float(_strtof(text.encode("ascii"), None))


def read_tree_arrays(path: str, length_float32: bool = False):
    """-> (names[n], children[n] = list of (child index, length) in file order, root index)
    length_float32: branch lengths through float32 (`cpp_compat` of the counting stages)."""
    with open(path, "r") as f:
        lines = f.read().strip().split("\n")
    try:
        n, word = lines[0].split(" ")
        if word != "nodes":
            raise ValueError
        n = int(n)
    except Exception:
        raise Exception(f"Tree file: {path} should start with '[num_nodes] nodes'. "
                        f"It started with: '{lines[0]}'")
    names = lines[1:1 + n]
    index = {v: i for i, v in enumerate(names)}
    try:
        m, word = lines[n + 1].split(" ")
        if word != "edges":
            raise ValueError
        m = int(m)
    except Exception:
        raise Exception(f"Tree file: {path} should have line '[num_edges] edges' at position "
                        f"{n + 1}, but it had line: '{lines[n + 1]}'")
    if len(lines) != n + m + 2:
        raise Exception(f"Tree file: {path} should have {m} edges, but it has "
                        f"{len(lines) - n - 2} edges instead.")
    children: List[List[Tuple[int, float]]] = [[] for _ in range(n)]
    has_parent = np.zeros(n, dtype=bool)
    for i in range(n + 2, n + 2 + m):
        try:
            u, v, length = lines[i].split(" ")
            length = parse_float32(length) if length_float32 else float(length)
        except Exception:
            raise Exception(f"Tree file: {path} should have line '[u] [v] [length]' at position "
                            f"{i}, but it had line: '{lines[i]}'")
        if u not in index or v not in index:
            raise Exception(f"In Tree file {path}: {u} and {v} should be nodes in the tree")
        if has_parent[index[v]]:
            raise Exception(f"Node {v} already has a parent - graph is not a tree.")
        has_parent[index[v]] = True
        children[index[u]].append((index[v], length))
    roots = np.flatnonzero(~has_parent)
    if len(roots) != 1:
        raise Exception(f"Tree should have one root, but found: {[names[r] for r in roots]}")
    return names, children, int(roots[0])


def read_msa(path: str) -> Dict[str, str]:
    with open(path, "r") as f:
        lines = f.read().strip().split("\n")
    if len(lines) % 2 != 0:
        raise Exception(f"The MSA at {path} should have an even number of lines")
    msa = {}
    for i in range(0, len(lines), 2):
        if not lines[i].startswith(">"):
            raise Exception(f"MSA at {path}: at line {i} expected '>[seq_name]' but found {lines[i]}")
        msa[lines[i][1:]] = lines[i + 1]
    return msa


def read_site_rates(path: str) -> np.ndarray:
    lines = open(path).read().strip().split("\n")
    try:
        num_sites, word = lines[0].split(" ")
        if word != "sites":
            raise ValueError
        num_sites = int(num_sites)
    except Exception:
        raise Exception(f"Site rates file: {path} should start with line '[num_sites] sites', "
                        f"but started with: {lines[0]} instead.")
    res = np.array([float(x) for x in lines[1].split(" ")], dtype=np.float64)
    if len(res) != num_sites:
        raise Exception(f"Site rates file: {path} was supposed to have {num_sites} sites, "
                        f"but it has {len(res)}")
    return res


def read_contact_map(path: str) -> np.ndarray:
    lines = open(path).read().strip().split("\n")
    try:
        num_sites, word = lines[0].split(" ")
        if word != "sites":
            raise ValueError
        num_sites = int(num_sites)
    except Exception:
        raise Exception("Contact map file should start with line '[num_sites] sites', "
                        f"but started with: {lines[0]} instead.")
    if len(lines) != num_sites + 1:
        raise Exception(f"Contact Map at: {path} should have {num_sites} rows, "
                        f"but has {len(lines) - 1}")
    raw = np.frombuffer("".join(lines[1:]).encode("ascii"), dtype=np.uint8)
    if raw.size != num_sites * num_sites:
        raise Exception(f"Contact Map at: {path} is not {num_sites} x {num_sites}")
    return (raw.reshape(num_sites, num_sites) - ord("0")).astype(np.int32)


# ----------------------------------------------------------------------- encoding
def encode_msa(msa: Dict[str, str], names: List[str], amino_acids: List[str]) -> np.ndarray:
    """int8 codes [len(names), L]: state index, -1 for any other symbol."""
    lut = np.full(256, -1, dtype=np.int8)
    for i, a in enumerate(amino_acids):
        if len(a) != 1:
            raise ValueError("single-character states only")
        lut[ord(a)] = i
    L = len(msa[names[0]]) if names else 0
    out = np.empty((len(names), L), dtype=np.int8)
    for r, nm in enumerate(names):
        seq = msa[nm]
        if len(seq) != L:
            raise Exception(f"sequence {nm} has length {len(seq)}, expected {L}")
        out[r] = lut[np.frombuffer(seq.encode("latin-1"), dtype=np.uint8)]
    return out


# ------------------------------------------------------------------------ pairing
def build_pairs(children, root: int, mode: str) -> List[Tuple[int, int, float, float]]:
    """[(node a, node b, len_a, len_b)] in the reference's visiting order."""
    n = len(children)
    out: List[Tuple[int, int, float, float]] = []
    if mode == "edge":
        for u in range(n):
            for v, ln in children[u]:
                out.append((u, v, ln, 0.0))
        return out
    if mode == "cherry":
        for u in range(n):
            ch = children[u]
            if len(ch) == 2 and not children[ch[0][0]] and not children[ch[1][0]]:
                out.append((ch[0][0], ch[1][0], ch[0][1], ch[1][1]))
        return out
    if mode != "cherry++":
        raise ValueError(f"unknown edge_or_cherry: {mode}")
    # iterative post-order: result[u] = (unmatched leaf, distance) or None
    result = [None] * n
    stack = [(root, 0)]
    while stack:
        u, ci = stack.pop()
        ch = children[u]
        if not ch:
            result[u] = (u, 0.0)
            continue
        if ci < len(ch):
            stack.append((u, ci + 1))
            stack.append((ch[ci][0], 0))
            continue
        unmatched, dist = [], []
        for v, ln in ch:
            if result[v] is not None:
                unmatched.append(result[v][0])
                dist.append(result[v][1] + ln)
        for i in range(0, len(unmatched) - 1, 2):
            out.append((unmatched[i], unmatched[i + 1], dist[i], dist[i + 1]))
        result[u] = (unmatched[-1], dist[-1]) if len(unmatched) % 2 == 1 else None
    return out
