"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement (pure Python / numpy, small inputs only) of the reference's counting
stage, the producer of the count-matrix bank:

  * cherryml/counting/_count_transitions.py:37-198      single-site transitions
  * cherryml/counting/_count_co_transitions.py:38-224   pair-of-sites co-transitions
  * cherryml/utils.py:35-56                             quantization_idx
  * cherryml/io/_tree.py:193-266, _msa.py:51-77, _site_rates.py:5-26,
    _contact_map.py:6-31                                the four text formats

Pinned by tests/test_counting_cpu.py against the reference tests' own data
(tests/golden/counting/tiny*: inputs + expected result.txt) and against outputs of the
reference's Python counters on a synthetic family set (tests/golden/counting_synth.npz,
made by tests/golden/make_golden_counting.py).
"""
import os
from typing import Dict, List, Optional, Tuple

import numpy as np


# ----------------------------------------------------------------------- formats
def read_tree(path: str):
    """-> (nodes in file order, children: {u: [(v, length), ...] in file order})"""
    lines = open(path).read().strip().split("\n")
    n = int(lines[0].split(" ")[0])
    nodes = lines[1:1 + n]
    m = int(lines[1 + n].split(" ")[0])
    children = {u: [] for u in nodes}
    has_parent = set()
    for ln in lines[2 + n:2 + n + m]:
        u, v, length = ln.split(" ")
        children[u].append((v, float(length)))
        has_parent.add(v)
    roots = [u for u in nodes if u not in has_parent]
    assert len(roots) == 1
    return nodes, children, roots[0]


def read_msa(path: str) -> Dict[str, str]:
    lines = open(path).read().strip().split("\n")
    return {lines[2 * i][1:]: lines[2 * i + 1] for i in range(len(lines) // 2)}


def read_site_rates(path: str) -> List[float]:
    lines = open(path).read().strip().split("\n")
    return list(map(float, lines[1].split(" ")))


def read_contact_map(path: str) -> np.ndarray:
    lines = open(path).read().strip().split("\n")
    n = int(lines[0].split(" ")[0])
    return np.array([[int(ch) for ch in lines[i + 1]] for i in range(n)], dtype=int)


# ----------------------------------------------------------------- quantisation
def quantization_idx(branch_length: float, grid: np.ndarray) -> Optional[int]:
    """utils.py:35-56: nearest grid point in RELATIVE error, ties to the right,
    None outside [grid[0], grid[-1]]."""
    if branch_length < grid[0] or branch_length > grid[-1]:
        return None
    k = int(np.searchsorted(grid, branch_length))
    if k == 0:
        return 0
    left, right = grid[k - 1], grid[k]
    return k - 1 if (branch_length / left - 1) < (right / branch_length - 1) else k


# -------------------------------------------------------------------- pairings
def transition_pairs(nodes, children, root, mode: str) -> List[Tuple[str, str, float, float]]:
    """(a, b, len_a, len_b) for every counted pair.  edge: (parent, child, length, 0);
    cherry: two-leaf-children nodes; cherry++: post-order greedy pairing of the unmatched
    leaves under each node, in child order (_count_transitions.py:65-125)."""
    out = []
    if mode == "edge":
        for u in nodes:
            for v, ln in children[u]:
                out.append((u, v, ln, 0.0))
    elif mode == "cherry":
        for u in nodes:
            ch = children[u]
            if len(ch) == 2 and all(len(children[c]) == 0 for c, _ in ch):
                out.append((ch[0][0], ch[1][0], ch[0][1], ch[1][1]))
    elif mode == "cherry++":
        def dfs(u):
            if len(children[u]) == 0:
                return (u, 0.0)
            un, ds = [], []
            for v, ln in children[u]:
                leaf, d = dfs(v)
                if leaf is not None:
                    un.append(leaf)
                    ds.append(d + ln)
            i = 0
            while i + 1 <= len(un) - 1:
                out.append((un[i], un[i + 1], ds[i], ds[i + 1]))
                i += 2
            return (None, None) if len(un) % 2 == 0 else (un[-1], ds[-1])
        import sys
        sys.setrecursionlimit(max(10000, sys.getrecursionlimit()))
        dfs(root)
    else:
        raise ValueError(mode)
    return out


# --------------------------------------------------------------------- counters
def count_transitions(tree_dir, msa_dir, site_rates_dir, families, amino_acids,
                      quantization_points, edge_or_cherry) -> np.ndarray:
    grid = np.array(sorted(float(q) for q in quantization_points))
    idx = {a: i for i, a in enumerate(amino_acids)}
    S = len(amino_acids)
    C = np.zeros((len(grid), S, S))
    if edge_or_cherry.startswith("cherry++"):
        edge_or_cherry = "cherry++"
    for fam in families:
        nodes, children, root = read_tree(os.path.join(tree_dir, fam + ".txt"))
        msa = read_msa(os.path.join(msa_dir, fam + ".txt"))
        rates = read_site_rates(os.path.join(site_rates_dir, fam + ".txt"))
        for a, b, la, lb in transition_pairs(nodes, children, root, edge_or_cherry):
            sa, sb = msa[a], msa[b]
            total = la if edge_or_cherry == "edge" else la + lb
            for k in range(len(sa)):
                q = quantization_idx(total * rates[k], grid)
                if q is None or sa[k] not in idx or sb[k] not in idx:
                    continue
                x, y = idx[sa[k]], idx[sb[k]]
                if edge_or_cherry == "edge":
                    C[q, x, y] += 1
                else:
                    C[q, x, y] += 0.5
                    C[q, y, x] += 0.5
    return C


def co_count_pair(Cq, code_a, code_b, contacts, S, symmetric) -> None:
    """The per-pair inner loop of the reference's Python co-transition counter (_count_co_transitions.py:108-140;
    C++ :359-381) on integer state codes (-1 = not in the alphabet): for every contact (i, j) of a pair whose four
    residues are known, states s = a_i S + a_j, e = b_i S + b_j and their site-swapped versions s', e' get
    0.5 at (s, e), (s', e') for an edge, 0.25 at (s, e), (s', e'), (e, s), (e', s') for a cherry."""
    for i, j in contacts:
        ai, aj, bi, bj = code_a[i], code_a[j], code_b[i], code_b[j]
        if ai < 0 or aj < 0 or bi < 0 or bj < 0:
            continue
        s1, s1r, s2, s2r = ai * S + aj, aj * S + ai, bi * S + bj, bj * S + bi
        if not symmetric:
            Cq[s1, s2] += 0.5
            Cq[s1r, s2r] += 0.5
        else:
            for s, e in ((s1, s2), (s1r, s2r), (s2, s1), (s2r, s1r)):
                Cq[s, e] += 0.25


def count_co_transitions(tree_dir, msa_dir, contact_map_dir, families, amino_acids,
                         quantization_points, edge_or_cherry,
                         minimum_distance_for_nontrivial_contact) -> np.ndarray:
    grid = np.array(sorted(float(q) for q in quantization_points))
    idx = {a: i for i, a in enumerate(amino_acids)}
    S = len(amino_acids)
    C = np.zeros((len(grid), S * S, S * S))
    if edge_or_cherry.startswith("cherry++"):
        edge_or_cherry = "cherry++"
    for fam in families:
        nodes, children, root = read_tree(os.path.join(tree_dir, fam + ".txt"))
        msa = read_msa(os.path.join(msa_dir, fam + ".txt"))
        cm = read_contact_map(os.path.join(contact_map_dir, fam + ".txt"))
        contacts = [(i, j) for i, j in zip(*np.where(cm == 1))
                    if abs(i - j) >= minimum_distance_for_nontrivial_contact and i < j]
        for a, b, la, lb in transition_pairs(nodes, children, root, edge_or_cherry):
            total = la if edge_or_cherry == "edge" else la + lb
            q = quantization_idx(total, grid)
            if q is None:
                continue
            co_count_pair(C[q], [idx.get(c, -1) for c in msa[a]], [idx.get(c, -1) for c in msa[b]], contacts, S,
                          edge_or_cherry != "edge")
    return C
