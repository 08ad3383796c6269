"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement (plain Python / numpy loops) of the reference's SiteRM count and pseudocount
assembly, cherryml/_siterm/_site_specific_rate_matrix.py:

  cherry_transitions      :87-139   greedy post-order pairing of leaves ("cherry++")
  edge_transitions        :393-405
  raw_count_matrices      :189-261  per-site, per-bucket transition counts
  count_prior_matrices    :325-355  diag(pi0) expm(t_b Q0) by the reversible factorisation
                                    (markov_chain/_markov_chain.py:56-155)
  mixed_count_matrices    :503-567  pseudocounts l1[l,b] * prior[b_adjusted(l,b)], lambda mix
  compactify              :577-602  non-empty buckets first, per site; init = Q0 * rate_l

Pinned by tests/test_oracle_golden.py::test_siterm_assembly_* against vectors produced by running
the reference itself (tests/golden/make_golden_siterm_assembly.py).  Trees are anything with the
reference's Tree interface (root / children / is_leaf / edges / leaves / nodes)."""
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


def quantization_idx(branch_length: float, grid: np.ndarray) -> Optional[int]:
    """cherryml/utils.py:35-56 (same comparisons, same divisions)."""
    if branch_length < grid[0] or branch_length > grid[-1]:
        return None
    idx = int(np.searchsorted(grid, branch_length))
    if idx == 0:
        return 0
    left, right = grid[idx - 1], grid[idx]
    rel_left = branch_length / left - 1.0
    rel_right = right / branch_length - 1.0
    return idx - 1 if rel_left < rel_right else idx


def cherry_transitions(tree, msa: Dict[str, str]) -> List[Tuple[str, str, float]]:
    """:87-139 -- iterative post-order; children in insertion order; leftovers bubble up."""
    cherries: List[Tuple[str, str, float]] = []
    result: Dict[str, Tuple[Optional[str], Optional[float]]] = {}
    stack = [(tree.root(), False)]
    while stack:
        node, done = stack.pop()
        if tree.is_leaf(node):
            result[node] = (node, 0.0)
            continue
        if not done:
            stack.append((node, True))
            for child, _ in reversed(tree.children(node)):
                stack.append((child, False))
            continue
        leaves, dists = [], []
        for child, bl in tree.children(node):
            leaf, d = result[child]
            if leaf is not None:
                leaves.append(leaf)
                dists.append(d + bl)
        i = 0
        while i + 1 <= len(leaves) - 1:
            cherries.append((msa[leaves[i]], msa[leaves[i + 1]], dists[i] + dists[i + 1]))
            i += 2
        result[node] = (None, None) if len(leaves) % 2 == 0 else (leaves[-1], dists[-1])
    return cherries


def edge_transitions(tree, msa: Dict[str, str]) -> List[Tuple[str, str, float]]:
    return [(msa[u], msa[v], t) for (u, v, t) in tree.edges()]


def raw_count_matrices(transitions, grid_sorted: Sequence[float], alphabet: List[str],
                       include_reverse_transitions: bool = True) -> np.ndarray:
    grid = np.asarray(grid_sorted, dtype=np.float64)
    code = {c: i for i, c in enumerate(alphabet)}
    L, B, S = len(transitions[0][0]), len(grid), len(alphabet)
    raw = np.zeros((L, B, S, S))
    for x, y, t in transitions:
        b = quantization_idx(t, grid)
        if b is None:
            continue
        for l in range(L):
            xi, yi = code.get(x[l], -1), code.get(y[l], -1)
            if xi >= 0 and yi >= 0:
                raw[l, b, xi, yi] += 1.0
    if include_reverse_transitions:
        raw = (raw + raw.transpose(0, 1, 3, 2)) / 2.0
    return raw


def stationary_distribution(Q: np.ndarray) -> np.ndarray:
    w, v = np.linalg.eig(Q.transpose())
    idx = int(np.argmin(np.abs(w.real)))
    p = v[:, idx].real
    return p / p.sum()


def count_prior_matrices(Q0: np.ndarray, grid_sorted: Sequence[float]) -> np.ndarray:
    pi = stationary_distribution(Q0)
    P1, P2 = np.diag(np.sqrt(pi)), np.diag(np.sqrt(1.0 / pi))
    D, U = np.linalg.eigh(P1 @ Q0 @ P2)
    out = np.zeros((len(grid_sorted), len(pi), len(pi)))
    for b, t in enumerate(grid_sorted):
        expm = (P2 @ U) @ (np.diag(np.exp(t * D)) @ (U.T @ P1))
        out[b] = pi[:, None] * expm
        if abs(float(out[b].sum()) - 1.0) > 1e-6:
            raise ValueError("count_prior_probability_matrices[b, :, :] does not add up to 1!")
    return out


def mixed_count_matrices(raw: np.ndarray, prior: np.ndarray, site_rates: Sequence[float],
                         grid_sorted: Sequence[float], lam: float) -> np.ndarray:
    grid = np.asarray(grid_sorted, dtype=np.float64)
    L, B = raw.shape[:2]
    pseudo = np.zeros_like(raw)
    l1 = raw.sum(axis=(2, 3))
    for l in range(L):
        for b in range(B):
            if l1[l, b] <= 0:
                continue
            tt = grid[b] * site_rates[l]
            ba = quantization_idx(tt, grid)
            if ba is None:
                ba = B - 1 if tt > grid[-1] else 0
            pseudo[l, b] = l1[l, b] * prior[ba]
    return raw * (1.0 - lam) + pseudo * lam


def compactify(counts: np.ndarray, grid_sorted: Sequence[float], Q0: np.ndarray, site_rates: Sequence[float]):
    L, B, S, _ = counts.shape
    sums = counts.sum(axis=(2, 3))
    keep = [[b for b in range(B) if float(sums[l, b]) > 0] for l in range(L)]
    Be = max(len(k) for k in keep)
    cc = np.zeros((L, Be, S, S))
    tt = np.ones((L, Be))
    for l in range(L):
        for k, b in enumerate(keep[l]):
            cc[l, k] = counts[l, b]
            tt[l, k] = grid_sorted[b]
    init = np.stack([Q0 * site_rates[l] for l in range(L)])
    return cc, tt, init
