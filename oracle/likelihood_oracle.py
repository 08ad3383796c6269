"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement (numpy) of the reference's held-out log-likelihood computation,
cherryml/evaluation/_likelihood.py:47-327 (`dp_likelihood_computation`): Felsenstein pruning in log
space over a rooted tree, independent sites under (pi_1, Q_1) scaled by their site rate, contacting
pairs of sites (contact map, each site in at most one pair) under the 400-state model (pi_2, Q_2) at
rate 1, each site of a pair getting half of the pair's log-likelihood.  Unknown symbols are
all-ones observations; in a pair, a known partner restricts the 20 compatible pair states.

Pinned by tests/test_oracle_golden.py::test_likelihood_* on vectors produced by the reference itself
(tests/golden/make_golden_likelihood.py), including its published FastTree-verified values."""
from typing import Dict, List, Optional, Tuple

import numpy as np
from scipy.linalg import expm


def _obs_single(c: str, code: Dict[str, int], S: int) -> np.ndarray:
    v = np.zeros(S)
    if c in code:
        v[code[c]] = 1.0
    else:
        v[:] = 1.0
    return v


def _obs_pair(c1: str, c2: str, code: Dict[str, int], S: int) -> np.ndarray:
    a = _obs_single(c1, code, S)
    b = _obs_single(c2, code, S)
    return np.outer(a, b).reshape(-1)      # state index = i1 * S + i2


def log_likelihood(tree, msa: Dict[str, str], contact_map: Optional[np.ndarray], site_rates: List[float],
                   amino_acids: List[str], pi_1: np.ndarray, Q_1: np.ndarray, pi_2: Optional[np.ndarray] = None,
                   Q_2: Optional[np.ndarray] = None) -> Tuple[float, List[float]]:
    S = len(amino_acids)
    code = {a: i for i, a in enumerate(amino_acids)}
    L = len(site_rates)
    if contact_map is not None and Q_2 is not None:
        pairs = [(int(i), int(j)) for i, j in zip(*np.where(np.asarray(contact_map) == 1)) if i < j]
    else:
        pairs = []
    flat = [s for p in pairs for s in p]
    if len(set(flat)) != len(flat):
        raise Exception(f"Each site can only be in contact with one other site. The contacting sites were: {pairs}")
    indep = [i for i in range(L) if i not in flat]
    order = tree.postorder_traversal()
    P1, P2 = {}, {}
    for v in order:
        if tree.is_root(v):
            continue
        _, t = tree.parent(v)
        P1[v] = {r: expm(t * r * Q_1) for r in set(site_rates[i] for i in indep)}
        if pairs:
            P2[v] = expm(t * Q_2)

    def prune(units, obs_of, P_of, pi):
        res = []
        for u in units:
            dp = {}
            for v in order:
                dp[v] = np.zeros(len(pi))
                if tree.is_leaf(v):
                    continue
                for child, _ in tree.children(v):
                    d = dp[child]
                    m = d.max()
                    arg = P_of(child, u) @ (np.exp(d - m) * obs_of(child, u))
                    arg = np.where(arg < 0, 0.0, arg)
                    with np.errstate(divide="ignore"):
                        dp[v] = dp[v] + np.log(arg) + m
            d = dp[tree.root()]
            m = d.max()
            arg = float(pi @ (np.exp(d - m) * obs_of(tree.root(), u)))
            res.append(np.log(max(arg, 0.0)) + m)
        return res

    def obs1(v, i):
        return _obs_single(msa[v][i], code, S) if (tree.is_leaf(v) and v in msa) else np.ones(S)

    def obs2(v, p):
        return _obs_pair(msa[v][p[0]], msa[v][p[1]], code, S) if (tree.is_leaf(v) and v in msa) else np.ones(S * S)

    lls = [0.0] * L
    for i, ll in zip(indep, prune(indep, obs1, lambda v, i: P1[v][site_rates[i]], np.asarray(pi_1).reshape(-1))):
        lls[i] = ll
    if pairs:
        for (i, j), ll in zip(pairs, prune(pairs, obs2, lambda v, p: P2[v], np.asarray(pi_2).reshape(-1))):
            lls[i] = ll / 2.0
            lls[j] = ll / 2.0
    return float(sum(lls)), lls
