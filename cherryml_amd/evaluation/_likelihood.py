"""Python face of cb_tree_likelihood (csrc/likelihood.hip.h): held-out log-likelihood of a family
by Felsenstein pruning on the GPU.  Mirrors cherryml/evaluation/_likelihood.py:
`dp_likelihood_computation` (:47-327, same arguments, same return value) and the stage
`compute_log_likelihoods` (:474-600, same directory layout and output files).  Independent sites
evolve under (pi_1, Q_1) scaled by their site rate; contacting pairs (contact map, each site in at
most one pair) under the S*S-state model (pi_2, Q_2) at rate 1, each site of a pair receiving
half of the pair's log-likelihood.  There is no CPU fallback."""
import os
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

from .. import _lib
from ..counting._host import read_contact_map, read_msa, read_site_rates
from ..io import Tree, read_probability_distribution, read_rate_matrix, read_tree


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _stationary_distribution(Q: np.ndarray) -> np.ndarray:
    """pi with pi Q = 0, sum(pi) = 1 (markov_chain/_markov_chain.py:11-19 takes the null eigenvector of Q^T;
    here the same vector from one linear solve, which at 400 states is ~30x cheaper than `eig`)."""
    A = np.array(Q, dtype=np.float64).T
    A[-1, :] = 1.0
    b = np.zeros(A.shape[0])
    b[-1] = 1.0
    return np.linalg.solve(A, b)


def _tree_arrays(tree: Tree):
    order = tree.postorder_traversal()
    index = {v: i for i, v in enumerate(tree.nodes())}
    n = len(index)
    parent = np.full(n, -1, dtype=np.int32)
    length = np.zeros(n)
    for v in order:
        if not tree.is_root(v):
            p, t = tree.parent(v)
            parent[index[v]] = index[p]
            length[index[v]] = t
    return index, np.array([index[v] for v in order], dtype=np.int32), parent, length


def tree_likelihood(tree: Tree, codes_a: np.ndarray, codes_b: Optional[np.ndarray], Q, pi_root, unit_rates,
                    reversible: bool = True, alphabet_size: Optional[int] = None, device: int = 0,
                    profile: Optional[dict] = None) -> np.ndarray:
    """Log-likelihood of every unit (column of `codes_a` [n_nodes, n_units], rows in `tree.nodes()`
    order, -1 = unobserved; `codes_b` for pairs).  `unit_rates[u]` scales Q for unit u."""
    Q, pi_root = _f64(Q), _f64(pi_root).reshape(-1)
    S = Q.shape[0]
    _, order, parent, length = _tree_arrays(tree)
    rates = _f64(unit_rates).reshape(-1)
    cat_rate, unit_cat = np.unique(rates, return_inverse=True)
    unit_cat = np.ascontiguousarray(unit_cat, dtype=np.int32)
    a = np.ascontiguousarray(codes_a, dtype=np.int8)
    b = None if codes_b is None else np.ascontiguousarray(codes_b, dtype=np.int8)
    n_units = a.shape[1]
    if a.shape[0] != parent.size or rates.size != n_units or (b is not None and b.shape != a.shape):
        raise ValueError("codes must be [n_nodes, n_units] and unit_rates [n_units]")
    pi_rev = _f64(_stationary_distribution(Q)) if reversible else None
    ll, ms = np.empty(n_units), np.zeros(2)
    S1 = 0 if b is None else int(alphabet_size if alphabet_size is not None else round(S ** 0.5))
    rc = _lib.load().cb_tree_likelihood(
        device, S, S1, Q.ctypes.data, None if pi_rev is None else pi_rev.ctypes.data, pi_root.ctypes.data,
        parent.size, order.ctypes.data, parent.ctypes.data, length.ctypes.data, cat_rate.size,
        cat_rate.ctypes.data, n_units, unit_cat.ctypes.data, a.ctypes.data, None if b is None else b.ctypes.data,
        ll.ctypes.data, ms.ctypes.data)
    _lib.check(rc, "cb_tree_likelihood")
    if profile is not None:
        profile["kernel_ms"] = profile.get("kernel_ms", 0.0) + float(ms[0])
        key = "prune_ms_pairs" if b is not None else "prune_ms_sites"
        profile[key] = profile.get(key, 0.0) + float(ms[1])
    return ll


def dp_likelihood_computation(tree: Tree, msa: Dict[str, str], contact_map: Optional[np.ndarray],
                              site_rates: List[float], amino_acids: List[str], pi_1: np.ndarray, Q_1: np.ndarray,
                              fact_1=None, reversible_1: bool = True, device_1=None,
                              pi_2: Optional[np.ndarray] = None, Q_2: Optional[np.ndarray] = None, fact_2=None,
                              reversible_2: Optional[bool] = True, device_2=None,
                              output_profiling_path: Optional[str] = None, device: int = 0,
                              profile: Optional[dict] = None) -> Tuple[float, List[float]]:
    """`dp_likelihood_computation` (_likelihood.py:47-327).  `fact_*` / `device_*` are accepted for
    call compatibility and ignored: the spectral factorisation happens on the GPU."""
    st_all = time.time()
    num_sites = len(site_rates)
    if contact_map is not None and Q_2 is not None:
        pairs = [(int(i), int(j)) for i, j in zip(*np.where(np.asarray(contact_map) == 1)) if i < j]
    else:
        pairs = []
    flat = [s for p in pairs for s in p]
    if len(set(flat)) != len(flat):   # :88-95
        raise Exception(f"Each site can only be in contact with one other site. The contacting sites were: {pairs}")
    in_pair = set(flat)
    indep = [i for i in range(num_sites) if i not in in_pair]
    nodes = tree.nodes()
    code = {aa: i for i, aa in enumerate(amino_acids)}
    lut = np.full(256, -1, dtype=np.int8)
    for aa, i in code.items():
        if len(aa) == 1 and ord(aa) < 256:
            lut[ord(aa)] = i
    codes = np.full((len(nodes), num_sites), -1, dtype=np.int8)
    for r, v in enumerate(nodes):
        if tree.is_leaf(v) and v in msa:
            codes[r] = lut[np.frombuffer(msa[v].encode("latin-1"), dtype=np.uint8)]
    lls = [0.0] * num_sites
    profile = {} if profile is None else profile
    if indep:
        ll1 = tree_likelihood(tree, codes[:, indep], None, Q_1, pi_1, [site_rates[i] for i in indep],
                              reversible=bool(reversible_1), device=device, profile=profile)
        for i, x in zip(indep, ll1):
            lls[i] = float(x)
    if pairs:
        ia, ib = [p[0] for p in pairs], [p[1] for p in pairs]
        ll2 = tree_likelihood(tree, codes[:, ia], codes[:, ib], Q_2, pi_2, np.ones(len(pairs)),
                              reversible=bool(reversible_2), alphabet_size=len(amino_acids), device=device,
                              profile=profile)
        for (i, j), x in zip(pairs, ll2):
            lls[i] = float(x) / 2.0
            lls[j] = float(x) / 2.0
    if output_profiling_path is not None:
        with open(output_profiling_path, "w") as f:
            f.write(f"GPU time (expm bank + pruning): {profile.get('kernel_ms', 0.0) / 1e3}\n"
                    f"Total time: {time.time() - st_all}\n")
    return sum(lls), lls


def write_log_likelihood(log_likelihood: Tuple[float, Optional[List[float]]], path: str) -> None:
    """cherryml/io/_log_likelihood.py:5-18."""
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    ll, lls = log_likelihood
    res = f"{ll}\n"
    if lls is not None:
        res += f"{len(lls)} sites\n" + " ".join(map(str, lls))
    with open(path, "w") as f:
        f.write(res)


def compute_log_likelihoods(tree_dir: str, msa_dir: str, site_rates_dir: str, contact_map_dir: Optional[str],
                            families: List[str], amino_acids: List[str], pi_1_path: str, Q_1_path: str,
                            reversible_1: bool, device_1: Optional[str], pi_2_path: Optional[str],
                            Q_2_path: Optional[str], reversible_2: Optional[bool], device_2: Optional[str],
                            output_likelihood_dir: str, num_processes: int = 1, device: int = 0, **_ignored) -> None:
    """The stage `compute_log_likelihoods` (_likelihood.py:474-600): `<family>.txt` (total, then the
    per-site values) and `<family>.profiling` in `output_likelihood_dir`.  Families run one after
    another on one GPU; `num_processes`, `device_1/2` and the CPU threading knobs are accepted and
    ignored."""
    st = time.time()
    os.makedirs(output_likelihood_dir, exist_ok=True)
    pairs_of_amino_acids = [a + b for a in amino_acids for b in amino_acids]
    pi_1_df, Q_1_df = read_probability_distribution(pi_1_path), read_rate_matrix(Q_1_path)
    pi_2_df = read_probability_distribution(pi_2_path) if pi_2_path is not None else None
    Q_2_df = read_rate_matrix(Q_2_path) if Q_2_path is not None else None
    # state validation (:379-408)
    if list(pi_1_df.index) != amino_acids:
        raise Exception(f"pi_1 index is:\n{list(pi_1_df.index)}\nbut expected amino acids:\n{amino_acids}")
    if pi_2_df is not None and list(pi_2_df.index) != pairs_of_amino_acids:
        raise Exception(f"pi_2 index is:\n{list(pi_2_df.index)}\nbut expected pairs of amino acids:\n{pairs_of_amino_acids}")
    if list(Q_1_df.index) != amino_acids or list(Q_1_df.columns) != amino_acids:
        raise Exception(f"Q_1 states are:\n{list(Q_1_df.index)}\n\nbut expected amino acids:\n{amino_acids}")
    if Q_2_df is not None and (list(Q_2_df.index) != pairs_of_amino_acids or list(Q_2_df.columns) != pairs_of_amino_acids):
        raise Exception(f"Q_2 states are:\n{list(Q_2_df.index)}\n\nbut expected pairs of amino acids:\n{pairs_of_amino_acids}")
    for family in families:
        tree = read_tree(os.path.join(tree_dir, family + ".txt"))
        msa = read_msa(os.path.join(msa_dir, family + ".txt"))
        site_rates = list(read_site_rates(os.path.join(site_rates_dir, family + ".txt")))
        contact_map = (read_contact_map(os.path.join(contact_map_dir, family + ".txt"))
                       if contact_map_dir is not None else None)
        res = dp_likelihood_computation(
            tree=tree, msa=msa, contact_map=contact_map, site_rates=site_rates, amino_acids=amino_acids,
            pi_1=pi_1_df.to_numpy(), Q_1=Q_1_df.to_numpy(), reversible_1=reversible_1,
            pi_2=pi_2_df.to_numpy() if pi_2_df is not None else None,
            Q_2=Q_2_df.to_numpy() if Q_2_df is not None else None, reversible_2=reversible_2,
            output_profiling_path=os.path.join(output_likelihood_dir, family + ".profiling"), device=device)
        write_log_likelihood(res, os.path.join(output_likelihood_dir, family + ".txt"))
    with open(os.path.join(output_likelihood_dir, "profiling_0.txt"), "w") as f:
        f.write(f"Total time: {time.time() - st}\n")
