"""SiteRM count / pseudocount assembly and the per-family estimator driver: the interface of the
reference's cherryml/_siterm/_site_specific_rate_matrix.py (file:line cited per function) with the
per-transition / per-site loops on the GPU (`cb_siterm_assemble`, csrc/counting.hip.h) and the
optimiser on the GPU (`cb_train_siterm`).  The count tensor [L,B,S,S] is produced on the device
and handed to the bank without a host round trip; `cb_create` drops the empty (site, bucket)
matrices, which is the reference's "compactification" (:577-602).

Host side (like the reference: tiny, closed form): the cherry++ pairing walk and the prior
matrices diag(pi0) expm(t_b Q0)."""
import time
from typing import Dict, List, Optional, Tuple

import numpy as np

from .. import _lib
from .._lib import CB_PTR_DEVICE
from ..counting._host import encode_msa
from ..counting._stage import PAIR_DTYPE


def get_cherry_transitions(tree, msa: Dict[str, str]) -> List[Tuple[str, str, float]]:
    """:87-139 (`_get_cherry_transitions`): (seq_1, seq_2, total length) per generalised cherry."""
    return [(msa[a], msa[b], la + lb) for a, b, la, lb in _cherry_pairs(tree)]


def get_edge_transitions(tree, msa: Dict[str, str]) -> List[Tuple[str, str, float]]:
    """:393-405 (`_get_edge_transitions`)."""
    assert sorted(tree.nodes()) == sorted(msa.keys())
    return [(msa[u], msa[v], t) for (u, v, t) in tree.edges()]


def _cherry_pairs(tree) -> List[Tuple[str, str, float, float]]:
    """Post-order greedy pairing of the unmatched leaves under every node, children in
    insertion order, an odd leaf is passed up with its accumulated distance (:96-136)."""
    pairs: List[Tuple[str, str, float, float]] = []
    up: Dict[str, Optional[Tuple[str, float]]] = {}
    stack = [(tree.root(), False)]
    while stack:
        node, done = stack.pop()
        if tree.is_leaf(node):
            up[node] = (node, 0.0)
            continue
        if not done:
            stack.append((node, True))
            for child, _ in reversed(tree.children(node)):
                stack.append((child, False))
            continue
        leaves, dists = [], []
        for child, bl in tree.children(node):
            if up[child] is not None:
                leaves.append(up[child][0])
                dists.append(up[child][1] + bl)
        for i in range(0, len(leaves) - 1, 2):
            pairs.append((leaves[i], leaves[i + 1], dists[i], dists[i + 1]))
        up[node] = (leaves[-1], dists[-1]) if len(leaves) % 2 == 1 else None
    assert len(pairs) == len(tree.leaves()) // 2
    return pairs


def _pairs_and_codes(tree, msa, alphabet, transitions_strategy):
    if transitions_strategy == "cherry++":
        assert sorted(tree.leaves()) == sorted(msa.keys())
        named = _cherry_pairs(tree)
    elif transitions_strategy == "edges":
        assert sorted(tree.nodes()) == sorted(msa.keys())
        named = [(u, v, t, 0.0) for (u, v, t) in tree.edges()]
    else:
        raise ValueError(f"Unknown transitions_strategy: {transitions_strategy}")
    names = list(msa.keys())
    row = {nm: i for i, nm in enumerate(names)}
    codes = encode_msa(msa, names, list(alphabet))
    L = codes.shape[1]
    pairs = np.zeros(len(named), dtype=PAIR_DTYPE)
    for k, (a, b, la, lb) in enumerate(named):
        pairs[k] = (row[a] * L, row[b] * L, 0, L, 0, la, lb)
    return pairs, codes


def get_count_prior_probability_matrices(rate_matrix: np.ndarray, quantization_points_sorted) -> np.ndarray:
    """:325-355: prior[b] = diag(pi0) expm(t_b Q0); ValueError when a matrix does not sum to 1.  The reference goes
    through the reversible factorisation on the host (markov_chain/_markov_chain.py:56-155); here the bank expm(t_b Q0)
    is the hot path's own spectral expm kernel (`cb_expm_bank`: symmetrised eigendecomposition + the phi_2 split on the
    device), only the stationary vector of the 20 x 20 model is host arithmetic."""
    from ..bank import CherryBank
    Q0 = np.asarray(rate_matrix, dtype=np.float64)
    grid = np.asarray(quantization_points_sorted, dtype=np.float64)
    w, v = np.linalg.eig(Q0.transpose())
    pi = v[:, int(np.argmin(np.abs(w.real)))].real
    pi = pi / pi.sum()
    with CherryBank.expm_only(grid, Q0.shape[0]) as bank:
        out = pi[None, :, None] * bank.expm_bank(Q0, pi)[0]
    bad = np.abs(out.sum(axis=(1, 2)) - 1.0) > 1e-6
    if np.any(bad) or not np.all(np.isfinite(out)):
        raise ValueError("count_prior_probability_matrices[b, :, :] does not add up to 1!")
    return out


def _assemble_batch(pairs_list, codes_list, grid, site_rates_list, prior, lam, reverse, S, device: int, to_torch: bool,
                    profile=None):
    """cb_siterm_assemble_batch: the mixed count tensor [sum L_f, B, S, S] of many families (sites concatenated)."""
    n_sites = np.ascontiguousarray([c.shape[1] for c in codes_list], dtype=np.int32)
    n_pairs = np.ascontiguousarray([len(p) for p in pairs_list], dtype=np.int64)
    B, Ltot = len(grid), int(n_sites.sum())
    flat = [np.ascontiguousarray(c, dtype=np.int8).reshape(-1) for c in codes_list]
    base = np.concatenate([[0], np.cumsum([f.size for f in flat])])
    shifted = []
    for f, p in enumerate(pairs_list):
        q = np.array(p, dtype=PAIR_DTYPE, copy=True)
        q["seq_a"] += base[f]
        q["seq_b"] += base[f]
        shifted.append(q)
    pairs = np.ascontiguousarray(np.concatenate(shifted)) if shifted else np.zeros(0, dtype=PAIR_DTYPE)
    codes = np.ascontiguousarray(np.concatenate(flat))
    grid = np.ascontiguousarray(grid, dtype=np.float64)
    for c, r in zip(codes_list, site_rates_list):
        if np.size(r) != c.shape[1]:
            raise ValueError(f"site_rates has {np.size(r)} entries, the MSA has {c.shape[1]} sites")
    rates = np.ascontiguousarray(np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1) for r in site_rates_list]))
    prior = np.ascontiguousarray(prior, dtype=np.float64)
    lib = _lib.load()
    if to_torch:
        import torch
        out = torch.empty((Ltot, B, S, S), dtype=torch.float64, device=torch.device("cuda", device))
        torch.cuda.synchronize(out.device)
        ptr, flags = out.data_ptr(), CB_PTR_DEVICE
    else:
        out = np.empty((Ltot, B, S, S))
        ptr, flags = out.ctypes.data, 0
    import ctypes
    ms = ctypes.c_double(0.0)
    rc = lib.cb_siterm_assemble_batch(device, S, B, len(codes_list), n_sites.ctypes.data, grid.ctypes.data, codes.ctypes.data,
                                      codes.size, pairs.ctypes.data, n_pairs.ctypes.data, rates.ctypes.data,
                                      prior.ctypes.data, float(lam), int(bool(reverse)), flags, ptr,
                                      ctypes.addressof(ms) if profile is not None else None)
    _lib.check(rc, "cb_siterm_assemble_batch")
    if profile is not None:
        profile["kernel_ms"] = ms.value
    return out


def _assemble(pairs, codes, grid, site_rates, prior, lam, reverse, S, device: int, to_torch: bool, profile=None):
    return _assemble_batch([pairs], [codes], grid, [site_rates], prior, lam, reverse, S, device, to_torch, profile)


def get_raw_count_matrices(transitions: List[Tuple[str, str, float]], quantization_points_sorted,
                           alphabet: List[str], include_reverse_transitions: bool = True,
                           device: int = 0) -> np.ndarray:
    """:189-261 (`_get_raw_count_matrices`): [L,B,S,S] numpy; counted on the GPU (lambda = 0)."""
    msa = {}
    pairs = np.zeros(len(transitions), dtype=PAIR_DTYPE)
    L = len(transitions[0][0])
    for k, (x, y, t) in enumerate(transitions):
        msa[f"a{k}"], msa[f"b{k}"] = x, y
        pairs[k] = (2 * k * L, (2 * k + 1) * L, 0, L, 0, t, 0.0)
    codes = encode_msa(msa, list(msa.keys()), list(alphabet))
    S, B = len(alphabet), len(quantization_points_sorted)
    return _assemble(pairs, codes, quantization_points_sorted, np.ones(L), np.zeros((B, S, S)), 0.0,
                     include_reverse_transitions, S, device, False)


def estimate_site_specific_rate_matrices_given_trees_and_site_rates(
    trees: List, site_rates: List[List[float]], msas: List[Dict[str, str]], alphabet: List[str],
    regularization_strength: float, regularization_rate_matrix: np.ndarray,
    quantization_points: List[float], optimization_num_epochs: int,
    transitions_strategy: str = "cherry++", include_reverse_transitions: bool = True,
    rate_matrix_parameterization: str = "pande_reversible", use_vectorized_cherryml_implementation: bool = True,
    vectorized_cherryml_implementation_device: str = "cpu",
) -> List[Dict]:
    """MANY families under the same prior / grid / epochs in one go (the reference runs this estimator family by
    family, over a process pool): one `cb_siterm_assemble_batch`, one bank over all families' sites, one device
    optimisation loop.  Sites are independent, so every family's "res" equals what the single-family function
    returns for it; the "time_*" entries are the batch's, repeated in every family's dictionary."""
    from .._device import resolve_device
    resolve_device(vectorized_cherryml_implementation_device, "estimate_site_specific_rate_matrices_given_tree_and_site_rates")
    if rate_matrix_parameterization != "pande_reversible":
        raise NotImplementedError("only the reference's default parameterisation 'pande_reversible'")
    import torch
    from ..bank import CherryBank
    from ._vectorized import _invert
    prof = {}
    st = time.time()
    grid = sorted(quantization_points)
    Q0 = np.asarray(regularization_rate_matrix, dtype=np.float64)
    S = len(alphabet)
    pc = [_pairs_and_codes(tree, msa, alphabet, transitions_strategy) for tree, msa in zip(trees, msas)]
    n_sites = [c.shape[1] for _, c in pc]
    prof["time_get_transitions"] = time.time() - st
    st = time.time()
    prior = get_count_prior_probability_matrices(Q0, grid)
    prof["time_get_count_prior_probability_matrices"] = time.time() - st
    st = time.time()
    dev = torch.cuda.current_device()
    counts = _assemble_batch([p for p, _ in pc], [c for _, c in pc], grid, site_rates, prior, regularization_strength,
                             include_reverse_transitions, S, dev, True)
    totals = counts.sum(dim=(1, 2, 3)).cpu().numpy()
    prof["time_get_count_matrices"] = time.time() - st
    st = time.time()
    rates = np.concatenate([np.asarray(r, dtype=np.float64).reshape(-1) for r in site_rates])
    init = Q0[None, :, :] * rates[:, None, None]
    res = init.copy()
    has = totals > 0
    if has.any():
        idx = np.flatnonzero(has)
        sub = counts if has.all() else counts[torch.as_tensor(idx, device=counts.device)]
        times = np.tile(np.asarray(grid, dtype=np.float64), (len(idx), 1))
        if use_vectorized_cherryml_implementation:
            th0, Th0 = _invert(init[idx])
            with CherryBank(times, sub) as bank:
                r = bank.train_siterm(th0, Th0, int(optimization_num_epochs), lr=0.1)
            res[idx] = r["res"]
        else:
            # The reference's per-site loop (:659-684): every site with counts goes through
            # `_quantized_transitions_mle` (:43-84) = RateMatrixLearner with the "pande_reversible"
            # parameterisation (rate.py:61-95: pi and the upper-triangular logits recovered from the
            # initialisation Q0 * rate_l), no mask, learned pi, Adam lr 0.1, normalised loss, best iterate.
            # Here: the same L optimisations as ONE batched device loop (cb_train_pande_reversible, L > 1).
            from ..estimation._ratelearn._rate_matrix import solve_stationery_dist
            p0 = solve_stationery_dist(Q0)        # scaling by rate_l leaves the stationary distribution alone
            if np.any(np.abs(p0) < 1e-8):
                raise ValueError("Stationary distribution of initialization is degenerate.")
            root = np.sqrt(p0)
            sym = (root[:, None] * Q0) / root[None, :]
            iu = np.triu_indices(S, k=1)
            with np.errstate(divide="ignore"):
                up0 = np.log(np.expm1(sym[iu][None, :] * rates[idx, None]))   # softplus^-1, per site
            lp0 = np.tile(np.log(p0), (len(idx), 1))
            with CherryBank(times, sub) as bank:
                r = bank.train_pande_reversible(up0, lp0, mask=None, num_epochs=int(optimization_num_epochs), lr=0.1,
                                                do_adam=True, normalize=True)
            res[idx] = r["Q_best"].reshape(len(idx), S, S) if int(optimization_num_epochs) > 0 else init[idx]
    prof["time_optimization"] = time.time() - st
    cuts = np.cumsum(n_sites)[:-1]
    return [{"res": part, **prof} for part in np.split(res, cuts)]


def estimate_site_specific_rate_matrices_given_tree_and_site_rates(
    tree, site_rates: List[float], msa: Dict[str, str], alphabet: List[str],
    regularization_strength: float, regularization_rate_matrix: np.ndarray,
    quantization_points: List[float], optimization_num_epochs: int,
    transitions_strategy: str = "cherry++", include_reverse_transitions: bool = True,
    rate_matrix_parameterization: str = "pande_reversible", log_dir: Optional[str] = None,
    plot_site_specific_rate_matrices: int = 0, use_vectorized_cherryml_implementation: bool = True,
    vectorized_cherryml_implementation_device: str = "cpu",
    vectorized_cherryml_implementation_num_cores: int = 1,
) -> Dict:
    """:442-731 (`_estimate_site_specific_rate_matrices_given_tree_and_site_rates`), vectorised path:
    {"res": [L,S,S] site-specific rate matrices, "time_*": seconds per sub-step}.  Sites without
    any count (e.g. all gaps) get the prior `Q0 * rate_l`, like the reference's per-site path
    (:655-658).  `use_vectorized_cherryml_implementation=False` selects the reference's per-site
    semantics (the pande_reversible parameterisation of `_quantized_transitions_mle`, :43-84) -- batched
    over the sites on the device all the same."""
    return estimate_site_specific_rate_matrices_given_trees_and_site_rates(
        [tree], [site_rates], [msa], alphabet, regularization_strength, regularization_rate_matrix, quantization_points,
        optimization_num_epochs, transitions_strategy=transitions_strategy,
        include_reverse_transitions=include_reverse_transitions, rate_matrix_parameterization=rate_matrix_parameterization,
        use_vectorized_cherryml_implementation=use_vectorized_cherryml_implementation,
        vectorized_cherryml_implementation_device=vectorized_cherryml_implementation_device)[0]
