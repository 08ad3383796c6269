"""SiteRM callers of the hot path: `learn_site_rate_matrices` (reference
cherryml/_siterm/_learn_site_rate_matrix.py:1109-1282) and the public
`learn_site_specific_rate_matrices` (cherryml/_siterm_public_api.py:21-172), same names, arguments and
result keys.  Every numerical step runs on the GPU through libcherrybank:

  site rates      `_estimate_site_rates_fast` (:387-474): log expm(rate * t_cherry * Q) from the expm bank
                  (`cb_ble_log_bank`), per-site arg-max over the rate grid (`cb_site_rate_gather`)
  count tensors   `cb_siterm_assemble`          (cherry++ transitions, pseudocounts; bit-exact)
  optimiser       `cb_train_siterm`             (all sites, all epochs, one call)

`tree=None` runs FastCherries (cherryml_amd/phylogeny_estimation/_fast_cherries.py: the reference's seeded
divide-and-conquer pairing on the host, branch lengths and site rates on the GPU) exactly as the reference
does here -- 20 rate categories, 50 iterations, seed 1234, the default 129-point grid, sequences in sorted
name order -- and takes its site rates.  The non-vectorised per-site CPU loop of the reference
(`use_vectorized_implementation=False`) has no counterpart: there is no CPU path."""
import time
from typing import Dict, List, Optional

import numpy as np

from ..phylogeny_estimation._ble import compute_log_transition_matrices
from ._assembly import _cherry_pairs, estimate_site_specific_rate_matrices_given_tree_and_site_rates
from ._site_rates import compute_optimal_site_rates

# the reference's module constants (_learn_site_rate_matrix.py:20-23)
QUANTIZATION_GRID_CENTER = 0.03
QUANTIZATION_GRID_STEP = 1.1
QUANTIZATION_GRID_NUM_STEPS = 64


def get_standard_site_rate_grid(num_site_rates: int = 20) -> List[float]:
    """Site rate grid of the FastCherries / SiteRM paper (:933-941)."""
    return [
        num_site_rates ** (-1.0 + 2.0 * (num_site_rates - i) / (num_site_rates - 1.0))
        for i in range(1, num_site_rates + 1)
    ][::-1]


def get_standard_site_rate_prior(num_site_rates: int = 20) -> List[float]:
    """Gamma(shape 3, scale 1/3) density on the standard grid (:944-952)."""
    from scipy.stats import gamma
    return [float(gamma.pdf(r, a=3.0, scale=1.0 / 3.0)) for r in get_standard_site_rate_grid(num_site_rates)]


def _matrix_and_states(rate_matrix, alphabet: Optional[List[str]] = None):
    """pandas DataFrame (the reference's type) or a plain array + alphabet -> (ndarray, states)."""
    if hasattr(rate_matrix, "to_numpy"):
        return np.asarray(rate_matrix.to_numpy(), dtype=np.float64), [str(c) for c in rate_matrix.columns]
    Q = np.asarray(rate_matrix, dtype=np.float64)
    if alphabet is None or len(alphabet) != Q.shape[0]:
        raise ValueError("a plain-array rate matrix needs its alphabet")
    return Q, list(alphabet)


def _stationary_if_reversible(Q: np.ndarray) -> Optional[np.ndarray]:
    """pi with pi Q = 0 when Q is reversible w.r.t. it (then the spectral expm kernels apply), else None."""
    w, v = np.linalg.eig(Q.T)
    p = np.real(v[:, np.argmin(np.abs(w))])
    if p.sum() == 0.0:
        return None
    p = p / p.sum()
    if np.any(p <= 0.0):
        return None
    flux = p[:, None] * Q
    if not np.allclose(flux, flux.T, rtol=1e-9, atol=1e-12 * np.abs(flux).max()):
        return None
    return p


def _estimate_site_rates_fast(tree, leaf_states: Dict[str, str], site_rate_grid: List[float],
                              site_rate_prior: List[float], rate_matrix, alphabet: Optional[List[str]] = None,
                              device: int = 0) -> List[float]:
    """:387-474.  Per site, the rate of the grid maximising log prior + sum over the cherries (both
    directions) of log expm(rate * t * Q)[x, y]; symbols outside the alphabet (gaps) contribute 0."""
    if len(site_rate_grid) == 1:
        return [site_rate_grid[0]] * len(next(iter(leaf_states.values())))
    Q, states = _matrix_and_states(rate_matrix, alphabet)
    S = Q.shape[0]
    code = np.full(256, S, dtype=np.int8)        # everything else -> the extra all-zero state
    for i, s in enumerate(states):
        if len(s) != 1 or ord(s) >= 128:
            raise ValueError("the fast site-rate path assumes a single-character ASCII alphabet (as the reference)")
        code[ord(s)] = i
    enc = {leaf: code[np.frombuffer(seq.encode("ascii", "replace"), dtype=np.uint8)] for leaf, seq in leaf_states.items()}
    pairs = _cherry_pairs(tree)
    if not pairs:
        raise ValueError("the tree has no cherry")
    fwd = [(enc[a], enc[b], la + lb) for a, b, la, lb in pairs]
    cherries = fwd + [(y, x, t) for (x, y, t) in fwd]
    lengths = np.array([t for (_, _, t) in fwd], dtype=np.float64)
    # log expm(t_c * rate_r * Q): [cherry, rate, S, S] from the expm bank of the hot path (a reversed
    # cherry has the length of its forward twin: the same matrices)
    logP = compute_log_transition_matrices(Q, lengths, site_rate_grid, device=device,
                                           stationary_distribution=_stationary_if_reversible(Q))
    tens = np.zeros((len(site_rate_grid), len(cherries), S + 1, S + 1))
    tens[:, :len(fwd), :S, :S] = np.transpose(logP, (1, 0, 2, 3))
    tens[:, len(fwd):, :S, :S] = tens[:, :len(fwd), :S, :S]
    num_sites = len(cherries[0][0])
    return compute_optimal_site_rates(num_sites, cherries, tens, list(site_rate_grid), list(site_rate_prior),
                                      device=device)


def learn_site_rate_matrices(
    tree,
    leaf_states: Dict[str, str],
    alphabet: List[str],
    regularization_rate_matrix,
    regularization_strength: float,
    use_vectorized_implementation: bool = True,
    vectorized_implementation_device: str = "cpu",
    vectorized_implementation_num_cores: int = 1,
    site_rate_grid: List[float] = [2.0 ** i for i in range(-10, 10)],
    site_rate_prior: List[float] = [1.0 for i in range(-10, 10)],
    alphabet_for_site_rate_estimation: Optional[List[str]] = None,
    rate_matrix_for_site_rate_estimation=None,
    num_epochs: int = 100,
    use_fast_site_rate_implementation: bool = True,
    quantization_grid_num_steps: int = QUANTIZATION_GRID_NUM_STEPS,
    just_run_fast_cherries: bool = False,
) -> Dict:
    """:1109-1282.  Returns {"learnt_rate_matrices": [L,S,S], "learnt_site_rates": [L], "learnt_tree": tree,
    "time_*": seconds}.  `use_vectorized_implementation=False` keeps the reference's per-site semantics
    (pande_reversible parameterisation per site) but still runs all sites as one batched device loop; only the
    fast site-rate estimator exists."""
    from .._device import resolve_device
    vectorized_implementation_device = resolve_device(vectorized_implementation_device, "learn_site_rate_matrices")
    prof = {}
    st = time.time()
    if alphabet_for_site_rate_estimation is None:
        alphabet_for_site_rate_estimation = list(alphabet)
    if rate_matrix_for_site_rate_estimation is None:
        rate_matrix_for_site_rate_estimation = regularization_rate_matrix
    Qreg, reg_states = _matrix_and_states(regularization_rate_matrix, alphabet)
    Qsr, sr_states = _matrix_and_states(rate_matrix_for_site_rate_estimation, alphabet_for_site_rate_estimation)
    assert sr_states == list(alphabet_for_site_rate_estimation)   # as the reference (:1189-1190)
    assert reg_states == list(alphabet)
    site_rate_grid, site_rate_prior = list(site_rate_grid), list(site_rate_prior)
    prof["time_init_learn_site_rate_matrices"] = time.time() - st
    st = time.time()
    import torch
    dev = torch.cuda.current_device()
    site_rates_fast_cherries = None
    if tree is None:
        # :1196-1228 (there through temporary files): FastCherries with the site-rate matrix / alphabet
        from ..phylogeny_estimation._fast_cherries import cherries_to_tree, fast_cherries_family
        names = sorted(leaf_states.keys())
        cherries, lengths, rates = fast_cherries_family(
            names, [leaf_states[n] for n in names], Qsr, sr_states, num_rate_categories=20, max_iters=50,
            seed=1234, device=dev)
        tree = cherries_to_tree(names, cherries, lengths)
        site_rates_fast_cherries = [float(r) for r in rates]
    elif just_run_fast_cherries:
        raise ValueError("If just_run_fast_cherries is True, then tree must be None.")
    time_estimate_tree = time.time() - st
    st = time.time()
    if site_rates_fast_cherries is not None:
        site_rates = site_rates_fast_cherries
    else:
        site_rates = _estimate_site_rates_fast(tree, leaf_states, site_rate_grid, site_rate_prior, Qsr, sr_states,
                                               device=dev)
    time_estimate_site_rate = time.time() - st
    if just_run_fast_cherries:
        return {"learnt_rate_matrices": None, "learnt_site_rates": site_rates, "learnt_tree": tree,
                "time_estimate_tree": time_estimate_tree, "time_estimate_site_rate": time_estimate_site_rate}
    # :650-716 `_learn_site_rate_matrices_given_site_rates_too`: the grid keeps its span, 2n+1 points
    st = time.time()
    step = QUANTIZATION_GRID_STEP ** (QUANTIZATION_GRID_NUM_STEPS / quantization_grid_num_steps)
    points = [QUANTIZATION_GRID_CENTER * step ** i
              for i in range(-quantization_grid_num_steps, quantization_grid_num_steps + 1)]
    prof["time_build_quantization_points"] = time.time() - st
    r = estimate_site_specific_rate_matrices_given_tree_and_site_rates(
        tree=tree, site_rates=site_rates, msa=leaf_states, alphabet=list(alphabet),
        regularization_strength=regularization_strength, regularization_rate_matrix=Qreg,
        quantization_points=points, optimization_num_epochs=num_epochs, transitions_strategy="cherry++",
        include_reverse_transitions=True, rate_matrix_parameterization="pande_reversible",
        use_vectorized_cherryml_implementation=bool(use_vectorized_implementation),
        vectorized_cherryml_implementation_device="cuda",
        vectorized_cherryml_implementation_num_cores=vectorized_implementation_num_cores)
    prof.update({k: v for k, v in r.items() if k.startswith("time_")})
    return {"learnt_rate_matrices": r["res"], "learnt_site_rates": site_rates, "learnt_tree": tree,
            "time_estimate_tree": time_estimate_tree, "time_estimate_site_rate": time_estimate_site_rate, **prof}


def learn_site_specific_rate_matrices(
    tree,
    msa: Dict[str, str],
    alphabet: List[str],
    regularization_rate_matrix,
    regularization_strength: float = 0.5,
    device: str = "cpu",
    num_rate_categories: int = 20,
    alphabet_for_site_rate_estimation: Optional[List[str]] = None,
    rate_matrix_for_site_rate_estimation=None,
    num_epochs: int = 100,
    quantization_grid_num_steps: int = 64,
    use_vectorized_implementation: bool = True,
    just_run_fast_cherries: bool = False,
) -> Dict:
    """The SiteRM public entry point (cherryml/_siterm_public_api.py:21-172): standard site-rate grid and
    Gamma(3, 1/3) prior, fast site-rate estimator, vectorised optimiser.  `device` keeps the reference's
    default "cpu"; either spelling runs on the MI355X (cherryml_amd/_device.py)."""
    return learn_site_rate_matrices(
        tree=tree, leaf_states=msa, alphabet=alphabet, regularization_rate_matrix=regularization_rate_matrix,
        regularization_strength=regularization_strength,
        use_vectorized_implementation=use_vectorized_implementation,
        vectorized_implementation_device=device, vectorized_implementation_num_cores=1,
        site_rate_grid=get_standard_site_rate_grid(num_site_rates=num_rate_categories),
        site_rate_prior=get_standard_site_rate_prior(num_site_rates=num_rate_categories),
        alphabet_for_site_rate_estimation=alphabet_for_site_rate_estimation,
        rate_matrix_for_site_rate_estimation=rate_matrix_for_site_rate_estimation,
        num_epochs=num_epochs, use_fast_site_rate_implementation=True,
        quantization_grid_num_steps=quantization_grid_num_steps, just_run_fast_cherries=just_run_fast_cherries)
