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
    """(node -> row, postorder, parent, branch length) of a tree; kept on the tree object: a family's independent sites and its
    contacting pairs are two calls on the same tree"""
    cached = getattr(tree, "_cb_tree_arrays", None)
    if cached is not None:
        return cached
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
    out = (index, np.array([index[v] for v in order], dtype=np.int32), parent, length)
    try:
        tree._cb_tree_arrays = out
    except AttributeError:
        pass
    return out


def _family_arrays(trees: List[Tree], codes_a, codes_b, unit_rates):
    """the concatenated per-family arrays cb_tl_model_run / cb_tree_likelihood_batch take"""
    pairs = codes_b is not None
    n_nodes, n_units, n_cats = [], [], []
    cat_order, cat_parent, cat_length, cat_rates, cat_ucat, cat_a, cat_b = [], [], [], [], [], [], []
    for f, tree in enumerate(trees):
        _, order, parent, length = _tree_arrays(tree)
        rates = _f64(unit_rates[f]).reshape(-1)
        cr, uc = np.unique(rates, return_inverse=True)
        a = np.ascontiguousarray(codes_a[f], dtype=np.int8)
        if a.ndim != 2 or a.shape[0] != parent.size or rates.size != a.shape[1]:
            raise ValueError("codes must be [n_nodes, n_units] and unit_rates [n_units]")
        if pairs:
            b = np.ascontiguousarray(codes_b[f], dtype=np.int8)
            if b.shape != a.shape:
                raise ValueError("codes must be [n_nodes, n_units] and unit_rates [n_units]")
            cat_b.append(b.reshape(-1))
        n_nodes.append(parent.size), n_units.append(a.shape[1]), n_cats.append(cr.size)
        cat_order.append(order), cat_parent.append(parent), cat_length.append(length)
        cat_rates.append(cr), cat_ucat.append(uc.astype(np.int32)), cat_a.append(a.reshape(-1))
    i32 = lambda x: np.ascontiguousarray(x, dtype=np.int32)   # noqa: E731
    return dict(n_nodes=i32(n_nodes), n_units=i32(n_units), n_cats=i32(n_cats), order=i32(np.concatenate(cat_order)),
                parent=i32(np.concatenate(cat_parent)), length=_f64(np.concatenate(cat_length)),
                cat_rate=_f64(np.concatenate(cat_rates)), unit_cat=i32(np.concatenate(cat_ucat)),
                a=np.ascontiguousarray(np.concatenate(cat_a), dtype=np.int8),
                b=np.ascontiguousarray(np.concatenate(cat_b), dtype=np.int8) if pairs else None)


class LikelihoodModel:
    """ONE substitution model resident on the device (`cb_tl_model_create` / `_run` / `_destroy`): Q, its stationary distributions,
    the counts-free expm handle with the model's eigendecomposition, the transition bank's and the messages' buffers are kept
    between calls.  The reference evaluates family after family under one model (evaluation/_likelihood.py:474-600, one process
    per family); `tree_likelihood_batch` follows that signature and makes, uses and frees all of the above per call -- 21 ms per
    1024-leaf family of the 400-state pair model around 10.7 ms of kernels (2.6 GB of bank allocated and freed each time).  Here:
    `model = LikelihoodModel(Q, pi_root, pairs=True)` once, then `model.log_likelihoods(trees, codes_a, codes_b, unit_rates)`."""

    def __init__(self, Q, pi_root, pairs: bool = False, reversible: bool = True, alphabet_size: Optional[int] = None,
                 device: int = 0):
        import ctypes
        Q, pi_root = _f64(Q), _f64(pi_root).reshape(-1)
        self.S, self.pairs, self.device = Q.shape[0], bool(pairs), int(device)
        self.S1 = 0 if not pairs else int(alphabet_size if alphabet_size is not None else round(self.S ** 0.5))
        pi_rev = _f64(_stationary_distribution(Q)) if reversible else None
        self._h = ctypes.c_void_p()
        _lib.check(_lib.load().cb_tl_model_create(self.device, self.S, self.S1, Q.ctypes.data,
                                                  None if pi_rev is None else pi_rev.ctypes.data, pi_root.ctypes.data,
                                                  ctypes.byref(self._h)), "cb_tl_model_create")

    def log_likelihoods(self, trees: List[Tree], codes_a: List[np.ndarray], codes_b: Optional[List[np.ndarray]], unit_rates: List,
                        profile: Optional[dict] = None) -> List[np.ndarray]:
        """per family the log-likelihood of every unit: what `tree_likelihood_batch` returns"""
        if self._h is None:
            raise _lib.CherryBankError("LikelihoodModel: closed")
        if len(trees) == 0:
            return []
        if (codes_b is not None) != self.pairs:
            raise ValueError("the model was made for " + ("pairs of sites" if self.pairs else "single sites"))
        z = _family_arrays(trees, codes_a, codes_b, unit_rates)
        ll, ms = np.empty(int(z["n_units"].sum())), np.zeros(2)
        rc = _lib.load().cb_tl_model_run(
            self._h, len(trees), z["n_nodes"].ctypes.data, z["order"].ctypes.data, z["parent"].ctypes.data, z["length"].ctypes.data,
            z["n_cats"].ctypes.data, z["cat_rate"].ctypes.data, z["n_units"].ctypes.data, z["unit_cat"].ctypes.data,
            z["a"].ctypes.data, None if z["b"] is None else z["b"].ctypes.data, ll.ctypes.data, ms.ctypes.data)
        _lib.check(rc, "cb_tl_model_run")
        if profile is not None:
            profile["kernel_ms"] = profile.get("kernel_ms", 0.0) + float(ms[0])
            key = "prune_ms_pairs" if self.pairs else "prune_ms_sites"
            profile[key] = profile.get(key, 0.0) + float(ms[1])
        return np.split(ll, np.cumsum(z["n_units"])[:-1])

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.load().cb_tl_model_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def tree_likelihood_batch(trees: List[Tree], codes_a: List[np.ndarray], codes_b: Optional[List[np.ndarray]], Q, pi_root,
                          unit_rates: List, reversible: bool = True, alphabet_size: Optional[int] = None,
                          device: int = 0, profile: Optional[dict] = None) -> List[np.ndarray]:
    """Log-likelihood of every unit of MANY families under one model, one call of cb_tree_likelihood_batch:
    family f has tree `trees[f]`, `codes_a[f]` [n_nodes, n_units] (rows in `tree.nodes()` order, -1 =
    unobserved; `codes_b[f]` for pairs) and `unit_rates[f]` [n_units] scaling Q per unit.  The model's
    eigendecomposition is shared by the families (400-state pair model: one eigensolve for the batch).  (A model made, used
    and freed by this call: `LikelihoodModel` keeps it for the next one.)"""
    if len(trees) == 0:
        return []
    with LikelihoodModel(Q, pi_root, pairs=codes_b is not None, reversible=reversible, alphabet_size=alphabet_size,
                         device=device) as model:
        return model.log_likelihoods(trees, codes_a, codes_b, unit_rates, profile=profile)


def tree_likelihood(tree: Tree, codes_a: np.ndarray, codes_b: Optional[np.ndarray], Q, pi_root, unit_rates,
                    reversible: bool = True, alphabet_size: Optional[int] = None, device: int = 0,
                    profile: Optional[dict] = None) -> np.ndarray:
    """Log-likelihood of every unit (column of `codes_a` [n_nodes, n_units], rows in `tree.nodes()`
    order, -1 = unobserved; `codes_b` for pairs).  `unit_rates[u]` scales Q for unit u."""
    return tree_likelihood_batch([tree], [codes_a], None if codes_b is None else [codes_b], Q, pi_root, [unit_rates],
                                 reversible=reversible, alphabet_size=alphabet_size, device=device, profile=profile)[0]


def _family_units(tree: Tree, msa: Dict[str, str], contact_map: Optional[np.ndarray], num_sites: int,
                  amino_acids: List[str], with_pairs: bool):
    """(independent sites, contacting pairs, leaf state codes [n_nodes, num_sites]) of one family (:88-126)"""
    if contact_map is not None and with_pairs:
        ii, jj = np.nonzero(np.triu(np.asarray(contact_map) == 1, 1))
        pairs = list(zip(ii.tolist(), jj.tolist()))
    else:
        pairs = []
    flat = [s for p in pairs for s in p]
    if len(set(flat)) != len(flat):   # :88-95
        raise Exception(f"Each site can only be in contact with one other site. The contacting sites were: {pairs}")
    in_pair = set(flat)
    indep = [i for i in range(num_sites) if i not in in_pair]
    nodes = tree.nodes()
    lut = np.full(256, -1, dtype=np.int8)
    for i, aa in enumerate(amino_acids):
        if len(aa) == 1 and ord(aa) < 256:
            lut[ord(aa)] = i
    codes = np.full((len(nodes), num_sites), -1, dtype=np.int8)
    # all leaf sequences through the look-up table in one pass (a row per leaf that the alignment holds)
    rows = [r for r, v in enumerate(nodes) if tree.is_leaf(v) and v in msa]
    if rows:
        text = "".join(msa[nodes[r]] for r in rows).encode("latin-1")
        if len(text) != len(rows) * num_sites:
            raise ValueError("every sequence of the alignment must have one character per site rate")
        codes[rows] = lut[np.frombuffer(text, dtype=np.uint8)].reshape(len(rows), num_sites)
    return indep, pairs, codes


def dp_likelihood_computation_batch(trees: List[Tree], msas: List[Dict[str, str]], contact_maps: List[Optional[np.ndarray]],
                                    site_rates: List[List[float]], amino_acids: List[str], pi_1: np.ndarray,
                                    Q_1: np.ndarray, reversible_1: bool = True, pi_2: Optional[np.ndarray] = None,
                                    Q_2: Optional[np.ndarray] = None, reversible_2: Optional[bool] = True,
                                    device: int = 0, profile: Optional[dict] = None,
                                    model_1: Optional["LikelihoodModel"] = None,
                                    model_2: Optional["LikelihoodModel"] = None) -> List[Tuple[float, List[float]]]:
    """`dp_likelihood_computation` for many families under the same models: two GPU calls for the whole batch
    (independent sites; contacting pairs), the reference's per-family results.  `model_1` / `model_2`: the two models already
    resident on the device (`LikelihoodModel`; Q_1 / Q_2 are then not looked at) -- what a caller that evaluates family after
    family keeps between its calls."""
    profile = {} if profile is None else profile
    units = [_family_units(t, m, c, len(r), amino_acids, Q_2 is not None)
             for t, m, c, r in zip(trees, msas, contact_maps, site_rates)]
    lls = [[0.0] * len(r) for r in site_rates]
    with_sites = [f for f, (indep, _, _) in enumerate(units) if indep]
    if with_sites:
        args = ([trees[f] for f in with_sites], [units[f][2][:, units[f][0]] for f in with_sites], None)
        rates_1 = [[site_rates[f][i] for i in units[f][0]] for f in with_sites]
        if model_1 is not None:
            out = model_1.log_likelihoods(*args, rates_1, profile=profile)
        else:
            out = tree_likelihood_batch(*args, Q_1, pi_1, rates_1, reversible=bool(reversible_1), device=device, profile=profile)
        for f, ll1 in zip(with_sites, out):
            for i, x in zip(units[f][0], ll1):
                lls[f][i] = float(x)
    with_pairs = [f for f, (_, pairs, _) in enumerate(units) if pairs]
    if with_pairs:
        ia = {f: [p[0] for p in units[f][1]] for f in with_pairs}
        ib = {f: [p[1] for p in units[f][1]] for f in with_pairs}
        args = ([trees[f] for f in with_pairs], [units[f][2][:, ia[f]] for f in with_pairs], [units[f][2][:, ib[f]] for f in with_pairs])
        rates_2 = [np.ones(len(ia[f])) for f in with_pairs]
        if model_2 is not None:
            out = model_2.log_likelihoods(*args, rates_2, profile=profile)
        else:
            out = tree_likelihood_batch(*args, Q_2, pi_2, rates_2, reversible=bool(reversible_2), alphabet_size=len(amino_acids),
                                        device=device, profile=profile)
        for f, ll2 in zip(with_pairs, out):
            for (i, j), x in zip(units[f][1], ll2):
                lls[f][i] = float(x) / 2.0
                lls[f][j] = float(x) / 2.0
    return [(sum(l), l) for l in lls]


def dp_likelihood_computation(tree: Tree, msa: Dict[str, str], contact_map: Optional[np.ndarray],
                              site_rates: List[float], amino_acids: List[str], pi_1: np.ndarray, Q_1: np.ndarray,
                              fact_1=None, reversible_1: bool = True, device_1=None,
                              pi_2: Optional[np.ndarray] = None, Q_2: Optional[np.ndarray] = None, fact_2=None,
                              reversible_2: Optional[bool] = True, device_2=None,
                              output_profiling_path: Optional[str] = None, device: int = 0,
                              profile: Optional[dict] = None, model_1: Optional["LikelihoodModel"] = None,
                              model_2: Optional["LikelihoodModel"] = None) -> Tuple[float, List[float]]:
    """`dp_likelihood_computation` (_likelihood.py:47-327).  `fact_*` / `device_*` are accepted for
    call compatibility and ignored: the spectral factorisation happens on the GPU."""
    st_all = time.time()
    profile = {} if profile is None else profile
    res = dp_likelihood_computation_batch([tree], [msa], [contact_map], [list(site_rates)], amino_acids, pi_1, Q_1,
                                          reversible_1, pi_2, Q_2, reversible_2, device=device, profile=profile,
                                          model_1=model_1, model_2=model_2)[0]
    if output_profiling_path is not None:
        with open(output_profiling_path, "w") as f:
            f.write(f"GPU time (expm bank + pruning): {profile.get('kernel_ms', 0.0) / 1e3}\n"
                    f"Total time: {time.time() - st_all}\n")
    return res


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
    per-site values) and `<family>.profiling` in `output_likelihood_dir`.  Families run in batches
    on one GPU (cb_tree_likelihood_batch: shared eigendecompositions and uploads); `num_processes`, `device_1/2` and the CPU threading knobs are accepted and
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
    # families in chunks (bounded host memory); each chunk is two GPU calls on the two models, which stay on the device for the
    # whole stage (their eigendecompositions, expm handles and buffers are made once: LikelihoodModel)
    chunk = 64
    model_1 = LikelihoodModel(Q_1_df.to_numpy(), pi_1_df.to_numpy(), pairs=False, reversible=bool(reversible_1), device=device)
    model_2 = (LikelihoodModel(Q_2_df.to_numpy(), pi_2_df.to_numpy(), pairs=True, reversible=bool(reversible_2),
                               alphabet_size=len(amino_acids), device=device) if Q_2_df is not None else None)
    for c0 in range(0, len(families), chunk):
        fams = families[c0:c0 + chunk]
        st_c = time.time()
        trees = [read_tree(os.path.join(tree_dir, f + ".txt")) for f in fams]
        msas = [read_msa(os.path.join(msa_dir, f + ".txt")) for f in fams]
        rates = [list(read_site_rates(os.path.join(site_rates_dir, f + ".txt"))) for f in fams]
        cmaps = [read_contact_map(os.path.join(contact_map_dir, f + ".txt")) if contact_map_dir is not None else None
                 for f in fams]
        profile = {}
        results = dp_likelihood_computation_batch(
            trees, msas, cmaps, rates, amino_acids, pi_1_df.to_numpy(), Q_1_df.to_numpy(), reversible_1,
            pi_2_df.to_numpy() if pi_2_df is not None else None, Q_2_df.to_numpy() if Q_2_df is not None else None,
            reversible_2, device=device, profile=profile, model_1=model_1, model_2=model_2)
        for family, res in zip(fams, results):
            write_log_likelihood(res, os.path.join(output_likelihood_dir, family + ".txt"))
            with open(os.path.join(output_likelihood_dir, family + ".profiling"), "w") as f:   # per family: its share
                f.write(f"GPU time (expm bank + pruning): {profile.get('kernel_ms', 0.0) / 1e3 / len(fams)}\n"
                        f"Total time: {(time.time() - st_c) / len(fams)}\n")
    model_1.close()
    if model_2 is not None:
        model_2.close()
    with open(os.path.join(output_likelihood_dir, "profiling_0.txt"), "w") as f:
        f.write(f"Total time: {time.time() - st}\n")
