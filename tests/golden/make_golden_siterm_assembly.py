#!/usr/bin/env python3
"""Golden vectors for the SiteRM count / pseudocount assembly, made by RUNNING THE
REFERENCE (cherryml/_siterm/_site_specific_rate_matrix.py) in the build container: same scratch
recipe as make_golden.py.  Written: tests/golden/siterm_assembly.npz with, per case <c>:

  inputs   <c>_edges_u / _edges_v (node names), _edges_t, _msa_names, _msa_seqs, _site_rates,
           _alphabet, _Q0, _lambda, _grid, _strategy, _reverse
  outputs  <c>_n_transitions, <c>_tr_a / _tr_b (sequences of every transition, in order) / _tr_t,
           <c>_raw      _get_raw_count_matrices                         [L,B,S,S]
           <c>_prior    _get_count_prior_probability_matrices           [B,S,S]
           <c>_counts, <c>_times, <c>_init   what the reference hands to
                        quantized_transitions_mle_vectorized_over_sites (compactified)
           <c>_res      the estimator's result for <c>_epochs epochs    [L,S,S]

Usage:  python tests/golden/make_golden_siterm_assembly.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _prepare_scratch  # noqa: E402


def main():
    _prepare_scratch()
    import cherryml.io as cio
    import cherryml._siterm._site_specific_rate_matrix as M

    grid = [0.03 * 1.1 ** i for i in range(-64, 65)]
    out = {}

    def equ(n):
        Q = np.full((n, n), 1.0 / (n - 1))
        np.fill_diagonal(Q, -1.0)
        return Q

    def record(case, tree, msa, site_rates, alphabet, Q0, lam, strategy, reverse, epochs):
        captured = {}
        orig = M.quantized_transitions_mle_vectorized_over_sites

        def spy(counts, times, num_epochs, initialization=None, **kw):
            captured["counts"] = np.array(counts, dtype=np.float64)
            captured["times"] = np.array(times, dtype=np.float64)
            captured["init"] = np.array(initialization, dtype=np.float64)
            return orig(counts=counts, times=times, num_epochs=num_epochs, initialization=initialization, **kw)

        M.quantized_transitions_mle_vectorized_over_sites = spy
        try:
            r = M._estimate_site_specific_rate_matrices_given_tree_and_site_rates(
                tree=tree, site_rates=site_rates, msa=msa, alphabet=alphabet,
                regularization_strength=lam, regularization_rate_matrix=Q0, quantization_points=grid,
                optimization_num_epochs=epochs, transitions_strategy=strategy,
                include_reverse_transitions=reverse, use_vectorized_cherryml_implementation=True)
        finally:
            M.quantized_transitions_mle_vectorized_over_sites = orig
        tr = (M._get_cherry_transitions(tree=tree, msa=msa) if strategy == "cherry++"
              else M._get_edge_transitions(tree=tree, msa=msa))
        raw = M._get_raw_count_matrices(transitions=tr, quantization_points_sorted=sorted(grid),
                                        alphabet=alphabet, include_reverse_transitions=reverse)
        prior = M._get_count_prior_probability_matrices(rate_matrix=Q0, quantization_points_sorted=sorted(grid))
        edges = tree.edges()
        names = list(msa.keys())
        out.update({
            f"{case}_edges_u": np.array([u for u, _, _ in edges]), f"{case}_edges_v": np.array([v for _, v, _ in edges]),
            f"{case}_edges_t": np.array([t for _, _, t in edges], dtype=np.float64),
            f"{case}_nodes": np.array(tree.nodes()),
            f"{case}_msa_names": np.array(names), f"{case}_msa_seqs": np.array([msa[k] for k in names]),
            f"{case}_site_rates": np.array(site_rates, dtype=np.float64), f"{case}_alphabet": np.array(alphabet),
            f"{case}_Q0": Q0, f"{case}_lambda": np.float64(lam), f"{case}_grid": np.array(grid),
            f"{case}_strategy": np.array(strategy), f"{case}_reverse": np.bool_(reverse), f"{case}_epochs": np.int64(epochs),
            f"{case}_tr_a": np.array([a for a, _, _ in tr]), f"{case}_tr_b": np.array([b for _, b, _ in tr]),
            f"{case}_tr_t": np.array([t for _, _, t in tr], dtype=np.float64),
            f"{case}_raw_nz": np.argwhere(raw != 0).astype(np.int32), f"{case}_raw_val": raw[raw != 0],
            f"{case}_raw_shape": np.array(raw.shape), f"{case}_prior": prior,
            f"{case}_counts": captured["counts"], f"{case}_times": captured["times"], f"{case}_init": captured["init"],
            f"{case}_res": np.array(r["res"], dtype=np.float64),
        })
        print(case, "transitions", len(tr), "raw", raw.shape, "compact", captured["counts"].shape)

    # -- the reference's own test fixtures (:734-832, :1129-1190)
    alpha7 = ["A", "D", "G", "S", "T", "V", "-"]
    for equal in (True, False):
        tree = M._get_test_tree_2(node_prefix="node-", equal_edge_lengths=equal)
        msa = M._get_test_msa_2(node_prefix="node-")
        tag = "eq" if equal else "uneq"
        record(f"t2_{tag}_cherry", tree, msa, [2.0, 0.5], alpha7, equ(7), 0.5, "cherry++", True, 30)
        mp = M._maximum_parsimony(tree=tree, msa=msa)
        record(f"t2_{tag}_edges", tree, mp, [2.0, 0.5], alpha7, equ(7), 0.5, "edges", False, 30)
        record(f"t2_{tag}_edges_rev", tree, mp, [2.0, 0.5], alpha7, equ(7), 0.5, "edges", True, 30)
    tree = M._get_test_tree_2(node_prefix="node-")
    record("t2_some_missing", tree, M._get_test_msa_some_all_missing(node_prefix="node-"), [2.0, 0.5], alpha7, equ(7),
           0.5, "cherry++", True, 10)

    # -- a larger random family: 37 leaves (odd: one leaf stays unpaired), multifurcations, gaps,
    #    site rates that push t * rate beyond both ends of the grid, 20-state alphabet, lambda 0.3
    rng = np.random.default_rng(3)
    aa = list("ARNDCQEGHILKMFPSTWYV")
    tree = cio.Tree()
    n_leaves, next_id = 37, 0
    nodes = ["r"]
    tree.add_node("r")
    frontier = ["r"]
    leaves = []
    while len(leaves) + len(frontier) < n_leaves:
        u = frontier.pop(int(rng.integers(len(frontier))))
        for _ in range(int(rng.choice([2, 2, 2, 3]))):
            v = f"n{next_id}"
            next_id += 1
            tree.add_node(v)
            tree.add_edge(u, v, float(np.round(rng.exponential(0.15) + 1e-4, 6)))
            frontier.append(v)
    leaves = [u for u in tree.nodes() if tree.is_leaf(u)]
    L = 23
    lg = np.loadtxt  # noqa: F841
    from cherryml.io import read_rate_matrix
    Q0 = read_rate_matrix("data/rate_matrices/lg.txt").to_numpy()
    msa = {}
    for u in leaves:
        s = rng.choice(aa + ["-"], size=L, p=[0.045] * 20 + [0.1])
        msa[u] = "".join(s)
    site_rates = list(np.round(rng.gamma(3.0, 1.0 / 3.0, size=L), 4))
    site_rates[0], site_rates[1] = 1e-4, 400.0
    record("rand_cherry", tree, msa, site_rates, aa, Q0, 0.3, "cherry++", True, 25)

    np.savez_compressed(os.path.join(HERE, "siterm_assembly.npz"), **out)
    print("wrote", os.path.join(HERE, "siterm_assembly.npz"))


if __name__ == "__main__":
    main()
