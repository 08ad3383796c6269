#!/usr/bin/env python3
"""Golden vectors for the SiteRM callers (`learn_site_rate_matrices`, reference
cherryml/_siterm/_learn_site_rate_matrix.py:1109-1282, with `_estimate_site_rates_fast` :387-474),
made by RUNNING THE REFERENCE in the build container (same scratch recipe as make_golden.py).
Trees are built through the reference's own Tree API (its newick helper needs ete3, absent here).

Written: tests/golden/siterm_learn.npz with, per case <c>:
  inputs   <c>_edges_u / _edges_v / _edges_t (case "fc" = tree=None: the edges of the FastCherries tree, see below), <c>_msa_names / _msa_seqs, <c>_alphabet, <c>_Q0, <c>_lambda,
           <c>_sr_alphabet, <c>_sr_Q, <c>_grid, <c>_prior, <c>_epochs, <c>_qsteps
  outputs  <c>_site_rates (learnt_site_rates), <c>_res (learnt_rate_matrices [L,S,S])

Usage:  python tests/golden/make_golden_siterm_learn.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _prepare_scratch  # noqa: E402


def main():
    _prepare_scratch()
    import pandas as pd
    import cherryml.io as cio
    import cherryml._siterm._learn_site_rate_matrix as M

    out = {}

    def equ(states, scale=1.0):
        n = len(states)
        Q = np.full((n, n), 1.0 / (n - 1))
        np.fill_diagonal(Q, -1.0)
        return pd.DataFrame(Q * scale, index=states, columns=states)

    def make_tree(edges):
        t = cio.Tree()
        nodes = []
        for u, v, _ in edges:
            for x in (u, v):
                if x not in nodes:
                    nodes.append(x)
        t.add_nodes(nodes)
        t.add_edges(edges)
        return t

    def record(case, edges, msa, alphabet, Q0, lam, sr_alphabet, sr_Q, num_rates, epochs, qsteps):
        grid = M.get_standard_site_rate_grid(num_site_rates=num_rates)
        prior = M.get_standard_site_rate_prior(num_site_rates=num_rates)
        r = M.learn_site_rate_matrices(
            tree=make_tree(edges), leaf_states=msa, alphabet=alphabet, regularization_rate_matrix=Q0,
            regularization_strength=lam, use_vectorized_implementation=True,
            vectorized_implementation_device="cpu", vectorized_implementation_num_cores=1,
            site_rate_grid=grid, site_rate_prior=prior, alphabet_for_site_rate_estimation=sr_alphabet,
            rate_matrix_for_site_rate_estimation=sr_Q, num_epochs=epochs,
            use_fast_site_rate_implementation=True, quantization_grid_num_steps=qsteps)
        out[f"{case}_edges_u"] = np.array([e[0] for e in edges])
        out[f"{case}_edges_v"] = np.array([e[1] for e in edges])
        out[f"{case}_edges_t"] = np.array([e[2] for e in edges], dtype=np.float64)
        names = sorted(msa)
        out[f"{case}_msa_names"] = np.array(names)
        out[f"{case}_msa_seqs"] = np.array([msa[n] for n in names])
        out[f"{case}_alphabet"] = np.array(alphabet)
        out[f"{case}_Q0"] = Q0.to_numpy()
        out[f"{case}_lambda"] = np.float64(lam)
        out[f"{case}_sr_alphabet"] = np.array(sr_alphabet if sr_alphabet is not None else alphabet)
        out[f"{case}_sr_Q"] = (sr_Q if sr_Q is not None else Q0).to_numpy()
        out[f"{case}_grid"] = np.array(grid)
        out[f"{case}_prior"] = np.array(prior)
        out[f"{case}_epochs"] = np.int64(epochs)
        out[f"{case}_qsteps"] = np.int64(qsteps)
        out[f"{case}_site_rates"] = np.array(r["learnt_site_rates"], dtype=np.float64)
        out[f"{case}_res"] = np.array(r["learnt_rate_matrices"], dtype=np.float64)
        print(case, "site rates", np.round(out[f"{case}_site_rates"][:6], 6), "res", out[f"{case}_res"].shape)

    dna = ["A", "C", "G", "T"]
    # 1. the reference's own public-API test (_siterm_public_api.py:175-209): expected site rate 0.6231236
    edges1 = [("r", "a", 1.0), ("a", "b", 1.0), ("b", "leaf_1", 1.0), ("b", "leaf_2", 1.0),
              ("r", "c", 1.0), ("c", "d", 1.0), ("d", "leaf_3", 1.0), ("d", "leaf_4", 1.0)]
    record("pub", edges1, {"leaf_1": "C", "leaf_2": "C", "leaf_3": "C", "leaf_4": "G"}, dna, equ(dna), 0.5,
           None, None, 20, 100, 64)
    # 2. random multifurcating tree, gaps, separate site-rate alphabet / matrix, coarse grid
    rng = np.random.default_rng(3)
    n_leaves, L = 13, 24
    edges, leaves, internals, nxt = [], [], ["n0"], 1
    while len(leaves) < n_leaves:
        p = internals[rng.integers(len(internals))]
        if rng.random() < 0.45 and len(internals) < 8:
            c = f"n{nxt}"
            internals.append(c)
        else:
            c = f"s{nxt}"
            leaves.append(c)
        nxt += 1
        edges.append((p, c, float(np.round(rng.gamma(2.0, 0.05), 6))))
    # internal nodes without children become leaves too
    parents = {e[0] for e in edges}
    leaves = [v for (_, v, _) in edges if v not in parents]
    alpha5 = dna + ["-"]
    msa = {}
    base = rng.integers(0, 4, size=L)
    for leaf in leaves:
        seq = base.copy()
        flip = rng.random(L) < 0.25
        seq[flip] = rng.integers(0, 4, size=int(flip.sum()))
        chars = np.array(list("ACGT"))[seq]
        chars[rng.random(L) < 0.1] = "-"
        msa[leaf] = "".join(chars)
    # (nodes that are internal in `internals` but childless were renamed leaves above; give them sequences)
    record("rand", edges, msa, alpha5, equ(alpha5), 0.3, dna, equ(dna, 1.3), 8, 25, 8)
    # 3. tree=None: FastCherries estimates the cherries and the site rates (20 categories, seed 1234)
    n_leaves, L = 21, 40
    base = rng.integers(0, 4, size=L)
    msa3 = {}
    for i in range(n_leaves):
        seq = base.copy()
        flip = rng.random(L) < (0.1 + 0.3 * (i % 3))
        seq[flip] = rng.integers(0, 4, size=int(flip.sum()))
        chars = np.array(list("ACGT"))[seq]
        chars[rng.random(L) < 0.05] = "-"
        msa3[f"t{i:02d}"] = "".join(chars)
    # The reference's wrapper assembles the star-of-cherries tree with ete3 (absent here), so this case is
    # glued from the reference's two halves: its C++ PROGRAM (oracle/_ref/libref_fc.so, as in
    # make_golden_fast_cherries.py; arguments as learn_site_rate_matrices passes them, :1212-1223: 20
    # categories, 50 iterations, seed 1234, sequences in sorted order as write_msa writes them) and its
    # `_learn_site_rate_matrices_given_site_rates_too` (:650-716), with the tree rule of
    # _fast_cherries.py:116-131 restated in between (root -> internal-<i> at 1.0 -> two leaves at d/2 each;
    # the odd sequence out under the root at 1.0).
    import ctypes as C
    from make_golden_fast_cherries import run_program
    lib = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libref_fc.so"))
    names3 = sorted(msa3)
    ch, le, ra = run_program(lib, names3, [msa3[n] for n in names3], equ(dna).to_numpy(), dna, 20, 50, 1234)
    edges3, paired = [], set()
    for i, ((a, b), d) in enumerate(zip(ch, le)):
        edges3 += [("root", f"internal-{i}", 1.0), (f"internal-{i}", str(a), float(d) / 2.0),
                   (f"internal-{i}", str(b), float(d) / 2.0)]
        paired |= {str(a), str(b)}
    missing = [n for n in names3 if n not in paired]
    if len(missing) & 1:
        edges3.append(("root", missing[-1], 1.0))
    r3 = M._learn_site_rate_matrices_given_site_rates_too(
        tree=make_tree(edges3), site_rates=[float(x) for x in ra], leaf_states=msa3, alphabet=alpha5,
        regularization_rate_matrix=equ(alpha5), regularization_strength=0.5,
        use_vectorized_cherryml_implementation=True, vectorized_cherryml_implementation_device="cpu",
        vectorized_cherryml_implementation_num_cores=1, num_epochs=20, quantization_grid_num_steps=16)
    out["fc_msa_names"] = np.array(names3)
    out["fc_msa_seqs"] = np.array([msa3[n] for n in names3])
    out["fc_edges_u"] = np.array([e[0] for e in edges3])
    out["fc_edges_v"] = np.array([e[1] for e in edges3])
    out["fc_edges_t"] = np.array([e[2] for e in edges3], dtype=np.float64)
    out["fc_site_rates"] = np.array(ra, dtype=np.float64)
    out["fc_res"] = np.array(r3["res"], dtype=np.float64)
    print("fc", len(ch), "cherries; site rates", np.round(ra[:5], 4), "res", out["fc_res"].shape)
    np.savez_compressed(os.path.join(HERE, "siterm_learn.npz"), **out)
    print("wrote siterm_learn.npz")


if __name__ == "__main__":
    main()
