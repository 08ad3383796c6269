#!/usr/bin/env python3
"""Golden vectors for the held-out log-likelihood DP (cherryml/evaluation/_likelihood.py:47-327,
`dp_likelihood_computation`), made by RUNNING THE REFERENCE in the build container.

Cases (inputs + the reference's outputs `ll`, `lls`; reversible and non-reversible expm where cheap):
  wag3, wag4, wag4_gaps   the reference tests' own small cases (tests/evaluation_tests/likelihood_test.py:
                          242-378), whose published values -7.343870, -10.091868, [-10.092142, -7.344207]
                          ("manually verified with FastTree") are stored as `*_published`
  wagxwag3                pair-site model WAG x WAG on 3 sequences (:546-592), published -7.343870 + -9.714873
  rand_single             random 25-leaf multifurcating tree, 40 sites, 4 rate categories, gaps, LG
  rand_pair               random 9-leaf tree, 12 sites of which 4 contacting pairs, partial gaps, WAG x WAG + WAG
  demo_single             demo_data family 1a92_1_A with its tree and site rates, LG
Writes tests/golden/likelihood.npz."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _prepare_scratch  # noqa: E402


def main():
    _prepare_scratch()
    import cherryml
    from cherryml.evaluation._likelihood import dp_likelihood_computation
    from cherryml.io import Tree, read_msa, read_rate_matrix, read_site_rates, read_tree
    from cherryml.markov_chain import (FactorizedReversibleModel, chain_product, compute_stationary_distribution,
                                       wag_matrix, wag_stationary_distribution)
    aa = cherryml.utils.amino_acids
    wag = wag_matrix().to_numpy()
    pi_wag = wag_stationary_distribution().to_numpy().reshape(-1)
    lg = read_rate_matrix("data/rate_matrices/lg.txt").to_numpy()
    pi_lg = compute_stationary_distribution(lg)
    wxw = chain_product(wag, wag)
    pi_wxw = compute_stationary_distribution(wxw)
    out = dict(amino_acids=np.array(aa), wag=wag, pi_wag=pi_wag, lg=lg, pi_lg=pi_lg)

    def run(case, tree, msa, cm, rates, pi1, Q1, pi2, Q2, revs=(True, False)):
        edges = tree.edges()
        names = list(msa.keys())
        out.update({f"{case}_nodes": np.array(tree.nodes()), f"{case}_eu": np.array([u for u, _, _ in edges]),
                    f"{case}_ev": np.array([v for _, v, _ in edges]), f"{case}_et": np.array([t for _, _, t in edges]),
                    f"{case}_names": np.array(names), f"{case}_seqs": np.array([msa[k] for k in names]),
                    f"{case}_contact_map": np.array(cm if cm is not None else np.zeros((0, 0))),
                    f"{case}_has_cm": np.bool_(cm is not None), f"{case}_rates": np.array(rates, dtype=np.float64)})
        for rev in revs:
            f1 = FactorizedReversibleModel(Q1) if rev else None
            f2 = FactorizedReversibleModel(Q2) if (rev and Q2 is not None) else None
            ll, lls = dp_likelihood_computation(
                tree=tree, msa=msa, contact_map=cm, site_rates=list(rates), amino_acids=aa, pi_1=pi1, Q_1=Q1,
                fact_1=f1, reversible_1=rev, device_1="cpu", pi_2=pi2, Q_2=Q2, fact_2=f2, reversible_2=rev,
                device_2="cpu" if Q2 is not None else None, output_profiling_path=os.devnull)
            tag = "rev" if rev else "gen"
            out[f"{case}_ll_{tag}"] = np.float64(ll)
            out[f"{case}_lls_{tag}"] = np.array(lls, dtype=np.float64)
            print(case, tag, ll)

    def tree_of(nodes, edges):
        t = Tree()
        t.add_nodes(nodes)
        t.add_edges(edges)
        return t
    t3 = tree_of(["r", "l1", "l2", "l3"], [("r", "l1", 0.0), ("r", "l2", 1.120547166), ("r", "l3", 3.402392896)])
    run("wag3", t3, {"l1": "S", "l2": "T", "l3": "G"}, None, [1.0], pi_wag, wag, None, None)
    out["wag3_published"] = np.float64(-7.343870)
    t4 = tree_of(["r", "i1", "l1", "l2", "l3", "l4"], [("r", "l1", 0.0), ("r", "l2", 1.121352212), ("r", "i1", 1.840784231),
                                                       ("i1", "l3", 1.870540996), ("i1", "l4", 2.678783814)])
    run("wag4", t4, {"l1": "S", "l2": "T", "l3": "G", "l4": "D"}, None, [1.0], pi_wag, wag, None, None)
    out["wag4_published"] = np.float64(-10.091868)
    t4g = tree_of(["r", "i1", "l1", "l2", "l3", "l4"], [("r", "l1", 0.0), ("r", "l2", 1.121562482), ("r", "i1", 1.719057732),
                                                        ("i1", "l3", 1.843908633), ("i1", "l4", 2.740236263)])
    run("wag4_gaps", t4g, {"l1": "SS", "l2": "TT", "l3": "GG", "l4": "D-"}, None, [1.0, 1.0], pi_wag, wag, None, None)
    out["wag4_gaps_published"] = np.array([-10.092142, -7.344207])
    run("wagxwag3", t3, {"l1": "SK", "l2": "TI", "l3": "GL"}, np.ones((2, 2)), [1.0, 1.0], pi_wag, wag, pi_wxw, wxw,
        revs=(True,))
    out["wagxwag3_published"] = np.float64(-7.343870 + -9.714873)

    rng = np.random.default_rng(21)

    def random_tree(n_leaves):
        t = Tree()
        t.add_node("r")
        frontier, nid = ["r"], 0
        while len(frontier) < n_leaves:
            u = frontier.pop(int(rng.integers(len(frontier))))
            for _ in range(int(rng.choice([2, 2, 3]))):
                v = f"n{nid}"
                nid += 1
                t.add_node(v)
                t.add_edge(u, v, float(np.round(rng.exponential(0.3) + 1e-3, 6)))
                frontier.append(v)
        return t
    t = random_tree(25)
    L = 40
    msa = {u: "".join(rng.choice(aa + ["-"], size=L, p=[0.046] * 20 + [0.08])) for u in t.leaves()}
    rates = list(rng.choice([0.3, 0.8, 1.4, 3.1], size=L))
    run("rand_single", t, msa, None, rates, pi_lg, lg, None, None)
    t = random_tree(9)
    L = 12
    msa = {u: "".join(rng.choice(aa + ["-"], size=L, p=[0.045] * 20 + [0.1])) for u in t.leaves()}
    cm = np.eye(L, dtype=int)
    for i, j in [(0, 7), (2, 9), (3, 11), (5, 6)]:
        cm[i, j] = cm[j, i] = 1
    run("rand_pair", t, msa, cm, list(rng.choice([0.5, 1.0, 2.0], size=L)), pi_wag, wag, pi_wxw, wxw, revs=(True,))
    demo = os.path.join(REF, "demo_data")
    fam = "1a92_1_A"
    run("demo_single", read_tree(os.path.join(demo, "trees", fam + ".txt")), read_msa(os.path.join(demo, "msas", fam + ".txt")),
        None, read_site_rates(os.path.join(demo, "site_rates", fam + ".txt")), pi_lg, lg, None, None, revs=(True,))
    np.savez_compressed(os.path.join(HERE, "likelihood.npz"), **out)
    print("wrote", os.path.join(HERE, "likelihood.npz"))


if __name__ == "__main__":
    main()
