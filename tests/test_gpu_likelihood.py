"""Parity of the GPU held-out log-likelihood (cb_tree_likelihood, csrc/likelihood.hip.h) with the
reference's outputs (tests/golden/likelihood.npz, produced by cherryml/evaluation/_likelihood.py)
and with the oracle on seeded inputs.  Tolerance: 1e-9 relative on every per-site value (f64
arithmetic; the two sides differ in summation order and in the expm algorithm only)."""
import os

import numpy as np
import pytest

from test_oracle_golden import LIKELIHOOD_CASES, _chain_product, _likelihood_case, load_golden

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _models(z, model, pair):
    Q1, pi1 = z[model], z["pi_" + model]
    return Q1, pi1, (_chain_product(Q1) if pair else None), (np.kron(pi1, pi1) if pair else None)


@pytest.mark.parametrize("case,model,pair", LIKELIHOOD_CASES)
def test_likelihood_matches_reference_golden(case, model, pair):
    from cherryml_amd.evaluation import dp_likelihood_computation
    z = load_golden("likelihood.npz")
    tree, msa, cm, rates = _likelihood_case(z, case)
    aa = [str(a) for a in z["amino_acids"]]
    Q1, pi1, Q2, pi2 = _models(z, model, pair)
    ll, lls = dp_likelihood_computation(tree, msa, cm, rates, aa, pi1, Q1, reversible_1=True, pi_2=pi2, Q_2=Q2,
                                        reversible_2=True)
    assert np.allclose(lls, z[case + "_lls_rev"], rtol=RTOL, atol=1e-12)
    assert abs(ll - float(z[case + "_ll_rev"])) <= RTOL * abs(ll)
    if case + "_published" in z:
        pub = z[case + "_published"]
        assert np.allclose(ll if pub.ndim == 0 else lls, pub, atol=1e-4)


@pytest.mark.parametrize("case,model", [(c, m) for c, m, p in LIKELIHOOD_CASES if not p])
def test_likelihood_non_reversible_expm_matches_reference(case, model):
    """reversible_1 = False: the general scaling-and-squaring bank (reference: torch.matrix_exp)."""
    from cherryml_amd.evaluation import dp_likelihood_computation
    z = load_golden("likelihood.npz")
    tree, msa, cm, rates = _likelihood_case(z, case)
    aa = [str(a) for a in z["amino_acids"]]
    Q1, pi1, _, _ = _models(z, model, False)
    ll, lls = dp_likelihood_computation(tree, msa, cm, rates, aa, pi1, Q1, reversible_1=False)
    key = case + ("_lls_gen" if case + "_lls_gen" in z else "_lls_rev")   # demo_single: reversible run only
    assert np.allclose(lls, z[key], rtol=RTOL, atol=1e-12)


def _random_tree(rng, n_leaves, max_children=3):
    from cherryml_amd.io import Tree
    tree = Tree()
    names = [f"leaf{i}" for i in range(n_leaves)]
    tree.add_nodes(names)
    pool, k = list(names), 0
    while len(pool) > 1:
        take = min(len(pool), int(rng.integers(2, max_children + 1)))
        idx = sorted(rng.choice(len(pool), size=take, replace=False), reverse=True)
        kids = [pool.pop(i) for i in idx]
        v = f"int{k}"
        k += 1
        tree.add_node(v)
        for c in kids:
            tree.add_edge(v, c, float(rng.choice([0.0, rng.uniform(0.005, 2.5)], p=[0.05, 0.95])))
        pool.append(v)
    return tree, names


def _random_msa(rng, names, L, aa, gap=0.1):
    alphabet = np.array(list(aa) + ["-"])
    p = np.r_[np.full(len(aa), (1 - gap) / len(aa)), gap]
    return {n: "".join(rng.choice(alphabet, size=L, p=p)) for n in names}


@pytest.mark.parametrize("n_leaves,L,n_rates", [(2, 5, 1), (40, 70, 6), (150, 33, 33)])
def test_likelihood_single_sites_match_oracle(n_leaves, L, n_rates):
    from cherryml_amd.evaluation import dp_likelihood_computation
    from oracle import likelihood_oracle as lo
    z = load_golden("likelihood.npz")
    aa = [str(a) for a in z["amino_acids"]]
    rng = np.random.default_rng(100 + n_leaves)
    tree, names = _random_tree(rng, n_leaves)
    msa = _random_msa(rng, names, L, aa)
    rates = list(rng.choice(np.round(rng.uniform(0.05, 4.0, n_rates), 3), size=L))
    ll_o, lls_o = lo.log_likelihood(tree, msa, None, rates, aa, z["pi_lg"], z["lg"])
    ll, lls = dp_likelihood_computation(tree, msa, None, rates, aa, z["pi_lg"], z["lg"])
    assert np.allclose(lls, lls_o, rtol=RTOL, atol=1e-12)
    assert abs(ll - ll_o) <= RTOL * abs(ll_o)


@pytest.mark.parametrize("n_leaves,n_pairs,extra", [(6, 1, 0), (12, 19, 5), (30, 40, 3)])
def test_likelihood_pairs_match_oracle(n_leaves, n_pairs, extra):
    """400-state MFMA path: blocks of 16 pairs with a ragged tail, partly observed pairs, independent sites
    mixed in; the pair model is a random reversible 400-state Q (not a product chain)."""
    from cherryml_amd.evaluation import dp_likelihood_computation
    from oracle import likelihood_oracle as lo
    z = load_golden("likelihood.npz")
    aa = [str(a) for a in z["amino_acids"]]
    rng = np.random.default_rng(7 + n_pairs)
    tree, names = _random_tree(rng, n_leaves)
    L = 2 * n_pairs + extra
    msa = _random_msa(rng, names, L, aa, gap=0.2)
    perm = rng.permutation(L)
    cm = np.zeros((L, L), dtype=int)
    for k in range(n_pairs):
        i, j = perm[2 * k], perm[2 * k + 1]
        cm[i, j] = cm[j, i] = 1
    pi2 = rng.dirichlet(np.full(400, 5.0))
    sym = rng.uniform(0.0, 1.0, (400, 400)) * (rng.uniform(size=(400, 400)) < 0.3)
    sym = np.triu(sym, 1) + np.triu(sym, 1).T
    Q2 = sym * pi2[None, :]
    Q2[np.diag_indices(400)] = -Q2.sum(1)
    Q2 /= -(pi2 * np.diag(Q2)).sum()
    rates = list(rng.uniform(0.2, 3.0, L))
    ll_o, lls_o = lo.log_likelihood(tree, msa, cm, rates, aa, z["pi_wag"], z["wag"], pi2, Q2)
    ll, lls = dp_likelihood_computation(tree, msa, cm, rates, aa, z["pi_wag"], z["wag"], pi_2=pi2, Q_2=Q2)
    assert np.allclose(lls, lls_o, rtol=RTOL, atol=1e-12)
    assert abs(ll - ll_o) <= RTOL * abs(ll_o)


def _random_pair_model(rng):
    pi2 = rng.dirichlet(np.full(400, 5.0))
    sym = rng.uniform(0.0, 1.0, (400, 400)) * (rng.uniform(size=(400, 400)) < 0.3)
    sym = np.triu(sym, 1) + np.triu(sym, 1).T
    Q2 = sym * pi2[None, :]
    Q2[np.diag_indices(400)] = -Q2.sum(1)
    Q2 /= -(pi2 * np.diag(Q2)).sum()
    return pi2, Q2


@pytest.mark.parametrize("reversible_2", [True, False])
def test_likelihood_batch_of_families_equals_family_by_family(reversible_2):
    """cb_tree_likelihood_batch: ragged families (2 .. 60 leaves, 0 .. 23 pairs, one family without pairs,
    one without independent sites) in two GPU calls; every per-site value equals the single-family call's
    bit for bit (same kernels, shared eigendecomposition) and the oracle's to 1e-9."""
    from cherryml_amd.evaluation import dp_likelihood_computation, dp_likelihood_computation_batch
    from oracle import likelihood_oracle as lo
    z = load_golden("likelihood.npz")
    aa = [str(a) for a in z["amino_acids"]]
    rng = np.random.default_rng(2024)
    pi2, Q2 = _random_pair_model(rng)
    trees, msas, cms, rates = [], [], [], []
    for n_leaves, n_pairs, extra in [(9, 5, 3), (2, 0, 7), (60, 23, 1), (17, 4, 0), (33, 16, 20)]:
        tree, names = _random_tree(rng, n_leaves)
        L = 2 * n_pairs + extra
        cm = np.zeros((L, L), dtype=int)
        perm = rng.permutation(L)
        for k in range(n_pairs):
            i, j = perm[2 * k], perm[2 * k + 1]
            cm[i, j] = cm[j, i] = 1
        trees.append(tree), msas.append(_random_msa(rng, names, L, aa, gap=0.15)), cms.append(cm)
        rates.append(list(rng.choice(np.round(rng.uniform(0.1, 3.0, 4), 3), size=L)))
    profile = {}
    got = dp_likelihood_computation_batch(trees, msas, cms, rates, aa, z["pi_wag"], z["wag"], True, pi2, Q2, reversible_2,
                                          profile=profile)
    assert len(got) == 5 and profile["kernel_ms"] > 0
    for f in range(5):
        one = dp_likelihood_computation(trees[f], msas[f], cms[f], rates[f], aa, z["pi_wag"], z["wag"], pi_2=pi2, Q_2=Q2,
                                        reversible_2=reversible_2)
        assert np.array_equal(np.array(got[f][1]), np.array(one[1]), equal_nan=True), f
        if f in (0, 3):
            # family 3 has two leaves at distance 0 from their parent in different states: probability 0, and the
            # reference's log-space pruning turns that into NaN (-inf - -inf, _likelihood.py:238-296) -- so do we
            ll_o, lls_o = lo.log_likelihood(trees[f], msas[f], cms[f], rates[f], aa, z["pi_wag"], z["wag"], pi2, Q2)
            assert np.isnan(lls_o).all() == (f == 3)
            assert np.allclose(got[f][1], lls_o, rtol=RTOL, atol=1e-12, equal_nan=True)
            assert np.isnan(ll_o) if f == 3 else abs(got[f][0] - ll_o) <= RTOL * abs(ll_o)


def test_resident_models_give_the_per_call_results_family_after_family():
    """`LikelihoodModel` (cb_tl_model_create / cb_tl_model_run): the two models made once, ragged families evaluated one after
    the other on them -- larger families after smaller ones (the transition bank's buffer and the expm handle grow), smaller after
    larger (they stay) -- every per-site value equal to the per-call entry's bit for bit; a model refuses the other unit kind."""
    from cherryml_amd.evaluation import LikelihoodModel, dp_likelihood_computation
    z = load_golden("likelihood.npz")
    aa = [str(a) for a in z["amino_acids"]]
    rng = np.random.default_rng(77)
    pi2, Q2 = _random_pair_model(rng)
    with LikelihoodModel(z["wag"], z["pi_wag"], pairs=False) as m1, LikelihoodModel(Q2, pi2, pairs=True, alphabet_size=20) as m2:
        for n_leaves, n_pairs, extra in [(9, 5, 3), (40, 12, 9), (4, 1, 0), (60, 23, 1), (17, 4, 6)]:
            tree, names = _random_tree(rng, n_leaves)
            L = 2 * n_pairs + extra
            cm = np.zeros((L, L), dtype=int)
            perm = rng.permutation(L)
            for k in range(n_pairs):
                i, j = perm[2 * k], perm[2 * k + 1]
                cm[i, j] = cm[j, i] = 1
            msa = _random_msa(rng, names, L, aa, gap=0.1)
            rates = list(rng.choice(np.round(rng.uniform(0.1, 3.0, 4), 3), size=L))
            prof = {}
            got = dp_likelihood_computation(tree, msa, cm, rates, aa, z["pi_wag"], z["wag"], pi_2=pi2, Q_2=Q2, profile=prof,
                                            model_1=m1, model_2=m2)
            one = dp_likelihood_computation(tree, msa, cm, rates, aa, z["pi_wag"], z["wag"], pi_2=pi2, Q_2=Q2)
            assert np.array_equal(np.array(got[1]), np.array(one[1]), equal_nan=True), (n_leaves, n_pairs, extra)
            assert prof["kernel_ms"] > 0
        tree, names = _random_tree(rng, 5)
        with pytest.raises(ValueError):
            m1.log_likelihoods([tree], [np.zeros((9, 2), dtype=np.int8)], [np.zeros((9, 2), dtype=np.int8)], [np.ones(2)])


def test_likelihood_small_alphabet_pairs_and_general_S():
    """S1 = 4 (16 pair states, lane-group kernel with pair observations) and S1 = 9 (81 states, MFMA kernel,
    S not a multiple of 4 or 16)."""
    from cherryml_amd.evaluation import dp_likelihood_computation
    from oracle import likelihood_oracle as lo
    rng = np.random.default_rng(5)
    for S1 in (4, 9):
        aa = list("ACGTDEFHI"[:S1])
        tree, names = _random_tree(rng, 14)
        L = 11
        msa = _random_msa(rng, names, L, aa, gap=0.15)
        cm = np.zeros((L, L), dtype=int)
        for i, j in [(0, 7), (2, 3), (9, 5)]:
            cm[i, j] = cm[j, i] = 1

        def rev_model(S):
            pi = rng.dirichlet(np.full(S, 4.0))
            sym = rng.uniform(0.1, 1.0, (S, S))
            sym = np.triu(sym, 1) + np.triu(sym, 1).T
            Q = sym * pi[None, :]
            Q[np.diag_indices(S)] = -Q.sum(1)
            return pi, Q / -(pi * np.diag(Q)).sum()
        pi1, Q1 = rev_model(S1)
        pi2, Q2 = rev_model(S1 * S1)
        rates = list(rng.uniform(0.3, 2.0, L))
        ll_o, lls_o = lo.log_likelihood(tree, msa, cm, rates, aa, pi1, Q1, pi2, Q2)
        ll, lls = dp_likelihood_computation(tree, msa, cm, rates, aa, pi1, Q1, pi_2=pi2, Q_2=Q2)
        assert np.allclose(lls, lls_o, rtol=RTOL, atol=1e-12), S1


def test_likelihood_single_site_model_with_100_states():
    """S > 64 WITHOUT pairs (S1 = 0): a 100-letter alphabet, one rate category -- the leaf gather's single-site form
    (observed: one column of P_v; gap: the row sum) and the MFMA kernel with 20 units (ragged 32-block)."""
    from cherryml_amd.evaluation import dp_likelihood_computation
    from oracle import likelihood_oracle as lo
    rng = np.random.default_rng(77)
    S = 100
    aa = [chr(48 + i) for i in range(S)]
    assert "-" not in aa
    tree, names = _random_tree(rng, 21)
    L = 20
    msa = _random_msa(rng, names, L, aa, gap=0.2)
    pi = rng.dirichlet(np.full(S, 4.0))
    sym = rng.uniform(0.1, 1.0, (S, S))
    sym = np.triu(sym, 1) + np.triu(sym, 1).T
    Q = sym * pi[None, :]
    Q[np.diag_indices(S)] = -Q.sum(1)
    Q /= -(pi * np.diag(Q)).sum()
    rates = [1.0] * L
    ll_o, lls_o = lo.log_likelihood(tree, msa, None, rates, aa, pi, Q)
    ll, lls = dp_likelihood_computation(tree, msa, None, rates, aa, pi, Q)
    assert np.allclose(lls, lls_o, rtol=RTOL, atol=1e-12)
    with pytest.raises(NotImplementedError):   # more than one rate category at S > 64
        dp_likelihood_computation(tree, msa, None, [1.0] * (L - 1) + [2.0], aa, pi, Q)


@pytest.mark.parametrize("S", [40, 64])
def test_likelihood_33_to_64_states_with_several_rate_categories(S):
    """33 <= S <= 64 keeps several rate categories AND takes the shared counts-free bank handle of the large path: the
    handle's capacity and the uploaded branch lengths are [category][node] (they were [node], so every
    category but the first read an unwritten bank).  Single family and a ragged batch, against the oracle."""
    from cherryml_amd.evaluation import dp_likelihood_computation, dp_likelihood_computation_batch
    from oracle import likelihood_oracle as lo
    rng = np.random.default_rng(4000 + S)
    aa = [chr(48 + i) for i in range(S)]
    assert "-" not in aa
    pi = rng.dirichlet(np.full(S, 4.0))
    sym = rng.uniform(0.1, 1.0, (S, S))
    sym = np.triu(sym, 1) + np.triu(sym, 1).T
    Q = sym * pi[None, :]
    Q[np.diag_indices(S)] = -Q.sum(1)
    Q /= -(pi * np.diag(Q)).sum()
    trees, msas, rates = [], [], []
    for n_leaves, L, n_rates in [(13, 11, 5), (4, 7, 1), (31, 23, 3)]:
        tree, names = _random_tree(rng, n_leaves)
        trees.append(tree), msas.append(_random_msa(rng, names, L, aa, gap=0.15))
        rates.append(list(rng.choice(np.round(rng.uniform(0.1, 3.5, n_rates), 3), size=L)))
    want = [lo.log_likelihood(trees[f], msas[f], None, rates[f], aa, pi, Q) for f in range(3)]
    for reversible in (True, False):
        got = dp_likelihood_computation_batch(trees, msas, [None] * 3, rates, aa, pi, Q, reversible)
        for f in range(3):
            assert np.allclose(got[f][1], want[f][1], rtol=RTOL, atol=1e-12), (reversible, f)
            assert abs(got[f][0] - want[f][0]) <= RTOL * abs(want[f][0])
            one = dp_likelihood_computation(trees[f], msas[f], None, rates[f], aa, pi, Q, reversible_1=reversible)
            assert np.array_equal(np.array(one[1]), np.array(got[f][1])), (reversible, f)


def test_likelihood_errors():
    from cherryml_amd.evaluation import dp_likelihood_computation, tree_likelihood
    z = load_golden("likelihood.npz")
    tree, msa, cm, rates = _likelihood_case(z, "wag4_gaps")
    aa = [str(a) for a in z["amino_acids"]]
    with pytest.raises(Exception, match="Each site can only be in contact with one other site"):
        bad = np.ones((3, 3), dtype=int)
        t3, m3, _, _ = _likelihood_case(z, "wag3")
        dp_likelihood_computation(t3, {k: v * 3 for k, v in m3.items()}, bad, [1.0] * 3, aa, z["pi_wag"], z["wag"],
                                  pi_2=np.kron(z["pi_wag"], z["pi_wag"]), Q_2=_chain_product(z["wag"]))
    codes = np.full((tree.num_nodes(), 2), -1, dtype=np.int8)
    # the general (scaling-and-squaring) transition bank covers the 400-state pair model too
    Q2 = _chain_product(z["wag"])
    ll_gen = tree_likelihood(tree, codes, codes, Q2, np.kron(z["pi_wag"], z["pi_wag"]), [1.0, 1.0], reversible=False)
    ll_rev = tree_likelihood(tree, codes, codes, Q2, np.kron(z["pi_wag"], z["pi_wag"]), [1.0, 1.0], reversible=True)
    assert np.allclose(ll_gen, ll_rev, rtol=1e-9, atol=1e-9)
    with pytest.raises(NotImplementedError, match="one rate category"):
        tree_likelihood(tree, codes, codes, _chain_product(z["wag"]), np.kron(z["pi_wag"], z["pi_wag"]), [1.0, 2.0])


def test_compute_log_likelihoods_stage(tmp_path):
    """The stage writes the reference's files: <family>.txt = total, "<n> sites", per-site values."""
    from cherryml_amd.evaluation import compute_log_likelihoods
    from cherryml_amd.io import write_probability_distribution, write_rate_matrix, write_tree
    z = load_golden("likelihood.npz")
    aa = [str(a) for a in z["amino_acids"]]
    tree, msa, cm, rates = _likelihood_case(z, "rand_pair")
    d = {k: tmp_path / k for k in ("trees", "msas", "rates", "cms", "out")}
    for p in d.values():
        os.makedirs(p, exist_ok=True)
    write_tree(tree, str(d["trees"] / "fam.txt"))
    (d["msas"] / "fam.txt").write_text("".join(f">{k}\n{v}\n" for k, v in msa.items()))
    (d["rates"] / "fam.txt").write_text(f"{len(rates)} sites\n" + " ".join(map(str, rates)))
    n = cm.shape[0]
    (d["cms"] / "fam.txt").write_text(f"{n} sites\n" + "\n".join("".join(str(int(x)) for x in row) for row in cm) + "\n")
    pairs = [a + b for a in aa for b in aa]
    write_rate_matrix(z["wag"], aa, str(tmp_path / "Q1.txt"))
    write_probability_distribution(z["pi_wag"], aa, str(tmp_path / "pi1.txt"))
    write_rate_matrix(_chain_product(z["wag"]), pairs, str(tmp_path / "Q2.txt"))
    write_probability_distribution(np.kron(z["pi_wag"], z["pi_wag"]), pairs, str(tmp_path / "pi2.txt"))
    compute_log_likelihoods(str(d["trees"]), str(d["msas"]), str(d["rates"]), str(d["cms"]), ["fam"], aa,
                            str(tmp_path / "pi1.txt"), str(tmp_path / "Q1.txt"), True, "cpu", str(tmp_path / "pi2.txt"),
                            str(tmp_path / "Q2.txt"), True, "cpu", str(d["out"]))
    lines = (d["out"] / "fam.txt").read_text().split("\n")
    assert lines[1] == f"{len(rates)} sites"
    lls = np.array([float(x) for x in lines[2].split(" ")])
    assert np.allclose(lls, z["rand_pair_lls_rev"], rtol=RTOL, atol=1e-12)
    assert abs(float(lines[0]) - float(z["rand_pair_ll_rev"])) <= RTOL * abs(float(lines[0]))
    assert (d["out"] / "fam.profiling").exists() and (d["out"] / "profiling_0.txt").exists()
