"""SiteRM count / pseudocount assembly on the GPU (cb_siterm_assemble) and the per-family estimator
driver, against vectors produced by running the reference (tests/golden/siterm_assembly.npz) and
against the oracle.  Needs an MI355X: `pytest -m gpu`."""
import numpy as np
import pytest

from conftest import load_golden, relerr
from test_oracle_golden import ASSEMBLY_CASES, _assembly_case

pytestmark = pytest.mark.gpu


def _inputs(z, c):
    tree, msa, raw = _assembly_case(z, c)
    return dict(tree=tree, msa=msa, raw=raw, grid=sorted(z[c + "_grid"].tolist()),
                alphabet=[str(a) for a in z[c + "_alphabet"]], strategy=str(z[c + "_strategy"]),
                reverse=bool(z[c + "_reverse"]), rates=z[c + "_site_rates"], Q0=z[c + "_Q0"],
                lam=float(z[c + "_lambda"]), epochs=int(z[c + "_epochs"]))


@pytest.mark.parametrize("case", ASSEMBLY_CASES)
def test_raw_counts_bit_exact(case):
    from cherryml_amd._siterm import get_cherry_transitions, get_edge_transitions, get_raw_count_matrices
    z = load_golden("siterm_assembly.npz")
    a = _inputs(z, case)
    tr = (get_cherry_transitions(a["tree"], a["msa"]) if a["strategy"] == "cherry++"
          else get_edge_transitions(a["tree"], a["msa"]))
    assert [(x, y) for x, y, _ in tr] == [(str(x), str(y)) for x, y in zip(z[case + "_tr_a"], z[case + "_tr_b"])]
    assert np.array_equal(np.array([t for _, _, t in tr]), z[case + "_tr_t"])
    raw = get_raw_count_matrices(tr, a["grid"], a["alphabet"], a["reverse"])
    assert np.array_equal(raw, a["raw"])


@pytest.mark.parametrize("case", ASSEMBLY_CASES)
def test_mixed_counts_bit_exact_and_prior(case):
    """The full assembly (raw -> pseudocounts -> lambda mix) equals what the reference hands to its
    optimiser, bit for bit, once compactified the reference's way."""
    from cherryml_amd._siterm import get_count_prior_probability_matrices
    from cherryml_amd._siterm._assembly import _assemble, _pairs_and_codes
    from oracle import siterm_assembly_oracle as sa
    z = load_golden("siterm_assembly.npz")
    a = _inputs(z, case)
    prior = get_count_prior_probability_matrices(a["Q0"], a["grid"])
    assert np.allclose(prior, z[case + "_prior"], rtol=1e-12, atol=1e-15)
    pairs, codes = _pairs_and_codes(a["tree"], a["msa"], a["alphabet"], a["strategy"])
    mixed = _assemble(pairs, codes, a["grid"], a["rates"], z[case + "_prior"], a["lam"], a["reverse"],
                      len(a["alphabet"]), 0, False)
    cc, tt, _ = sa.compactify(mixed, a["grid"], a["Q0"], a["rates"])
    assert np.array_equal(tt, z[case + "_times"])
    assert np.array_equal(cc, z[case + "_counts"])


@pytest.mark.parametrize("case", ASSEMBLY_CASES)
def test_estimator_matches_reference(case):
    from cherryml_amd._siterm import estimate_site_specific_rate_matrices_given_tree_and_site_rates as est
    z = load_golden("siterm_assembly.npz")
    a = _inputs(z, case)
    r = est(tree=a["tree"], site_rates=list(a["rates"]), msa=a["msa"], alphabet=a["alphabet"],
            regularization_strength=a["lam"], regularization_rate_matrix=a["Q0"], quantization_points=a["grid"],
            optimization_num_epochs=a["epochs"], transitions_strategy=a["strategy"],
            include_reverse_transitions=a["reverse"])
    assert r["res"].shape == z[case + "_res"].shape
    for l in range(r["res"].shape[0]):
        assert relerr(r["res"][l], z[case + "_res"][l]) < 1e-6, l


def test_all_gap_site_gets_the_prior_and_bad_arguments_raise():
    from cherryml_amd._siterm import estimate_site_specific_rate_matrices_given_tree_and_site_rates as est
    from cherryml_amd._siterm import get_raw_count_matrices
    z = load_golden("siterm_assembly.npz")
    a = _inputs(z, "t2_some_missing")
    alphabet = a["alphabet"][:-1]   # without the gap state: site 0 has no counts at all
    Q0 = np.full((6, 6), 0.2)
    np.fill_diagonal(Q0, -1.0)
    r = est(tree=a["tree"], site_rates=[2.0, 0.5], msa=a["msa"], alphabet=alphabet, regularization_strength=0.5,
            regularization_rate_matrix=Q0, quantization_points=a["grid"], optimization_num_epochs=5)
    assert np.array_equal(r["res"][0], Q0 * 2.0)
    assert not np.array_equal(r["res"][1], Q0 * 0.5)
    with pytest.raises(ValueError):
        est(tree=a["tree"], site_rates=[2.0, 0.5], msa=a["msa"], alphabet=alphabet, regularization_strength=0.5,
            regularization_rate_matrix=Q0, quantization_points=a["grid"], optimization_num_epochs=5,
            transitions_strategy="nope")
    # the reference's default spelling "cpu" is accepted (same result: there is one execution target, the GPU)
    r_cpu = est(tree=a["tree"], site_rates=[2.0, 0.5], msa=a["msa"], alphabet=alphabet, regularization_strength=0.5,
                regularization_rate_matrix=Q0, quantization_points=a["grid"], optimization_num_epochs=5,
                vectorized_cherryml_implementation_device="cpu")
    assert np.array_equal(r_cpu["res"], r["res"])
    with pytest.raises(ValueError):
        est(tree=a["tree"], site_rates=[2.0, 0.5], msa=a["msa"], alphabet=alphabet, regularization_strength=0.5,
            regularization_rate_matrix=Q0, quantization_points=a["grid"], optimization_num_epochs=5,
            vectorized_cherryml_implementation_device="tpu")
    with pytest.raises(ValueError):   # unsorted grid is refused by the C ABI
        get_raw_count_matrices([("AD", "DA", 0.1)], [0.2, 0.1], ["A", "D"])


@pytest.mark.parametrize("case", ["t2_eq_cherry", "t2_some_missing", "rand_cherry"])
def test_per_site_dispatch_matches_reference(case):
    """`use_vectorized_cherryml_implementation=False`: the reference's per-site loop
    (_site_specific_rate_matrix.py:43-84, 659-684: RateMatrixLearner per site, "pande_reversible",
    initialisation Q0 * rate_l) -- here one batched device loop (cb_train_pande_reversible with L > 1).
    Golden: tests/golden/make_golden_siterm_persite.py (the reference per site, as is in float32 and
    through its float64 recipe).  Stated tolerance 1e-6 to the float64 run."""
    from cherryml_amd._siterm import estimate_site_specific_rate_matrices_given_tree_and_site_rates as est
    z = load_golden("siterm_assembly.npz")
    g = load_golden("siterm_persite.npz")
    a = _inputs(z, case)
    r = est(tree=a["tree"], site_rates=list(a["rates"]), msa=a["msa"], alphabet=a["alphabet"],
            regularization_strength=a["lam"], regularization_rate_matrix=a["Q0"], quantization_points=a["grid"],
            optimization_num_epochs=a["epochs"], transitions_strategy=a["strategy"],
            include_reverse_transitions=a["reverse"], use_vectorized_cherryml_implementation=False)
    want64, want32 = g[case + "_res_f64"], g[case + "_res_f32"]
    assert r["res"].shape == want64.shape
    e64 = [relerr(r["res"][l], want64[l]) for l in range(len(want64))]
    e32 = [relerr(r["res"][l], want32[l]) for l in range(len(want64))]
    print(f"{case}: per-site path, max rel. Frobenius to the f64 reference {max(e64):.2e}, to its float32 run {max(e32):.2e}")
    assert max(e64) < 1e-6
    assert max(e32) < 5e-3      # (the float32 run itself sits up to 1.3e-3 from the float64 one at the extreme site rates)
    # and it is a different estimator from the vectorised one (other parameterisation, same optimum family)
    rv = est(tree=a["tree"], site_rates=list(a["rates"]), msa=a["msa"], alphabet=a["alphabet"],
             regularization_strength=a["lam"], regularization_rate_matrix=a["Q0"], quantization_points=a["grid"],
             optimization_num_epochs=a["epochs"], transitions_strategy=a["strategy"],
             include_reverse_transitions=a["reverse"], use_vectorized_cherryml_implementation=True)
    assert rv["res"].shape == r["res"].shape


def test_batch_of_families_equals_family_by_family():
    """cb_siterm_assemble_batch + one bank over all families' sites: the count tensor is bit-identical to the
    families' own tensors stacked, and every family's rate matrices equal the single-family estimator's
    (sites are independent; the families here have different trees, site counts and gap patterns)."""
    from cherryml_amd._siterm import estimate_site_specific_rate_matrices_given_tree_and_site_rates as est
    from cherryml_amd._siterm import estimate_site_specific_rate_matrices_given_trees_and_site_rates as est_batch
    from cherryml_amd._siterm._assembly import _assemble, _assemble_batch, _pairs_and_codes
    z = load_golden("siterm_assembly.npz")
    cases = [c for c in ASSEMBLY_CASES if str(z[c + "_strategy"]) == "cherry++"]
    a0 = _inputs(z, cases[0])
    same = [c for c in cases if [str(x) for x in z[c + "_alphabet"]] == a0["alphabet"]
            and sorted(z[c + "_grid"].tolist()) == a0["grid"] and np.array_equal(z[c + "_Q0"], a0["Q0"])]
    assert len(same) >= 2, same
    fams = [_inputs(z, c) for c in same]
    _run_batch_vs_single(z, same[0], fams)
    # a 20-state family, the same family without its last three sites, and with its sites reversed
    r = _inputs(z, "rand_cherry")
    cut = dict(r, msa={k: v[:-3] for k, v in r["msa"].items()}, rates=r["rates"][:-3])
    rev = dict(r, msa={k: v[::-1] for k, v in r["msa"].items()}, rates=r["rates"][::-1].copy())
    _run_batch_vs_single(z, "rand_cherry", [r, cut, rev])


def _run_batch_vs_single(z, case0, fams):
    from cherryml_amd._siterm import estimate_site_specific_rate_matrices_given_tree_and_site_rates as est
    from cherryml_amd._siterm import estimate_site_specific_rate_matrices_given_trees_and_site_rates as est_batch
    from cherryml_amd._siterm._assembly import _assemble, _assemble_batch, _pairs_and_codes
    a0 = fams[0]
    same = [case0]
    S = len(a0["alphabet"])
    pc = [_pairs_and_codes(f["tree"], f["msa"], a0["alphabet"], "cherry++") for f in fams]
    prior = z[same[0] + "_prior"]
    stacked = np.concatenate([_assemble(p, c, a0["grid"], f["rates"], prior, a0["lam"], True, S, 0, False)
                              for (p, c), f in zip(pc, fams)])
    batch = _assemble_batch([p for p, _ in pc], [c for _, c in pc], a0["grid"], [f["rates"] for f in fams], prior,
                            a0["lam"], True, S, 0, False)
    assert np.array_equal(batch, stacked)
    kw = dict(alphabet=a0["alphabet"], regularization_strength=a0["lam"], regularization_rate_matrix=a0["Q0"],
              quantization_points=a0["grid"], optimization_num_epochs=a0["epochs"])
    got = est_batch([f["tree"] for f in fams], [list(f["rates"]) for f in fams], [f["msa"] for f in fams], **kw)
    for f, g in zip(fams, got):
        one = est(tree=f["tree"], site_rates=list(f["rates"]), msa=f["msa"], **kw)
        assert g["res"].shape == one["res"].shape
        # (not bit for bit: how a site's buckets are split over workgroups depends on the number of sites in the bank)
        assert max(relerr(g["res"][l], one["res"][l]) for l in range(len(one["res"]))) < 1e-8
