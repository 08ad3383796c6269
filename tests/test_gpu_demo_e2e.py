"""BASELINE.json configs 1 and 3 on three real demo_data families: the whole pipelines of this package
(counting [GPU] -> JTT-IPW -> optimiser [GPU]) against what the REFERENCE pipelines produced on the same
files (tests/golden/demo_e2e.npz, made by tests/golden/make_golden_demo_e2e.py).
Stated tolerance: learned Q within 1e-6 relative Frobenius of the reference run in float64; the
distance to the reference's float32 result is reported and bounded by 1e-3.  Needs an MI355X."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu
AA = list("ARNDCQEGHILKMFPSTWYV")


def _materialise(tmp_path, z):
    dirs = {}
    for kind in ("msa", "tree", "site_rates", "contact_map"):
        d = tmp_path / kind
        d.mkdir()
        for fam, text in zip(z["families"], z[f"text_{kind}"]):
            (d / f"{fam}.txt").write_text(str(text))
        dirs[kind] = str(d)
    return dirs, [str(f) for f in z["families"]]


def test_lg_pipeline_on_demo_data(tmp_path):
    import cherryml_amd
    from cherryml_amd import caching
    from cherryml_amd.io import read_count_matrices_arrays, read_rate_matrix
    z = load_golden("demo_e2e.npz")
    dirs, fams = _materialise(tmp_path, z)
    caching.set_cache_dir(str(tmp_path / "cache"))
    try:
        res = cherryml_amd.lg_end_to_end_with_cherryml_optimizer(
            msa_dir=dirs["msa"], families=fams, tree_estimator=None, initial_tree_estimator_rate_matrix_path=None,
            num_epochs=int(z["lg_epochs"]), tree_dir=dirs["tree"], site_rates_dir=dirs["site_rates"])
    finally:
        caching.set_cache_dir(None)
    assert list(res["quantization_points"]) == [str(q) for q in z["quantization_points"]]
    q, C, states = read_count_matrices_arrays(os.path.join(res["count_matrices_dir_0"], "result.txt"))
    assert states == AA and np.array_equal(q, z["lg_t"])
    assert np.array_equal(C, z["lg_counts"])                                  # counting: bit-exact
    init = read_rate_matrix(os.path.join(res["jtt_ipw_dir_0"], "result.txt")).to_numpy()
    assert np.allclose(init, z["lg_init"], rtol=1e-12, atol=1e-15)            # JTT-IPW initialiser
    learned = read_rate_matrix(res["learned_rate_matrix_path"]).to_numpy()
    assert relerr(learned, z["lg_Q_best_f64"]) < 1e-6                         # the stated tolerance
    d32 = relerr(learned, z["lg_learned_f32"])
    print(f"LG demo_data: rel. Frobenius to the f64 reference {relerr(learned, z['lg_Q_best_f64']):.2e}, "
          f"to the reference's float32 result {d32:.2e}")
    assert d32 < 1e-3


def test_coevolution_pipeline_on_demo_data(tmp_path):
    import cherryml_amd
    from cherryml_amd import caching
    from cherryml_amd.io import read_count_matrices_arrays, read_rate_matrix
    z = load_golden("demo_e2e.npz")
    dirs, fams = _materialise(tmp_path, z)
    pairs = [a + b for a in AA for b in AA]
    mask = np.unpackbits(z["co_mask_packed"])[:160000].reshape(400, 400)
    mpath = str(tmp_path / "mask.txt")
    pd.DataFrame(mask, index=pairs, columns=pairs).to_csv(mpath, sep=" ")
    caching.set_cache_dir(str(tmp_path / "cache"))
    try:
        res = cherryml_amd.coevolution_end_to_end_with_cherryml_optimizer(
            msa_dir=dirs["msa"], contact_map_dir=dirs["contact_map"], minimum_distance_for_nontrivial_contact=7,
            coevolution_mask_path=mpath, families=fams, tree_estimator=None,
            initial_tree_estimator_rate_matrix_path=None, num_epochs=int(z["co_epochs"]), tree_dir=dirs["tree"])
    finally:
        caching.set_cache_dir(None)
    q, C, states = read_count_matrices_arrays(os.path.join(res["count_matrices_dir_0"], "result.txt"))
    ref = np.zeros_like(C)
    ref[tuple(z["co_counts_nz"].T)] = z["co_counts_val"]
    assert states == pairs and np.array_equal(q, z["co_t"]) and np.array_equal(C, ref)   # bit-exact
    init = read_rate_matrix(os.path.join(res["jtt_ipw_dir_0"], "result.txt")).to_numpy()
    assert np.allclose(init, z["co_init"], rtol=1e-11, atol=1e-15)
    learned = read_rate_matrix(res["learned_rate_matrix_path"]).to_numpy()
    assert relerr(learned, z["co_Q_best_f64"]) < 1e-6
    d32 = relerr(learned, z["co_learned_f32"])
    print(f"co-evolution demo_data: to the f64 reference {relerr(learned, z['co_Q_best_f64']):.2e}, "
          f"to the reference's float32 result {d32:.2e}")
    assert d32 < 1e-3


def test_all_32_demo_families_are_counted_exactly_like_the_reference(tmp_path):
    """BASELINE.json config 3 as it really is (parity margin c): ALL 32 demo_data families (the reference's own
    files, tests/golden/demo32_co_inputs.npz) through this package's maximal matching + `cb_count_co_transitions` against
    the counts the REFERENCE's pipeline produced from them (coevo_demo_full.npz: Python `count_co_transitions`): every one
    of the 730 864 non-zero bins and sum C = 1 057 194 bit for bit; the masked JTT-IPW initialiser (now from the device
    statistics pass) to 1e-11 on its support; and the resident chain counts the same pairs."""
    import cherryml_amd
    from cherryml_amd import caching
    from cherryml_amd.estimation_end_to_end import coevolution_fit_resident, create_maximal_matching_contact_map
    from cherryml_amd.io import read_count_matrices_arrays, read_rate_matrix
    zin = load_golden("demo32_co_inputs.npz")
    z = dict(np.load(os.path.join(os.path.dirname(__file__), "golden", "coevo_demo_full.npz")))
    fams = [str(f) for f in zin["families"]]
    assert fams == [str(f) for f in z["families"]] and len(fams) == 32
    dirs = {}
    for kind in ("msa", "tree", "contact_map"):
        d = tmp_path / kind
        d.mkdir()
        off, blob = zin[f"{kind}_offsets"], zin[f"{kind}_bytes"].tobytes()
        for k, fam in enumerate(fams):
            (d / f"{fam}.txt").write_bytes(blob[off[k]:off[k + 1]])
        dirs[kind] = str(d)
    pairs = [a + b for a in AA for b in AA]
    mask = np.unpackbits(z["mask_packed"])[:160000].reshape(400, 400)
    mpath = str(tmp_path / "mask.txt")
    pd.DataFrame(mask, index=pairs, columns=pairs).to_csv(mpath, sep=" ")
    caching.set_cache_dir(str(tmp_path / "cache"))
    try:
        res = cherryml_amd.coevolution_end_to_end_with_cherryml_optimizer(
            msa_dir=dirs["msa"], contact_map_dir=dirs["contact_map"], minimum_distance_for_nontrivial_contact=7,
            coevolution_mask_path=mpath, families=fams, tree_estimator=None,
            initial_tree_estimator_rate_matrix_path=None, num_epochs=1, tree_dir=dirs["tree"])
        cm_dir = create_maximal_matching_contact_map(
            i_contact_map_dir=dirs["contact_map"], families=fams, minimum_distance_for_nontrivial_contact=7,
            num_processes=1)["o_contact_map_dir"]
    finally:
        caching.set_cache_dir(None)
    q, C, states = read_count_matrices_arrays(os.path.join(res["count_matrices_dir_0"], "result.txt"))
    assert states == pairs and np.array_equal(q, z["t"])
    ref = np.zeros(tuple(z["C_shape"]))
    ref[z["C_b"].astype(np.int64), z["C_i"].astype(np.int64), z["C_j"].astype(np.int64)] = z["C_quarters"] * 0.25
    assert C.sum() == 1057194.0 and np.count_nonzero(C) == 730864
    assert np.array_equal(C, ref)                                                     # bit-exact, all 129 x 400 x 400 bins
    init = read_rate_matrix(os.path.join(res["jtt_ipw_dir_0"], "result.txt")).to_numpy()
    sup = mask.astype(bool) | np.eye(400, dtype=bool)   # (the golden keeps masked matrices on the mask's support + diagonal)
    assert np.all(init[~sup] == 0.0) and np.allclose(init[sup], z["init_support"], rtol=1e-11, atol=1e-15)
    r = coevolution_fit_resident(tree_dir=dirs["tree"], msa_dir=dirs["msa"], contact_map_dir=cm_dir, families=fams,
                                 amino_acids=AA, quantization_points=[float(x) for x in res["quantization_points"]],
                                 edge_or_cherry="cherry++", minimum_distance_for_nontrivial_contact=7,
                                 mask=mask.astype(np.float64), num_epochs=1)
    assert r["n_pairs"] == 1057194.0
    assert np.allclose(r["initialization"][sup], z["init_support"], rtol=1e-11, atol=1e-15)


def test_coevolution_resident_chain_equals_the_file_passing_pipeline(tmp_path):
    """count -> JTT-IPW -> optimise as ONE resident chain (estimation_end_to_end/_resident.py: device-resident counts, the
    initialiser from two reduced S x S sums, no 84 MB count file between the stages) against the same reference golden
    as the pipeline above: initialiser 1e-11, learned Q 1e-6 (stated tolerance), and the reference's float32 result."""
    import cherryml_amd
    from cherryml_amd import caching
    from cherryml_amd.estimation_end_to_end import coevolution_fit_resident, create_maximal_matching_contact_map
    z = load_golden("demo_e2e.npz")
    dirs, fams = _materialise(tmp_path, z)
    mask = np.unpackbits(z["co_mask_packed"])[:160000].reshape(400, 400).astype(np.float64)
    caching.set_cache_dir(str(tmp_path / "cache"))
    try:
        cm_dir = create_maximal_matching_contact_map(
            i_contact_map_dir=dirs["contact_map"], families=fams, minimum_distance_for_nontrivial_contact=7,
            num_processes=1)["o_contact_map_dir"]
    finally:
        caching.set_cache_dir(None)
    grid = [float(q) for q in z["quantization_points"]]
    r = coevolution_fit_resident(tree_dir=dirs["tree"], msa_dir=dirs["msa"], contact_map_dir=cm_dir, families=fams,
                                 amino_acids=AA, quantization_points=grid, edge_or_cherry="cherry++",
                                 minimum_distance_for_nontrivial_contact=7, mask=mask, num_epochs=int(z["co_epochs"]))
    assert r["n_pairs"] == float(z["co_counts_val"].sum())
    assert np.allclose(r["initialization"], z["co_init"], rtol=1e-11, atol=1e-15)
    assert relerr(r["Q_best"], z["co_Q_best_f64"]) < 1e-6
    assert relerr(r["Q_best"], z["co_learned_f32"]) < 1e-3


def test_public_api_lg_with_given_trees_and_with_fast_cherries(tmp_path):
    """`cherryml_public_api` (what `python -m cherryml` calls).  With the demo trees handed over it is the
    LG pipeline above (same golden, same tolerance).  With `tree_estimator_name="FastCherries"` and two
    iterations every stage runs on this package's own code (pairing, branch lengths / site rates,
    counting, JTT-IPW, optimiser; the second iteration re-estimates the trees under the matrix learned
    in the first): no reference run exists for that chain (its wrapper needs ete3 and FastTree), so the
    checks are structural -- a valid reversible rate matrix close to LG-like behaviour, the cache
    holding one tree / count / rate-matrix directory per iteration."""
    import cherryml_amd
    from cherryml_amd._cherryml_public_api import cherryml_public_api
    from cherryml_amd.io import read_rate_matrix, write_rate_matrix
    z = load_golden("demo_e2e.npz")
    dirs, fams = _materialise(tmp_path, z)
    out1 = str(tmp_path / "learned_given_trees.txt")
    prof = cherryml_public_api(output_path=out1, model_name="LG", msa_dir=dirs["msa"], tree_dir=dirs["tree"],
                               site_rates_dir=dirs["site_rates"], cache_dir=str(tmp_path / "cache1"),
                               num_epochs=int(z["lg_epochs"]), families=fams, tree_estimator_name="FastTree")
    assert "time_optimization" in prof
    assert relerr(read_rate_matrix(out1).to_numpy(), z["lg_Q_best_f64"]) < 1e-6
    with pytest.raises(NotImplementedError):      # FastTree itself is an external program: not built
        cherryml_public_api(output_path=out1, model_name="LG", msa_dir=dirs["msa"], cache_dir=str(tmp_path / "c"),
                            initial_tree_estimator_rate_matrix_path=out1, families=fams, tree_estimator_name="FastTree")
    with pytest.raises(ValueError):
        cherryml_public_api(output_path=out1, model_name="WAG", msa_dir=dirs["msa"])
    with pytest.raises(ValueError):
        cherryml_public_api(output_path=out1, model_name="co-evolution", msa_dir=dirs["msa"], num_iterations=2,
                            tree_dir=dirs["tree"], contact_map_dir=dirs["contact_map"], families=fams)
    # FastCherries, two iterations, starting from the matrix learned above
    out2 = str(tmp_path / "learned_fast_cherries.txt")
    cache2 = tmp_path / "cache2"
    cherryml_public_api(output_path=out2, model_name="LG", msa_dir=dirs["msa"], cache_dir=str(cache2),
                        initial_tree_estimator_rate_matrix_path=out1, num_iterations=2, num_epochs=40,
                        families=fams, tree_estimator_name="FastCherries", num_rate_categories=4)
    Q = read_rate_matrix(out2).to_numpy()
    assert Q.shape == (20, 20) and np.all(np.isfinite(Q)) and np.allclose(Q.sum(1), 0.0, atol=1e-9)
    assert np.all(Q - np.diag(np.diag(Q)) >= 0.0) and np.all(np.diag(Q) < 0.0)
    w, v = np.linalg.eig(Q.T)
    pi = np.real(v[:, np.argmin(np.abs(w))])
    pi /= pi.sum()
    flux = pi[:, None] * Q
    assert np.all(pi > 0) and np.allclose(flux, flux.T, atol=1e-9)           # reversible
    assert len(os.listdir(cache2 / "fast_cherries")) == 2                    # one tree estimation per iteration
    for fam in fams:
        tdirs = [cache2 / "fast_cherries" / h / "output_tree_dir" for h in os.listdir(cache2 / "fast_cherries")]
        assert all((d / f"{fam}.txt").exists() and (d / "result.success").exists() for d in tdirs)


def test_public_api_coevolution_runs_to_completion(tmp_path):
    """`cherryml_public_api(model_name="co-evolution")` end to end (reference
    `_cherryml_public_api.py:207-247`): same golden and tolerance as the pipeline test above; the reference
    returns nothing for this model, this build returns its (extra) profiling string."""
    from cherryml_amd._cherryml_public_api import cherryml_public_api
    from cherryml_amd.io import read_rate_matrix
    z = load_golden("demo_e2e.npz")
    dirs, fams = _materialise(tmp_path, z)
    pairs = [a + b for a in AA for b in AA]
    mask = np.unpackbits(z["co_mask_packed"])[:160000].reshape(400, 400)
    mpath = str(tmp_path / "mask.txt")
    pd.DataFrame(mask, index=pairs, columns=pairs).to_csv(mpath, sep=" ")
    out = str(tmp_path / "learned_co.txt")
    prof = cherryml_public_api(output_path=out, model_name="co-evolution", msa_dir=dirs["msa"],
                               contact_map_dir=dirs["contact_map"], tree_dir=dirs["tree"],
                               cache_dir=str(tmp_path / "cache"), num_epochs=int(z["co_epochs"]), families=fams,
                               coevolution_mask_path=mpath, tree_estimator_name="FastTree")
    assert isinstance(prof, str) and "time_optimization" in prof
    learned = read_rate_matrix(out)
    assert list(learned.columns) == pairs
    assert relerr(learned.to_numpy(), z["co_Q_best_f64"]) < 1e-6
