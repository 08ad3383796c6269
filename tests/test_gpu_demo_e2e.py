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
