"""FastCherries branch-length / site-rate kernels and the SiteRM site-rate gather on the GPU, against
the known answers of the reference's own tests and outputs of the compiled reference
(tests/golden/ble.npz).  Needs an MI355X: `pytest -m gpu`."""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _pi(Q):
    w, v = np.linalg.eig(Q.T)
    p = v[:, int(np.argmin(np.abs(w.real)))].real
    return p / p.sum()


@pytest.mark.parametrize("spectral", [True, False])
def test_log_bank_matches_oracle(spectral):
    from cherryml_amd.phylogeny_estimation import compute_log_transition_matrices
    from oracle import ble_oracle as bo
    z = load_golden("ble.npz")
    grid, rates = z["grid_sr"][::8], z["rates_sr"]
    got = compute_log_transition_matrices(z["Q"], grid, rates, stationary_distribution=_pi(z["Q"]) if spectral else None)
    ref = bo.log_bank(z["Q"], grid, rates)
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-11)


def test_known_answers_of_the_reference_tests():
    from cherryml_amd.phylogeny_estimation import branch_lengths, compute_log_transition_matrices, rate_priors, site_rates
    z = load_golden("ble.npz")
    bank = compute_log_transition_matrices(z["Q"], z["grid_bl"], z["rates_bl"])
    for k in range(3):
        assert list(branch_lengths(z[f"bl{k}_x"], z[f"bl{k}_y"], bank, z["s2r_bl"])) == list(z[f"bl{k}_expected"]), k
    bank = compute_log_transition_matrices(z["Q"], z["grid_sr"], z["rates_sr"], stationary_distribution=_pi(z["Q"]))
    for k in range(6):
        x = z[f"sr{k}_x"]
        got = site_rates(x, z[f"sr{k}_y"], bank, z["lengths_sr"][:len(x)], rate_priors(z["rates_sr"]))
        assert list(got) == list(z[f"sr{k}_expected"]), k


@pytest.mark.parametrize("k", [0, 1, 2])
def test_random_families_match_the_compiled_reference(k):
    from cherryml_amd.phylogeny_estimation import (branch_lengths, compute_log_transition_matrices,
                                                  estimate_branch_lengths_and_site_rates, rate_priors, site_rates)
    z = load_golden("ble.npz")
    rates, grid = z[f"rnd{k}_rates"], z["grid_sr"]
    bank = compute_log_transition_matrices(z["Q"], grid, rates)
    cx, cy = z[f"rnd{k}_x"], z[f"rnd{k}_y"]
    assert list(branch_lengths(cx, cy, bank, z[f"rnd{k}_s2r"])) == list(z[f"rnd{k}_bl"])
    assert list(site_rates(cx, cy, bank, z[f"rnd{k}_li"], rate_priors(rates))) == list(z[f"rnd{k}_sr"])
    lengths, srates = estimate_branch_lengths_and_site_rates(cx, cy, np.concatenate([cx, cy]), bank, grid, rates,
                                                             z[f"rnd{k}_weights"], 50)
    assert np.array_equal(lengths, z[f"rnd{k}_ble_lengths"])
    assert np.array_equal(srates, z[f"rnd{k}_ble_rates"])


def test_siterm_site_rate_gather_matches_the_cython_reference():
    from cherryml_amd._siterm import compute_optimal_site_rates
    z = load_golden("ble.npz")
    cx, cy = z["gather_x"], z["gather_y"]
    cherries = [(list(map(int, cx[c])), list(map(int, cy[c])), 0.1) for c in range(cx.shape[0])]
    got = compute_optimal_site_rates(cx.shape[1], cherries, z["gather_tensor"], list(z["gather_grid"]),
                                     list(z["gather_prior"]))
    assert np.array_equal(np.array(got), z["gather_expected"])


def test_bad_arguments_raise():
    from cherryml_amd.phylogeny_estimation import branch_lengths
    bank = np.zeros((4, 2, 3, 3))
    with pytest.raises(ValueError):
        branch_lengths(np.zeros((2, 5)), np.zeros((2, 4)), bank, np.zeros(5))
    with pytest.raises(ValueError):
        branch_lengths(np.full((2, 5), 7), np.zeros((2, 5)), bank, np.zeros(5))      # state 7 >= S
    with pytest.raises(ValueError):
        branch_lengths(np.zeros((2, 5)), np.zeros((2, 5)), bank, np.full(5, 9))      # rate index >= R
