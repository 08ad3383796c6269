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


def test_the_resident_bank_gives_the_compiled_reference_family_after_family():
    """`BleBank` (cb_ble_bank_create / cb_ble_bank_run): the bank uploaded once, the range check, the transposes and the site
    statistics of the initial bins as kernels, the workspace kept between calls -- the three golden families of the compiled
    reference through ONE bank object, in both orders (a grown workspace, then a smaller family in it), and a ragged family
    (n and L not multiples of the 64 x 64 / 32-site tiles) against the per-call entry."""
    from cherryml_amd.phylogeny_estimation import (BleBank, compute_log_transition_matrices,
                                                  estimate_branch_lengths_and_site_rates)
    z = load_golden("ble.npz")
    grid = z["grid_sr"]
    for order in ((0, 1, 2), (2, 0, 1)):
        banks = {}
        for k in order:
            rates = z[f"rnd{k}_rates"]
            key = rates.tobytes()
            if key not in banks:
                banks[key] = BleBank.from_rate_matrix(z["Q"], grid, rates)
            cx, cy = z[f"rnd{k}_x"], z[f"rnd{k}_y"]
            prof = {}
            lengths, srates = banks[key].estimate(cx, cy, np.concatenate([cx, cy]), z[f"rnd{k}_weights"], 50, profile=prof)
            assert np.array_equal(lengths, z[f"rnd{k}_ble_lengths"]) and np.array_equal(srates, z[f"rnd{k}_ble_rates"]), k
            assert prof["iterations"] >= 1 and prof["kernel_ms"] > 0
        for b in banks.values():
            b.close()
    rng = np.random.default_rng(7)
    rates = z["rnd0_rates"]
    logP = compute_log_transition_matrices(z["Q"], grid, rates)
    S = z["Q"].shape[0]
    with BleBank(logP, grid, rates) as bank:
        for n, L in ((37, 91), (130, 33), (5, 200)):
            cx = rng.integers(-1, S, size=(n, L))
            cy = np.where(rng.random((n, L)) < 0.7, cx, rng.integers(-1, S, size=(n, L)))
            seqs = np.concatenate([cx, cy, rng.integers(-1, S, size=(3, L))])
            w = z["rnd0_weights"]
            a = bank.estimate(cx, cy, seqs, w, 50)
            b = estimate_branch_lengths_and_site_rates(cx, cy, seqs, logP, grid, rates, w, 50)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (n, L)
        with pytest.raises(Exception):
            bad = np.full((4, 8), S, dtype=np.int8)       # a state code outside the alphabet
            bank.estimate(bad, bad, bad, z["rnd0_weights"], 5)


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
