"""BASELINE.json's configurations at FULL size (B = 129 buckets; 20 / 400 states; 5000 SiteRM sites;
pair totals up to 1e8), checked through properties that do not need an oracle run of that size:
stochasticity, detailed balance and the semigroup law of the expm bank, linearity / additivity of the
count-weighted loss and gradient in the counts, a directional-derivative identity, optimality at the
generating model (the synthetic banks are noise-free expectations, so the generating Q is the exact
maximiser), independence of SiteRM sites.  The synthetic banks are bench.py's."""
import numpy as np
import pytest
import torch

from conftest import relerr

pytestmark = pytest.mark.gpu


def _coevo(n_pairs=None):
    import bench
    rng = np.random.default_rng(0)
    Q, pi, mask = bench.coevolution_truth(rng)
    t, C = bench.reversible_bank(Q, pi, 1057194.0 if n_pairs is None else n_pairs, rng)
    return Q, pi, mask, t, C


def test_expm_bank_400_states_stochastic_reversible_semigroup():
    from cherryml_amd import CherryBank
    Q, pi, _, t, _ = _coevo()
    # P(2t) = P(t)^2 needs both t and 2t in the bank: the 129 grid points and their doubles
    times = np.concatenate([t, 2.0 * t])
    with CherryBank(times, np.ones((times.size, 400, 400))) as bank:      # (the counts play no role here)
        P = bank.expm_bank(Q, pi)[0]
    assert P.shape == (258, 400, 400)
    assert np.all(P > -1e-15) and np.abs(P.sum(axis=2) - 1.0).max() < 1e-12          # stochastic
    flux = pi[None, :, None] * P
    assert np.abs(flux - np.transpose(flux, (0, 2, 1))).max() < 1e-15                  # detailed balance
    for b in (0, 17, 64, 100, 128):                                                    # semigroup
        assert relerr(P[129 + b], P[b] @ P[b]) < 1e-11, b
    # the smallest bucket keeps RELATIVE accuracy on its O(t^2) entries (double substitutions)
    # (reference: the Taylor series, which has no cancellation at t |Q| ~ 2e-3)
    sel = (Q == 0.0) & ~np.eye(400, dtype=bool)
    tQ = t[0] * Q
    T2 = tQ @ tQ
    series = (np.eye(400) + tQ + T2 / 2.0 + T2 @ tQ / 6.0 + T2 @ T2 / 24.0 + T2 @ T2 @ tQ / 120.0)[sel]
    live = series > 1e-14
    assert live.sum() > 100000 and series[live].max() < 1e-8      # all of them far below the rounding level of U e^{t L} U^T
    # (U e^{t L} U^T alone has ABSOLUTE error ~1e-16: relative 1e-4 and worse on these entries)
    assert np.abs(P[0][sel][live] / series[live] - 1.0).max() < 1e-7


@pytest.mark.parametrize("n_pairs", [1057194.0, 1.0e8])
def test_loss_and_gradient_are_linear_in_the_counts_400_states(n_pairs):
    """Config 3 (the demo bank's total) and config 5 (10 000 families: 1e8 pairs)."""
    from cherryml_amd import CherryBank
    Q, pi, _, t, C = _coevo(n_pairs)
    rng = np.random.default_rng(1)
    Qe = Q * np.exp(0.05 * rng.normal(size=Q.shape))
    d = np.sqrt(pi)
    Qe = 0.5 * (Qe + (Qe * pi[:, None]).T / pi[:, None])     # keep detailed balance w.r.t. pi
    np.fill_diagonal(Qe, 0.0)
    np.fill_diagonal(Qe, -Qe.sum(1))
    C1 = C * rng.uniform(0.0, 1.0, size=(C.shape[0], 1, 1))
    C1 = 0.5 * (C1 + np.transpose(C1, (0, 2, 1)))
    C2 = C - C1
    out = {}
    for name, cc in (("all", C), ("one", C1), ("two", C2), ("x3", 3.0 * C)):
        with CherryBank(t, cc) as bank:
            out[name] = bank.loss_grad(Qe, pi, normalize=False) + bank.loss_grad(Qe, pi, normalize=True)
    l, g, ln, gn = out["all"]
    assert abs(out["one"][0][0] + out["two"][0][0] - l[0]) < 1e-12 * abs(l[0])                  # additivity
    assert relerr(out["one"][1][0] + out["two"][1][0], g[0]) < 1e-11
    assert abs(out["x3"][0][0] - 3.0 * l[0]) < 1e-12 * abs(l[0]) and relerr(out["x3"][1][0], 3.0 * g[0]) < 1e-12
    assert abs(out["x3"][2][0] - ln[0]) < 1e-12 * abs(ln[0]) and relerr(out["x3"][3][0], gn[0]) < 1e-12  # normalised: invariant
    assert abs(ln[0] * C.sum() - l[0]) < 1e-12 * abs(l[0])
    # d/ds L((1+s) Q) at s = 0 equals <dL/dQ, Q> (scaling keeps pi): central difference
    with CherryBank(t, C) as bank:
        eps = 1e-5
        lp = bank.loss_grad((1 + eps) * Qe, pi, normalize=True)[0][0]
        lm = bank.loss_grad((1 - eps) * Qe, pi, normalize=True)[0][0]
    assert abs((lp - lm) / (2 * eps) - float((gn[0] * Qe).sum())) < 1e-7 * max(1.0, abs(float((gn[0] * Qe).sum())))


def test_generating_model_is_optimal_400_states_full_bank():
    """The bank is the expectation under `Q`: no parameter step can lower the loss below its value
    at Q, the parameter gradient vanishes there, and 30 epochs from JTT-IPW approach that value."""
    from cherryml_amd import CherryBank, RateMatrix
    from cherryml_amd.estimation import jtt_ipw_from_arrays
    Q, pi, mask, t, C = _coevo()
    mod = RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(mask), pi=torch.tensor(pi),
                     pi_requires_grad=True, initialization=Q)
    u0, p0 = mod.upper_diag.detach().numpy().copy(), mod._pi.detach().numpy().copy()
    with CherryBank(t, C) as bank:
        at_truth = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=4, lr=0.1)
        l_true = bank.loss_grad(Q, pi, normalize=True)[0][0]
        init = jtt_ipw_from_arrays(t, C, mask)
        mod2 = RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(mask),
                          pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
        learned = bank.train_pande_reversible(mod2.upper_diag.detach().numpy(), mod2._pi.detach().numpy(),
                                              mask=mask, num_epochs=30, lr=0.1)
    assert abs(at_truth["loss"][0] - l_true) < 1e-12 * abs(l_true)
    assert np.all(at_truth["loss"][1:] >= l_true - 1e-12)              # Adam can only move away from the optimum
    # best iterate = the start, up to the jitter of Adam on a gradient that is pure rounding noise there: 1e-12 of the
    # gradient's scale with the buckets summed before the last product (8e-9 in Q after 4 epochs), 1e-16 bucket by bucket
    assert relerr(at_truth["Q_best"], Q) < 1e-7
    assert np.all(learned["loss"] >= l_true - 1e-12)
    assert learned["loss"][-1] - l_true < 0.02 * (learned["loss"][0] - l_true)


def test_lg_full_bank_converges_to_the_generating_matrix():
    """Config 2's bank (1000 families x 200 sites x 64 cherries = 1.28e7 pairs, noise-free): the
    optimiser run to convergence recovers LG."""
    import bench
    from cherryml_amd import CherryBank, RateMatrix
    from cherryml_amd.estimation import jtt_ipw_from_arrays
    rng = np.random.default_rng(0)
    wl = bench.make_workload("lg20", 0, rng)
    lg = bench.lg_matrix()
    init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
    mod = RateMatrix(num_states=20, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                     pi=torch.ones(20, dtype=torch.float64) / 20, pi_requires_grad=True, initialization=init)
    with CherryBank(wl["t"], wl["C"]) as bank:
        r = bank.train_pande_reversible(mod.upper_diag.detach().numpy(), mod._pi.detach().numpy(),
                                        mask=wl["mask"], num_epochs=2000, lr=0.1)
        l_true = bank.loss_grad(lg, bench.stationary(lg), normalize=True)[0][0]
    assert abs(wl["C"].sum() - 1.28e7) < 1.0
    assert np.all(r["loss"] >= l_true - 1e-12)
    assert r["loss"].min() - l_true < 1e-7
    assert relerr(r["Q_best"], lg) < 2e-3
    assert abs(r["loss"][-1] - r["loss"][-2]) < 1e-6


def test_siterm_5000_sites_are_independent():
    """Config 4 at full size (L = 5000, N = 20, B = 129: a 2 GB count tensor): any subset of sites run on
    its own gives the same per-site trajectories and matrices as inside the full batch."""
    import bench
    from cherryml_amd import CherryBank
    from cherryml_amd._siterm._vectorized import _invert
    rng = np.random.default_rng(0)
    wl = bench.make_workload("siterm", 5000, rng)
    th0, Th0 = _invert(wl["init"])
    E = 6
    with CherryBank(wl["t"], wl["C"]) as bank:
        full = bank.train_siterm(th0, Th0, E, lr=0.1)
    sub = np.array([0, 1, 2, 777, 2500, 4998, 4999])
    with CherryBank(wl["t"][sub], wl["C"][sub]) as bank:
        part = bank.train_siterm(th0[sub], Th0[sub], E, lr=0.1)
    assert full["res"].shape == (5000, 20, 20) and np.all(np.isfinite(full["res"]))
    # (a small batch runs with more waves per site: another summation order, so equal to rounding, not bitwise)
    assert np.allclose(full["loss_per_epoch_per_site"][:, sub], part["loss_per_epoch_per_site"], rtol=1e-10, atol=0)
    for k, l in enumerate(sub):
        assert relerr(full["res"][l], part["res"][k]) < 1e-8, l
    lp = full["loss_per_epoch_per_site"]
    assert lp[-1].mean() < lp[0].mean() and np.mean(lp[-1] < lp[0]) > 0.8      # (Adam at lr 0.1 overshoots on a few sites)
