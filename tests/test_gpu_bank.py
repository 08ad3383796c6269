"""Parity of the HIP path (through the C ABI) against the oracle and the
reference-generated golden vectors.  Needs an MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from conftest import load_golden, relerr
from oracle import ratelearn_oracle as orc

pytestmark = pytest.mark.gpu


def _bank(t, C):
    from cherryml_amd import CherryBank
    return CherryBank(t, C, device=0)


def _pi_of(log_pi):
    p = np.exp(log_pi - np.max(log_pi))
    return p / p.sum()


def _sym_rate(rng, S, scale=1.0):
    """random reversible rate matrix and its stationary distribution"""
    pi = rng.dirichlet(np.full(S, 5.0))
    R = rng.gamma(2.0, 0.5, size=(S, S)) * scale
    R = np.triu(R, 1)
    R = R + R.T
    d = np.sqrt(pi)
    Q = R * d[None, :] / d[:, None]
    Q -= np.diag(Q.sum(1))
    return Q, pi


@pytest.mark.parametrize("S,L", [(3, 1), (4, 5), (16, 3), (20, 7), (21, 2), (32, 2)])
def test_eigh_small(S, L):
    rng = np.random.default_rng(S * 100 + L)
    A = np.zeros((L, S, S))
    for l in range(L):
        Q, pi = _sym_rate(rng, S)
        d = np.sqrt(pi)
        A[l] = d[:, None] * Q / d[None, :]
        A[l] = 0.5 * (A[l] + A[l].T)
    with _bank(np.ones((L, 1)), np.ones((L, 1, S, S))) as bank:
        lam, U = bank.eigh(A)
    for l in range(L):
        assert np.abs(U[l].T @ U[l] - np.eye(S)).max() < 1e-13
        assert np.abs(A[l] @ U[l] - U[l] * lam[l][None, :]).max() < 1e-13 * max(1.0, np.abs(A[l]).max())
        assert np.allclose(np.sort(lam[l]), np.linalg.eigvalsh(A[l]), atol=1e-13 * np.abs(A[l]).max())


@pytest.mark.parametrize("case", ["toy3_init", "toy3_mask", "s20_symmask"])
def test_expm_bank_small(case):
    g = load_golden(f"eval_{case}.npz")
    Q, pi = g["Q_f64"], _pi_of(g["log_pi"])
    with _bank(g["t"], g["C"]) as bank:
        P = bank.expm_bank(Q, pi)[0]
    ref = orc.expm_bank(Q, g["t"])
    # absolute error: eps-level, plus the unavoidable t * |dlambda| ~ t * eps * |Q|
    # of any spectral (or squaring) method -- one fixture bucket has t = 9999
    tol = 1e-13 + 8 * 2.2e-16 * g["t"][:, None, None] * np.abs(Q).max()
    assert np.all(np.abs(P - ref) < tol)
    small_t = g["t"] < 10
    # elementwise RELATIVE accuracy (tiny masked entries too) at ordinary t
    assert np.abs(P[small_t] / ref[small_t] - 1.0).max() < 1e-11


@pytest.mark.parametrize("case", ["toy3_init", "toy3_mask", "s20_symmask"])
def test_loss_grad_small_vs_reference_golden(case):
    g = load_golden(f"eval_{case}.npz")
    Q, pi = g["Q_f64"], _pi_of(g["log_pi"])
    with _bank(g["t"], g["C"]) as bank:
        loss, dQ = bank.loss_grad(Q, pi, normalize=True)
        loss_u, dQ_u = bank.loss_grad(Q, pi, normalize=False)
        n = bank.total_counts[0]
    assert abs(loss[0] - float(g["loss_f64"])) < 1e-12 * abs(float(g["loss_f64"]))
    assert relerr(dQ[0], g["dQ_f64"]) < 1e-11
    assert abs(n - g["C"].sum()) < 1e-9 * n
    assert abs(loss_u[0] / n - loss[0]) < 1e-12 * abs(loss[0])
    assert relerr(dQ_u[0] / n, dQ[0]) < 1e-13


def test_loss_grad_lg_bank_129_buckets():
    g = load_golden("traj_lgbank.npz")
    Q = g["Q_best_f64"]
    pi = orc.stationary_distribution(Q)
    Qt = torch.tensor(Q, requires_grad=True)
    ref = orc.bank_loss(Qt, torch.tensor(g["t"]), torch.tensor(g["C"]))
    ref.backward()
    with _bank(g["t"], g["C"]) as bank:
        loss, dQ = bank.loss_grad(Q, pi)
    assert abs(loss[0] - ref.item()) < 1e-12 * abs(ref.item())
    assert relerr(dQ[0], Qt.grad.numpy()) < 1e-10


def test_loss_grad_siterm_batch():
    """L independent sites with their own branch-length grids."""
    g = load_golden("siterm_aa.npz")
    counts, times = g["counts"], g["times"]
    L, B, N, _ = counts.shape
    rng = np.random.default_rng(1)
    Qs = np.zeros((L, N, N))
    pis = np.zeros((L, N))
    for l in range(L):
        Qs[l], pis[l] = _sym_rate(rng, N, scale=0.3)
    Qt = torch.tensor(Qs, requires_grad=True)
    per_site, total = orc.siterm_loss(Qt, torch.tensor(counts), torch.tensor(times))
    total.backward()
    with _bank(times, counts) as bank:
        loss, dQ = bank.loss_grad(Qs, pis)
    assert np.allclose(loss, per_site.detach().numpy(), rtol=1e-12, atol=0)
    for l in range(L):
        assert relerr(dQ[l], Qt.grad[l].numpy()) < 1e-10


def test_eigh_large():
    g = load_golden("eval_s400_mask.npz")
    Q, pi = g["Q_f64"], _pi_of(g["log_pi"])
    d = np.sqrt(pi)
    A = d[:, None] * Q / d[None, :]
    A = 0.5 * (A + A.T)
    with _bank(g["t"], g["C"]) as bank:
        lam, U = bank.eigh(A)
    lam, U = lam[0], U[0]
    assert np.abs(U.T @ U - np.eye(400)).max() < 1e-12
    assert np.abs(A @ U - U * lam[None, :]).max() < 1e-12 * np.abs(A).max()


def test_expm_bank_large():
    g = load_golden("eval_s400_mask.npz")
    Q, pi = g["Q_f64"], _pi_of(g["log_pi"])
    with _bank(g["t"], g["C"]) as bank:
        P = bank.expm_bank(Q, pi)[0]
    ref = orc.expm_bank(Q, g["t"])
    tol = 1e-13 + 8 * 2.2e-16 * g["t"][:, None, None] * np.abs(Q).max()
    assert np.all(np.abs(P - ref) < tol)
    small_t = g["t"] < 10
    assert np.abs(P[small_t] / ref[small_t] - 1.0).max() < 1e-10


def test_loss_grad_large_vs_reference_golden():
    g = load_golden("eval_s400_mask.npz")
    Q, pi = g["Q_f64"], _pi_of(g["log_pi"])
    with _bank(g["t"], g["C"]) as bank:
        loss, dQ = bank.loss_grad(Q, pi)
    assert abs(loss[0] - float(g["loss_f64"])) < 1e-12 * abs(float(g["loss_f64"]))
    assert relerr(dQ[0], g["dQ_f64"]) < 1e-10


def test_bad_arguments_raise():
    from cherryml_amd import CherryBank
    with pytest.raises(ValueError):
        CherryBank(np.ones(3), np.ones((3, 4, 5)))
    with pytest.raises(ValueError):
        CherryBank(np.ones(2), np.zeros((2, 3, 3)))  # zero total count


def _random_rate_matrix(rng, S):
    Q = rng.gamma(2.0, 0.3, size=(S, S)) + 0.01  # dense and NOT reversible
    np.fill_diagonal(Q, 0.0)
    Q -= np.diag(Q.sum(1))
    return Q


@pytest.mark.parametrize("S,B", [(3, 4), (20, 9), (21, 5), (32, 3), (33, 3), (48, 9), (100, 5), (161, 3)])
def test_general_path_non_reversible(S, B):
    """cb_loss_grad_general / cb_expm_bank(pi=NULL): arbitrary (non-reversible) Q.  S <= 32: one workgroup per
    site (general_small.hip.h); S > 32: batched 80 x 80-tile GEMMs over the buckets (general_large.hip.h)."""
    rng = np.random.default_rng(S)
    Q = _random_rate_matrix(rng, S)
    t = np.array([float("%.8f" % (0.03 * 1.1 ** i)) for i in np.linspace(-64, 64, B).astype(int)])
    C = rng.poisson(3.0, size=(B, S, S)).astype(float)
    Qt = torch.tensor(Q, requires_grad=True)
    ref = orc.bank_loss(Qt, torch.tensor(t), torch.tensor(C))
    ref.backward()
    with _bank(t, C) as bank:
        P = bank.expm_bank(Q, None)[0]
        loss, dQ = bank.loss_grad_general(Q)
    assert np.abs(P - orc.expm_bank(Q, t)).max() < 1e-13
    assert abs(loss[0] - ref.item()) < 1e-12 * abs(ref.item()), (loss[0], ref.item())
    assert relerr(dQ[0], Qt.grad.numpy()) < 1e-10, relerr(dQ[0], Qt.grad.numpy())


def test_general_path_reference_non_symmetric_mask_golden():
    """The reference's own 20x20 case (non-symmetric random mask): golden from the reference."""
    g = load_golden("eval_s20_mask.npz")
    with _bank(g["t"], g["C"]) as bank:
        loss, dQ = bank.loss_grad_general(g["Q_f64"])
        # and the reversible golden through the general path too: both paths must agree
        gs = load_golden("eval_s20_symmask.npz")
        loss_s, dQ_s = bank.loss_grad_general(gs["Q_f64"])
    assert abs(loss[0] - float(g["loss_f64"])) < 1e-11 * abs(float(g["loss_f64"]))
    assert relerr(dQ[0], g["dQ_f64"]) < 1e-9  # one bucket has t = 9999 (18 squarings)
    assert abs(loss_s[0] - float(gs["loss_f64"])) < 1e-11 * abs(float(gs["loss_f64"]))
    assert relerr(dQ_s[0], gs["dQ_f64"]) < 1e-9


def _oracle_loss_grad(Q, t, C):
    Qt = torch.tensor(Q, requires_grad=True)
    loss = orc.bank_loss(Qt, torch.tensor(t), torch.tensor(C))
    loss.backward()
    return loss.item(), Qt.grad.numpy()


@pytest.mark.parametrize("S,B", [(48, 3), (100, 2), (161, 2), (400, 2)])
def test_symmetric_counts_take_the_triangular_tile_path(S, B):
    """C_b == C_b^T (what cherry counting produces): K3 multiplies upper-triangular tiles only
    and mirrors; same answers as the oracle on the full matrices."""
    rng = np.random.default_rng(S + 7 * B)
    Q, pi = _sym_rate(rng, S, scale=4.0 / S)
    t = np.sort(rng.uniform(0.02, 1.5, size=B))
    C = rng.poisson(2.0, size=(B, S, S)).astype(np.float64)
    C = C + C.transpose(0, 2, 1)
    ref_loss, ref_grad = _oracle_loss_grad(Q, t, C)
    with _bank(t, C) as bank:
        loss, dQ = bank.loss_grad(Q, pi)
    assert abs(loss[0] - ref_loss) < 1e-12 * abs(ref_loss)
    assert relerr(dQ[0], ref_grad) < 1e-10


@pytest.mark.parametrize("S,B", [(2, 3), (5, 1), (9, 6), (12, 5), (17, 4), (21, 7), (24, 9), (25, 3), (32, 5), (33, 3), (48, 2), (100, 3), (161, 2), (415, 2),
                                 (640, 1), (1024, 1)])   # 1024 = the documented maximum
def test_odd_sizes_padding_and_partial_tiles(S, B):
    """S = 33..161 take the large path with LD = ceil16(S) zero padding and partial 80-tiles;
    S <= 32 the register-chained path with every (NT, KS) instantiation."""
    rng = np.random.default_rng(S * 7 + B)
    Q, pi = _sym_rate(rng, S, scale=2.0 / S)
    t = np.array([3e-4, 0.02, 0.3, 1.7, 9.0])[:B] if B <= 5 else np.linspace(0.01, 5, B)
    C = rng.poisson(2.0, size=(B, S, S)).astype(float)
    C[:, np.arange(S), np.arange(S)] += 20
    if B > 1:
        C[1] = 0.0  # an empty bucket contributes nothing
    ref_loss, ref_dQ = _oracle_loss_grad(Q, t, C)
    with _bank(t, C) as bank:
        loss, dQ = bank.loss_grad(Q, pi)
        P = bank.expm_bank(Q, pi)[0]
    assert abs(loss[0] - ref_loss) < 1e-12 * abs(ref_loss)
    assert relerr(dQ[0], ref_dQ) < 1e-10
    assert np.abs(P - orc.expm_bank(Q, t)).max() < 1e-13


def test_many_small_sites():
    """2000 independent 4-state sites with their own grids (SiteRM DNA shape)."""
    rng = np.random.default_rng(5)
    L, B, N = 2000, 7, 4
    Qs = np.zeros((L, N, N)); pis = np.zeros((L, N))
    for l in range(L):
        Qs[l], pis[l] = _sym_rate(rng, N)
    times = rng.uniform(0.01, 2.0, size=(L, B))
    counts = rng.poisson(5.0, size=(L, B, N, N)).astype(float) + 1.0
    Qt = torch.tensor(Qs, requires_grad=True)
    per_site, total = orc.siterm_loss(Qt, torch.tensor(counts), torch.tensor(times))
    total.backward()
    with _bank(times, counts) as bank:
        loss, dQ = bank.loss_grad(Qs, pis)
    assert np.allclose(loss, per_site.detach().numpy(), rtol=1e-12, atol=0)
    assert relerr(dQ, Qt.grad.numpy()) < 1e-11


def test_masked_rates_and_zero_counts():
    """Structural zeros in Q (co-evolution style mask) with zero counts on them."""
    rng = np.random.default_rng(11)
    S = 16
    Q, pi = _sym_rate(rng, S)
    mask = (rng.random((S, S)) < 0.4)
    mask = np.triu(mask, 1); mask = (mask | mask.T | np.eye(S, dtype=bool)).astype(float)
    d = np.sqrt(pi)
    R = (d[:, None] * Q / d[None, :]) * mask
    np.fill_diagonal(R, 0.0)
    Q = R * d[None, :] / d[:, None]
    Q -= np.diag(Q.sum(1))
    t = np.array([6.73e-5, 1e-3, 0.05, 2.0])
    P = orc.expm_bank(Q, t)
    C = np.round(1e6 * pi[None, :, None] * P)  # exact zeros where P is tiny
    ref_loss, ref_dQ = _oracle_loss_grad(Q, t, C)
    with _bank(t, C) as bank:
        loss, dQ = bank.loss_grad(Q, pi)
        Pd = bank.expm_bank(Q, pi)[0]
    assert abs(loss[0] - ref_loss) < 1e-12 * abs(ref_loss)
    assert relerr(dQ[0], ref_dQ) < 1e-10
    nz = P > 0
    assert np.abs(Pd[nz] / P[nz] - 1.0).max() < 1e-9  # relative, including O(t^2) entries


def test_non_finite_input_rejected():
    from cherryml_amd import CherryBank
    with pytest.raises(ValueError):
        CherryBank(np.array([0.1, np.nan]), np.ones((2, 3, 3)))
    with pytest.raises(ValueError):
        CherryBank(np.array([0.1, 0.2]), np.full((2, 3, 3), np.inf))


def test_results_are_bitwise_reproducible():
    g = load_golden("eval_s400_mask.npz")
    Q, pi = g["Q_f64"], _pi_of(g["log_pi"])
    with _bank(g["t"], g["C"]) as bank:
        a = bank.loss_grad(Q, pi)
    with _bank(g["t"], g["C"]) as bank:
        b = bank.loss_grad(Q, pi)
    assert a[0][0] == b[0][0] and np.array_equal(a[1], b[1])


# ---- empty buckets: stored and visited live-only; results identical to the dense oracle ----
def _with_empty_buckets(rng, C, frac=0.6):
    """zero out a random subset of buckets (never all)"""
    C = C.copy()
    B = C.shape[-3]
    kill = rng.random(B) < frac
    kill[rng.integers(B)] = False
    C[..., kill, :, :] = 0.0
    return C, kill


@pytest.mark.parametrize("S,B", [(20, 12), (48, 7), (100, 5)])
def test_empty_buckets_single_bank(S, B):
    rng = np.random.default_rng(S + B)
    Q, pi = _sym_rate(rng, S, scale=0.2)
    t = np.sort(rng.uniform(0.01, 2.0, size=B))
    C = rng.poisson(3.0, size=(B, S, S)).astype(np.float64)
    C, kill = _with_empty_buckets(rng, C)
    ref_loss, ref_grad = _oracle_loss_grad(Q, t, C)
    with _bank(t, C) as bank:
        assert bank.live_buckets[0] == int((~kill).sum())
        loss, dQ = bank.loss_grad(Q, pi)
        P = bank.expm_bank(Q, pi)[0]          # all B buckets, caller's order
    assert abs(loss[0] - ref_loss) < 1e-12 * abs(ref_loss)
    assert relerr(dQ[0], ref_grad) < 1e-10
    assert P.shape == (B, S, S)
    assert relerr(P, orc.expm_bank(Q, t)) < 1e-12


def test_empty_buckets_differ_per_site_and_general_path():
    rng = np.random.default_rng(11)
    L, B, N = 9, 10, 20
    counts = rng.poisson(2.0, size=(L, B, N, N)).astype(np.float64)
    times = np.sort(rng.uniform(0.02, 3.0, size=(L, B)), axis=1)
    live = []
    for l in range(L):
        counts[l], kill = _with_empty_buckets(rng, counts[l], frac=0.1 * l)
        live.append(int((~kill).sum()))
    Qs = np.zeros((L, N, N))
    pis = np.zeros((L, N))
    for l in range(L):
        Qs[l], pis[l] = _sym_rate(rng, N, scale=0.3)
    Qt = torch.tensor(Qs, requires_grad=True)
    per_site, total = orc.siterm_loss(Qt, torch.tensor(counts), torch.tensor(times))
    total.backward()
    with _bank(times, counts) as bank:
        assert list(bank.live_buckets) == live
        loss, dQ = bank.loss_grad(Qs, pis)
        loss_g, dQ_g = bank.loss_grad_general(Qs)
        P = bank.expm_bank(Qs, pis)
    assert np.allclose(loss, per_site.detach().numpy(), rtol=1e-12, atol=0)
    assert np.allclose(loss_g, per_site.detach().numpy(), rtol=1e-11, atol=0)
    for l in range(L):
        assert relerr(dQ[l], Qt.grad[l].numpy()) < 1e-10
        assert relerr(dQ_g[l], Qt.grad[l].numpy()) < 1e-9
        assert relerr(P[l], orc.expm_bank(Qs[l], times[l])) < 1e-12
