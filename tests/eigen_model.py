"""numpy model of the DEVICE algorithm (test helper, not product, not oracle).

It mirrors, step by step and in float64, what the HIP kernels in
cherryml_amd/csrc compute, so that a kernel bug can be bisected on the CPU:

  A   = sym(D^1/2 Q D^-1/2)           (pande_reversible => A symmetric)
  A   = U diag(lam) U^T               (Jacobi on the device; numpy eigh here)
  Pt_b = I + t_b A + U diag(phi2(t_b lam)) U^T,   phi2(x) = e^x - 1 - x
        (same as U e^{t lam} U^T, but entries that are O(t^2) -- masked
         double substitutions at tiny t -- keep full relative accuracy)
  loss = -sum_b <C_b, log Pt_b + (log d_j - log d_i)> / n
  Gt_b = -C_b / Pt_b / n
  M    = sum_b (U^T Gt_b U) o Phi(t_b),  Phi_ij = (e^{t li}-e^{t lj})/(li-lj)
  dL/dA = U M U^T ;  dL/dQ = D^1/2 (dL/dA) D^-1/2
"""
import numpy as np


def phi2(x):
    """e^x - 1 - x without cancellation."""
    x = np.asarray(x, dtype=np.float64)
    small = np.abs(x) < 0.5
    xs = np.where(small, x, 0.0)
    # Taylor to x^17/17!  (|x|<0.5 => term < 1e-20)
    acc = np.zeros_like(xs)
    for k in range(17, 1, -1):
        acc = (acc + 1.0) * xs / k if k > 2 else (acc + 1.0) * xs * xs / 2.0
    big = np.expm1(x) - x
    return np.where(small, acc, big)


def divided_difference(lam, t):
    """Phi_ij(t) = (e^{t li} - e^{t lj}) / (li - lj); diagonal t e^{t li}."""
    x = t * lam
    hi = np.maximum(x[:, None], x[None, :])
    delta = np.abs(x[:, None] - x[None, :])
    with np.errstate(invalid="ignore", divide="ignore"):
        ratio = np.where(delta > 1e-5, -np.expm1(-delta) / delta,
                         1.0 - delta / 2.0 + delta * delta / 6.0)
    return t * np.exp(hi) * ratio


def loss_and_grad_Q(Q, pi, t, C, normalize=True):
    """Returns loss, dL/dQ (free matrix), and intermediates."""
    S = Q.shape[0]
    d = np.sqrt(pi)
    A = d[:, None] * Q / d[None, :]
    A = 0.5 * (A + A.T)
    lam, U = np.linalg.eigh(A)
    n = C.sum() if normalize else 1.0
    logd = np.log(d)
    M = np.zeros((S, S))
    loss = 0.0
    for b in range(len(t)):
        Pt = np.eye(S) + t[b] * A + (U * phi2(t[b] * lam)) @ U.T
        nz = C[b] != 0
        logP = np.where(nz, np.log(np.where(nz, Pt, 1.0)) + (logd[None, :] - logd[:, None]), 0.0)
        loss -= (C[b] * logP).sum() / n
        Gt = np.where(nz, -C[b] / np.where(nz, Pt, 1.0) / n, 0.0)
        M += (U.T @ Gt @ U) * divided_difference(lam, t[b])
    dA = U @ M @ U.T
    dQ = d[:, None] * dA / d[None, :]
    return loss, dQ, dict(A=A, lam=lam, U=U, dA=dA)


def softplus(x):
    return np.logaddexp(0.0, x)


def params_to_A(upper, log_pi, mask):
    """(upper_diag, log_pi, mask) -> (A, pi, R) without going through Q."""
    S = log_pi.shape[0]
    iu = np.triu_indices(S, k=1)
    R = np.zeros((S, S))
    R[iu] = softplus(upper)
    R = (R + R.T) * mask
    p = np.exp(log_pi - log_pi.max())
    pi = p / p.sum()
    d = np.sqrt(pi)
    A = R.copy()
    A[np.diag_indices(S)] = -(R * d[None, :]).sum(1) / d
    return A, pi, R


def param_grads(dA, upper, log_pi, mask, C, normalize=True):
    """Back-propagate dL/dA (free matrix) + the direct pi term to the
    reference's parameters (upper_diag, log_pi)."""
    S = log_pi.shape[0]
    A, pi, R = params_to_A(upper, log_pi, mask)
    d = np.sqrt(pi)
    n = C.sum() if normalize else 1.0
    gdiag = np.diag(dA)
    # A_ij = R_ij (i != j);  A_ii = -sum_j R_ij d_j / d_i
    dR = mask * (dA - gdiag[:, None] * d[None, :] / d[:, None])
    np.fill_diagonal(dR, 0.0)
    iu = np.triu_indices(S, k=1)
    sig = 1.0 / (1.0 + np.exp(-upper))
    d_upper = sig * (dR[iu] + dR.T[iu])
    # log d_k derivative
    Ctot = C.sum(0)
    g_ld = -(R * gdiag[:, None] / d[:, None]).sum(0) * d - gdiag * np.diag(A)
    g_ld += -(Ctot.sum(0) - Ctot.sum(1)) / n
    d_log_pi = 0.5 * (g_ld - pi * g_ld.sum())
    return d_upper, d_log_pi
