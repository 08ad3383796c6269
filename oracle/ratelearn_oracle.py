"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by the product path.

CPU restatement of the reference's composite-likelihood optimiser, i.e. the
algorithm of

  * cherryml/estimation/_ratelearn/rate.py:61-91,167-188   (pande_reversible)
  * cherryml/estimation/_ratelearn/trainer.py:156-242       (epoch loop)
  * cherryml/estimation/_ratelearn/ratelearner.py:77-152    (dtypes, Adam, seed)
  * cherryml/_siterm/_cherryml_vectorized.py:173-383        (batched over sites)

(paths relative to the reference repo).  The arithmetic the reference delegates
to a third-party dependency -- `torch.matrix_exp` and its autograd backward,
`torch.optim.Adam`; the reference's requirements.txt does not pin torch, the
version used to pin this oracle is torch 2.10.0 (CPU kernels) -- is called here
exactly where the reference calls it, so that in "as-is" (float32) mode this
oracle reproduces the reference's numbers to rounding, and in float64 mode it
is the "f64 oracle recipe" of SURVEY.md section 8c.

Parity pinning: tests/test_oracle_golden.py checks every function below
against tests/golden/*.npz, which were produced by importing and running the
real reference (tests/golden/make_golden.py).  The HIP path (cherryml_amd) is
then compared with this oracle; nothing in cherryml_amd may import it.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
"""
from __future__ import annotations

import time
from typing import Dict, Optional, Tuple

import numpy as np
import torch

__all__ = [
    "stationary_distribution",
    "invert_pande_reversible",
    "pande_reversible_Q",
    "bank_loss",
    "evaluate",
    "train",
    "rate_matrix_Q",
    "evaluate_mode",
    "train_mode",
    "siterm_Q",
    "siterm_invert",
    "siterm_train",
    "jtt_ipw",
]


# --------------------------------------------------------------------- helpers
def stationary_distribution(Q: np.ndarray) -> np.ndarray:
    """Left null vector of Q, normalised (rate.py:10-18)."""
    w, v = np.linalg.eig(Q.T)
    k = int(np.argmin(np.abs(w.real)))
    p = v[:, k].real
    return p / p.sum()


def invert_pande_reversible(Q0: np.ndarray, mask: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Initialisation -> (upper_diag, log_pi), rate.py:61-86.

    Raises ValueError exactly where the reference does (degenerate pi,
    initialisation violating the mask)."""
    S = Q0.shape[0]
    pi = stationary_distribution(Q0)
    if np.any(np.abs(pi) < 1e-8):
        raise ValueError("Stationary distribution of initialization is degenerate.")
    if np.any(np.abs(mask * Q0 - Q0) > 1e-8):
        raise ValueError("initialization not compatible with mask")
    d = np.sqrt(pi)
    sym = (d[:, None] * Q0) / d[None, :]
    iu = np.triu_indices(S, k=1)
    with np.errstate(divide="ignore"):
        upper = np.log(np.exp(sym[iu]) - 1.0)  # softplus^-1; -inf where masked
    return upper, np.log(pi)


def pande_reversible_Q(upper_diag: torch.Tensor, log_pi: torch.Tensor,
                       mask: torch.Tensor) -> torch.Tensor:
    """theta -> Q, rate.py:167-188: R = sym(softplus(u)) * mask,
    Q = D^-1/2 R D^1/2 - diag(rowsum)."""
    S = log_pi.shape[0]
    iu = torch.triu_indices(S, S, offset=1)
    R = torch.zeros(S, S, dtype=upper_diag.dtype)
    R[iu[0], iu[1]] = torch.nn.functional.softplus(upper_diag)
    R = (R + R.T) * mask
    pi = torch.softmax(log_pi, dim=-1)
    root = pi.sqrt()
    Q = (torch.diag(root ** (-1)) @ R) @ torch.diag(root)
    return Q - torch.diag(Q.sum(1))


def bank_loss(Q: torch.Tensor, t: torch.Tensor, C: torch.Tensor,
              normalize: bool = True) -> torch.Tensor:
    """-sum_b <C_b, log expm(t_b Q)> [/ sum C], trainer.py:170-177."""
    logP = torch.log(torch.matrix_exp(t[:, None, None] * Q))
    loss = -(logP * C).sum()
    if normalize:
        loss = loss / C.sum()
    return loss


def evaluate(upper_diag: np.ndarray, log_pi: np.ndarray, mask: np.ndarray,
             t: np.ndarray, C: np.ndarray, dtype=torch.float64,
             normalize: bool = True) -> Dict[str, np.ndarray]:
    """One epoch body: Q, loss, dL/dQ, dL/dupper_diag, dL/dlog_pi.

    `dtype` is the parameter / expm dtype (float32 = reference as is); C is
    always float64 (ratelearner.py:147-152)."""
    u = torch.tensor(upper_diag, dtype=dtype, requires_grad=True)
    p = torch.tensor(log_pi, dtype=dtype, requires_grad=True)
    Q = pande_reversible_Q(u, p, torch.tensor(mask, dtype=torch.float32).to(dtype))
    Q.retain_grad()
    loss = bank_loss(Q, torch.tensor(t, dtype=dtype), torch.tensor(C, dtype=torch.float64),
                     normalize)
    loss.backward()
    f = lambda x: x.detach().numpy().astype(np.float64)
    return dict(Q=f(Q), loss=float(loss.item()), dQ=f(Q.grad), d_upper=f(u.grad),
                d_log_pi=f(p.grad))


def expm_bank(Q: np.ndarray, t: np.ndarray) -> np.ndarray:
    """P[b] = expm(t_b Q) in float64 (debug / cross-checks)."""
    return torch.matrix_exp(torch.tensor(t, dtype=torch.float64)[:, None, None]
                            * torch.tensor(Q, dtype=torch.float64)).numpy()


# ---------------------------------------------------------------- epoch loop
def train(t: np.ndarray, C: np.ndarray, mask: Optional[np.ndarray] = None,
          initialization: Optional[np.ndarray] = None,
          upper_diag: Optional[np.ndarray] = None, log_pi: Optional[np.ndarray] = None,
          num_epochs: int = 2000, lr: float = 0.1, do_adam: bool = True,
          loss_normalization: bool = True, return_best_iter: bool = True,
          dtype=torch.float64, record_time: bool = False) -> Dict[str, np.ndarray]:
    """The optimiser: ratelearner.py:66-145 + trainer.py:118-243.

    Parameters start from `initialization` (inverted as rate.py:61-86), or from
    explicit (upper_diag, log_pi), or -- like the reference with neither --
    from `0.01 * randn` under `torch.manual_seed(0)` and uniform pi.
    Returns loss curve and the Q snapshots the reference writes.
    """
    S = C.shape[-1]
    mask = np.ones((S, S)) if mask is None else np.asarray(mask, dtype=np.float64)
    torch.manual_seed(0)  # ratelearner.py:77
    if initialization is not None:
        u0, p0 = invert_pande_reversible(np.asarray(initialization, dtype=np.float64), mask)
    elif upper_diag is not None:
        u0, p0 = upper_diag, log_pi
    else:
        p0 = np.log(np.full(S, 1.0 / S))
        u0 = (0.01 * torch.randn(S * (S - 1) // 2)).numpy()  # rate.py:51-53 (float32 draw)
    u = torch.tensor(np.asarray(u0), dtype=dtype, requires_grad=True)
    p = torch.tensor(np.asarray(p0), dtype=dtype, requires_grad=True)
    m = torch.tensor(mask, dtype=torch.float32).to(dtype)
    tt = torch.tensor(t, dtype=dtype)
    CC = torch.tensor(C, dtype=torch.float64)
    # parameter order of the reference module: _pi first, then upper_diag
    opt = (torch.optim.Adam([p, u], lr=lr) if do_adam else torch.optim.SGD([p, u], lr=lr))

    snap = lambda Q: Q.detach().numpy().astype(np.float64).copy()
    losses, times, out = [], [], {}
    best, Q_best, Q = None, None, None
    start = time.time()
    for epoch in range(num_epochs):
        opt.zero_grad()
        Q = pande_reversible_Q(u, p, m)
        loss = bank_loss(Q, tt, CC, loss_normalization)
        if best is None or loss < best:  # strict <, Q *before* the step (trainer.py:179-181)
            best, Q_best = loss.detach().clone(), snap(Q)
        if (epoch & (epoch + 1)) == 0:  # epochs 1,2,4,8,... (trainer.py:183-184)
            out[f"Q_{epoch + 1}"] = snap(Q)
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
        times.append(time.time() - start)
    out["loss"] = np.array(losses)
    if record_time:
        out["time"] = np.array(times)
    if num_epochs > 0:
        out["Q_best"] = Q_best
        out["Q_last"] = snap(Q)
        out["result"] = Q_best if return_best_iter else out["Q_last"]
    out["upper_diag"] = u.detach().numpy().astype(np.float64)
    out["log_pi"] = p.detach().numpy().astype(np.float64)
    return out


# ------------------------------------------------- the other parameterisations
def rate_matrix_Q(mode: str, upper_diag: torch.Tensor, lower_diag: Optional[torch.Tensor], log_pi: torch.Tensor,
                  mask: torch.Tensor) -> torch.Tensor:
    """theta -> Q for every mode of the reference's RateMatrix.forward (rate.py:104-218):
      "default"                off-diagonals softplus(upper | lower) * mask, diagonal = -row sums          (:106-128)
      "stationary[_reversible]" R = softplus(upper) (+ its transpose | + softplus(lower)) * mask,
                               diag_i = -(R pi)_i / pi_i,  Q = (R + diag) diag(pi)                          (:130-161)
      "pande"                  as "pande_reversible" with an independent lower triangle                    (:190-217)
    """
    S = log_pi.shape[0]
    if mode == "pande_reversible":
        return pande_reversible_Q(upper_diag, log_pi, mask)
    iu = torch.triu_indices(S, S, offset=1)
    il = torch.tril_indices(S, S, offset=-1)
    R = torch.zeros(S, S, dtype=upper_diag.dtype)
    R[iu[0], iu[1]] = torch.nn.functional.softplus(upper_diag)
    if mode == "stationary_reversible":
        R = R + R.T
    else:
        R[il[0], il[1]] = torch.nn.functional.softplus(lower_diag)
    R = R * mask
    if mode == "default":
        return R - torch.diag(R.sum(1))
    pi = torch.softmax(log_pi, dim=-1)
    if mode in ("stationary", "stationary_reversible"):
        return (R + torch.diag(-(R @ pi) / pi)) @ torch.diag(pi)
    if mode == "pande":
        root = pi.sqrt()
        Q = (torch.diag(root ** (-1)) @ R) @ torch.diag(root)
        return Q - torch.diag(Q.sum(1))
    raise ValueError(f"Unknown rate matrix parameterization: {mode}")


def evaluate_mode(mode: str, upper_diag, lower_diag, log_pi, mask, t, C, normalize: bool = True) -> Dict[str, np.ndarray]:
    """One epoch body (trainer.py:156-186) in float64 for any parameterisation: Q, loss, dL/dQ and the parameter gradients
    (a parameter the mode does not use has gradient zero)."""
    dt = torch.float64
    u = torch.tensor(np.asarray(upper_diag), dtype=dt, requires_grad=True)
    lo = torch.tensor(np.asarray(lower_diag), dtype=dt, requires_grad=True) if lower_diag is not None else None
    p = torch.tensor(np.asarray(log_pi), dtype=dt, requires_grad=True)
    Q = rate_matrix_Q(mode, u, lo, p, torch.tensor(np.asarray(mask), dtype=dt))
    Q.retain_grad()
    loss = bank_loss(Q, torch.tensor(t, dtype=dt), torch.tensor(C, dtype=dt), normalize)
    loss.backward()
    g = lambda x: np.zeros(tuple(x.shape)) if x.grad is None else x.grad.numpy().copy()
    out = dict(Q=Q.detach().numpy().copy(), loss=float(loss.item()), dQ=Q.grad.numpy().copy(), d_upper=g(u), d_log_pi=g(p))
    if lo is not None:
        out["d_lower"] = g(lo)
    return out


def train_mode(mode: str, upper_diag, lower_diag, log_pi, mask, t, C, num_epochs: int, lr: float,
               do_adam: bool = True) -> Dict[str, np.ndarray]:
    """trainer.py:118-243 for any parameterisation, float64; parameters in the reference module's registration order
    (_pi, upper_diag, lower_diag: rate.py:44-57)."""
    dt = torch.float64
    u = torch.tensor(np.asarray(upper_diag), dtype=dt, requires_grad=True)
    lo = torch.tensor(np.asarray(lower_diag), dtype=dt, requires_grad=True) if lower_diag is not None else None
    p = torch.tensor(np.asarray(log_pi), dtype=dt, requires_grad=True)
    m = torch.tensor(np.asarray(mask), dtype=dt)
    tt, CC = torch.tensor(t, dtype=dt), torch.tensor(C, dtype=dt)
    params = [p, u] + ([lo] if lo is not None else [])
    opt = torch.optim.Adam(params, lr=lr) if do_adam else torch.optim.SGD(params, lr=lr)
    losses, best, Q_best, Q = [], None, None, None
    for _ in range(num_epochs):
        opt.zero_grad()
        Q = rate_matrix_Q(mode, u, lo, p, m)
        loss = bank_loss(Q, tt, CC, True)
        if best is None or loss < best:
            best, Q_best = loss.detach().clone(), Q.detach().numpy().copy()
        loss.backward()
        opt.step()
        losses.append(float(loss.item()))
    return dict(loss=np.array(losses), Q_best=Q_best, Q_last=Q.detach().numpy().copy())


# -------------------------------------------------------- SiteRM (vectorised)
def siterm_Q(theta: torch.Tensor, Theta: torch.Tensor) -> torch.Tensor:
    """(theta[L,N], Theta[L,N,N]) -> Q[L,N,N], _cherryml_vectorized.py:242-262."""
    N = theta.shape[1]
    pi = torch.softmax(theta, dim=1)
    upper = torch.triu(torch.ones(N, N, dtype=Theta.dtype), diagonal=1)
    half = torch.nn.functional.softplus(Theta + Theta.transpose(1, 2)) * upper
    Ssym = half + half.transpose(1, 2)
    root = pi.sqrt()
    off = torch.diag_embed(1.0 / root) @ Ssym @ torch.diag_embed(root)
    return off - torch.diag_embed(off.sum(dim=2))


def _siterm_stationary(Qs: np.ndarray) -> np.ndarray:
    """Power iteration of _cherryml_vectorized.py:72-104 (float32 expm, 100
    squarings with row renormalisation)."""
    scale = -1.0 / np.mean(np.diagonal(Qs, axis1=1, axis2=2), axis=1)
    Qn = Qs * scale[:, None, None]
    E = torch.matrix_exp(torch.tensor(Qn, dtype=torch.float32)).numpy()
    for _ in range(100):
        E = E @ E
        E /= E.sum(axis=2, keepdims=True)
    p = E[:, 0, :]
    return p / p.sum(axis=1, keepdims=True)


def siterm_invert(Qs: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Initialisation -> (theta, Theta) in float64, _cherryml_vectorized.py:191-240."""
    L, N, _ = Qs.shape
    pi = _siterm_stationary(Qs)
    if not (np.allclose(pi.sum(axis=1), 1, atol=1e-3) and np.all(pi > 1e-8)):
        raise ValueError("At least one stationary distribution is degenerate.")
    root = np.sqrt(pi)  # float32, like the reference's pi_all
    inv_root = 1.0 / root  # float32 reciprocal (:213), then widened by the product
    Ssym = (root[:, :, None] * Qs) * inv_root[:, None, :]
    iu = np.triu_indices(N, k=1)
    Th = np.zeros_like(Ssym)
    Th[:, iu[0], iu[1]] = np.log(np.exp(Ssym[:, iu[0], iu[1]]) - 1.0)
    Th = (Th + Th.transpose(0, 2, 1)) / 2.0
    return np.log(pi).astype(np.float64), Th.astype(np.float64)


def siterm_loss(Q: torch.Tensor, counts: torch.Tensor, times: torch.Tensor):
    """Per-site normalised loss, _cherryml_vectorized.py:264-293."""
    L, B = times.shape
    logP = torch.log(torch.matrix_exp(times.view(L, B, 1, 1) * Q.unsqueeze(1)))
    per_site = -(counts * logP).sum(dim=(1, 2, 3)) / counts.sum(dim=(1, 2, 3))
    return per_site, per_site.sum()


def siterm_train(counts: np.ndarray, times: np.ndarray, num_epochs: int,
                 initialization: Optional[np.ndarray] = None,
                 theta: Optional[np.ndarray] = None, Theta: Optional[np.ndarray] = None,
                 lr: float = 0.1) -> Dict[str, np.ndarray]:
    """_cherryml_vectorized.py:295-402: Adam(lr 0.1) on all sites at once, per-site best Q."""
    L, B, N, _ = counts.shape
    torch.manual_seed(42)  # set_seed(42), :307-315
    th = 0.01 * torch.randn(L, N)  # float32 draws, consumed even when overwritten
    Th = 0.01 * torch.randn(L, N, N)
    if initialization is not None:
        a, b = siterm_invert(np.asarray(initialization, dtype=np.float64))
        th, Th = torch.tensor(a), torch.tensor(b)
    elif theta is not None:
        th, Th = torch.tensor(theta), torch.tensor(Theta)
    th.requires_grad_(True)
    Th.requires_grad_(True)
    cnt = torch.tensor(counts, dtype=torch.float64)
    tms = torch.tensor(np.asarray(times), dtype=torch.float64)
    opt = torch.optim.Adam([th, Th], lr=lr)
    best = torch.full((L,), float("inf"), dtype=torch.float64)
    Q_best = siterm_Q(th, Th).detach()
    lpe = np.zeros(num_epochs)
    lpeps = np.zeros((num_epochs, L))
    for e in range(num_epochs):
        opt.zero_grad()
        Q = siterm_Q(th, Th)
        per_site, total = siterm_loss(Q, cnt, tms)
        better = per_site < best
        best = torch.where(better, per_site.detach().to(best.dtype), best)
        Q_best = torch.where(better.view(-1, 1, 1), Q.detach(), Q_best)
        lpeps[e] = per_site.detach().numpy()
        lpe[e] = float(total.item())
        total.backward()
        opt.step()
    return dict(res=Q_best.numpy().astype(np.float64), loss_per_epoch=lpe,
                loss_per_epoch_per_site=lpeps,
                theta=th.detach().numpy(), Theta=Th.detach().numpy())


# --------------------------------------------------------------------- JTT-IPW
def jtt_ipw(t: np.ndarray, C: np.ndarray, mask: Optional[np.ndarray] = None,
            use_ipw: bool = True, pseudocounts: float = 1e-8,
            symmetrize: bool = True) -> np.ndarray:
    """Closed-form initialiser, cherryml/estimation/_jtt_ipw.py:66-110."""
    B, S, _ = C.shape
    mask = np.ones((S, S)) if mask is None else mask
    Cs = C + pseudocounts
    if symmetrize:
        Cs = 0.5 * (Cs + Cs.transpose(0, 2, 1))
    Cs = Cs * mask[None]
    off = 1.0 - np.eye(S)
    F = Cs.sum(0)
    F_off = F * off
    ctp = F_off / F_off.sum(axis=1)[:, None]
    if use_ipw:
        M = ((Cs * off[None]).sum(axis=2) / np.asarray(t)[:, None]).sum(0) / F.sum(axis=1)
    else:
        M = F_off.sum(axis=1) / np.median(t) / F.sum(axis=1)
    res = M[:, None] * ctp
    np.fill_diagonal(res, -M)
    return res
