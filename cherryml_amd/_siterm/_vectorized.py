"""SiteRM: all sites at once (reference: cherryml/_siterm/_cherryml_vectorized.py:107-402).

Same signature and return dict as the reference's
`quantized_transitions_mle_vectorized_over_sites`; the per-site expm banks,
losses and gradients come from one libcherrybank call per epoch (one
workgroup per site), torch keeps the parameterisation (theta, Theta) and Adam.
"""
import logging
import time
import warnings
from typing import Dict, Optional

import numpy as np
import torch

from .._autograd import bank_loss
from ..bank import CherryBank


def _normalize_rate_matrices(Qs: np.ndarray) -> np.ndarray:
    scale = -1.0 / np.mean(np.diagonal(Qs, axis1=1, axis2=2), axis=1)
    return Qs * scale[:, None, None]


def solve_stationary_dist_fast(rate_matrices: np.ndarray, device: str = "cpu") -> np.ndarray:
    """Stationary distributions by repeated squaring of exp(Q/|mean diag|) (reference :72-104).

    This ONE-OFF initialisation (not the epoch loop) keeps the reference's own third-party call, `torch.matrix_exp` of a
    float32 tensor on the host, on purpose: the parameters the optimisation starts from are the inverse of THIS float32 result,
    and the trajectory goldens are followed from the reference's start to 1e-8.  Measured in round 6 with exp(Q_l) from the
    library's float64 kernels instead (`CherryBank.expm_only(..., num_sites=L).expm_bank`, rounded to float32): the first
    loss of the 5000-site batch moves by 2e-8 and the 100-epoch curves by 5.6e-4 (tests/test_gpu_siterm_cfg4.py) -- the
    float32 rounding of torch's Pade evaluation is part of the reference's starting point."""
    Qn = _normalize_rate_matrices(np.asarray(rate_matrices))
    E = torch.matrix_exp(torch.tensor(Qn, dtype=torch.float32)).numpy()
    for _ in range(100):
        E = E @ E
        E /= E.sum(axis=2, keepdims=True)
    p = E[:, 0, :]
    return p / p.sum(axis=1, keepdims=True)


def _invert(initialization: np.ndarray):
    L, N, _ = initialization.shape
    pi = solve_stationary_dist_fast(initialization)
    if not (np.allclose(pi.sum(axis=1), 1, atol=1e-3) and np.all(pi > 1e-8)):
        raise ValueError("At least one stationary distribution is degenerate.")
    root = np.sqrt(pi)
    inv_root = 1.0 / root
    Ssym = (root[:, :, None] * initialization) * inv_root[:, None, :]
    if not np.allclose(np.abs(Ssym - Ssym.transpose(0, 2, 1)), 0, atol=1e-4):
        warnings.warn("At least one S matrix is not symmetric up to 4 decimal places.")
    iu = np.triu_indices(N, k=1)
    Th = np.zeros_like(Ssym)
    Th[:, iu[0], iu[1]] = np.log(np.exp(Ssym[:, iu[0], iu[1]]) - 1.0)
    Th = (Th + Th.transpose(0, 2, 1)) / 2.0
    return np.log(pi).astype(np.float64), Th.astype(np.float64)


def _site_Q(theta: torch.Tensor, Theta: torch.Tensor, upper: torch.Tensor):
    pi = torch.softmax(theta, dim=1)
    half = torch.nn.functional.softplus(Theta + Theta.transpose(1, 2)) * upper
    sym = half + half.transpose(1, 2)
    root = pi.sqrt()
    off = sym * (root[:, None, :] / root[:, :, None])
    return off - torch.diag_embed(off.sum(dim=2)), pi


def quantized_transitions_mle_vectorized_over_sites(
    counts: np.ndarray, times, num_epochs: int, initialization: Optional[np.ndarray] = None,
    num_cores: int = 1, device: str = "cpu", fused: bool = True,
) -> Dict:
    prof = {}
    st = time.time()
    logger = logging.getLogger(__name__)
    from .._device import resolve_device
    resolve_device(device, "quantized_transitions_mle_vectorized_over_sites")   # "cpu" and "cuda": the MI355X
    if not torch.cuda.is_available():
        raise ValueError("device=cuda requested but device not available.")
    dev = torch.device("cuda")
    counts = np.ascontiguousarray(counts, dtype=np.float64)
    times = np.ascontiguousarray(np.asarray(times, dtype=np.float64))
    L, B, N, _ = counts.shape
    prof["time_preamble"] = time.time() - st
    st = time.time()
    bank = CherryBank(times, counts, device=dev.index or 0)
    prof["time_send_counts_to_gpu"] = time.time() - st
    st = time.time()
    logger.info(f"Going to estimate site rate matrices for L={L} sites, over N={N} states. "
                f"Number of time buckets: {B}.")
    # reference: set_seed(42); theta, Theta = 0.01*randn drawn even if overwritten
    torch.manual_seed(42)
    theta0 = (0.01 * torch.randn(L, N)).double()
    Theta0 = (0.01 * torch.randn(L, N, N)).double()
    if initialization is not None:
        a, b = _invert(np.asarray(initialization, dtype=np.float64))
        theta0, Theta0 = torch.tensor(a), torch.tensor(b)
    upper = torch.triu(torch.ones(N, N, dtype=torch.float64, device=dev), diagonal=1)
    if fused and N <= 32:
        # every site runs its whole optimisation in one workgroup of one kernel launch
        with torch.no_grad():
            Q0 = _site_Q(theta0.to(dev), Theta0.to(dev), upper)[0].cpu().numpy()
        if initialization is not None:
            np.testing.assert_almost_equal(Q0, initialization, decimal=3)
        prof["time_initialize_model"] = time.time() - st
        st = time.time()
        try:
            r = bank.train_siterm(theta0.numpy(), Theta0.numpy(), num_epochs, lr=0.1)
        finally:
            bank.close()
        lp = r["loss_per_epoch_per_site"]
        logger.info(f"Optimization complete. Time: {time.time() - st}")
        out = {"res": r["res"] if num_epochs > 0 else Q0, "loss_per_epoch": lp.sum(axis=1),
               "loss_per_epoch_per_site": lp, "time_zero_grad": 0.0, "time_get_Q": 0.0,
               "time_compute_loss": time.time() - st, "time_cpu_loss_analysis": 0.0,
               "time_backwards": 0.0, "time_optimizer_step": 0.0, "time_initialize_tensors": 0.0}
        return {**out, **prof}
    theta = theta0.to(dev).requires_grad_(True)
    Theta = Theta0.to(dev).requires_grad_(True)
    if initialization is not None:
        with torch.no_grad():
            np.testing.assert_almost_equal(_site_Q(theta, Theta, upper)[0].cpu().numpy(),
                                           initialization, decimal=3)
    optimizer = torch.optim.Adam([theta, Theta], lr=0.1)
    prof["time_initialize_model"] = time.time() - st
    st = time.time()
    lpeps = torch.zeros(num_epochs, L, dtype=torch.float64, device=dev)
    loss_best = torch.full((L,), float("inf"), dtype=torch.float64, device=dev)
    with torch.no_grad():
        Qs_best = _site_Q(theta, Theta, upper)[0].clone()
    prof["time_initialize_tensors"] = time.time() - st
    t_loss = t_back = t_step = 0.0
    st_all = time.time()
    try:
        for epoch in range(num_epochs):
            optimizer.zero_grad()
            Q, pi = _site_Q(theta, Theta, upper)
            st = time.time()
            per_site = bank_loss(Q, pi, bank, normalize=True)
            loss = per_site.sum()
            t_loss += time.time() - st
            with torch.no_grad():
                better = per_site < loss_best
                loss_best = torch.where(better, per_site, loss_best)
                Qs_best = torch.where(better.view(-1, 1, 1), Q, Qs_best)
                lpeps[epoch] = per_site
            st = time.time()
            loss.backward()
            t_back += time.time() - st
            st = time.time()
            optimizer.step()
            t_step += time.time() - st
        torch.cuda.synchronize()
    finally:
        bank.close()
    logger.info(f"Optimization complete. Time: {time.time() - st_all}")
    lpeps_np = lpeps.cpu().numpy()
    res = {
        "res": Qs_best.cpu().numpy(),
        "loss_per_epoch": lpeps_np.sum(axis=1),
        "loss_per_epoch_per_site": lpeps_np,
        "time_zero_grad": 0.0, "time_get_Q": 0.0, "time_compute_loss": t_loss,
        "time_cpu_loss_analysis": 0.0, "time_backwards": t_back, "time_optimizer_step": t_step,
    }
    return {**res, **prof}
