"""Parameterisation theta -> Q (reference: cherryml/estimation/_ratelearn/rate.py).

Same constructor arguments, parameter names (`upper_diag`, `lower_diag`, `_pi`),
modes and error behaviour as the reference's `RateMatrix`; arithmetic in
float64 (the reference's float32 is the only intended difference)."""
import warnings
from typing import Optional

import numpy as np
import torch
import torch.nn as nn

REVERSIBLE_MODES = ("pande_reversible", "stationary_reversible")
ALL_MODES = ("default", "stationary", "stationary_reversible", "pande", "pande_reversible")


def solve_stationery_dist(rate_matrix: np.ndarray) -> np.ndarray:
    """Left null vector (rate.py:10-18; the spelling is the reference's)."""
    w, v = np.linalg.eig(np.asarray(rate_matrix).T)
    p = v[:, int(np.argmin(np.abs(w.real)))].real
    return p / p.sum()


class RateMatrix(nn.Module):
    def __init__(self, num_states, mode, mask: torch.Tensor, pi=None, pi_requires_grad=False,
                 initialization: Optional[np.ndarray] = None, dtype=torch.float64):
        super().__init__()
        if mode not in ALL_MODES:
            raise ValueError(f"Unknown rate matrix parameterization: {mode}")
        self.num_states = int(num_states)
        self.mode = mode
        S = self.num_states
        half = S * (S - 1) // 2
        if pi is not None:
            assert pi.ndim == 1
            self._pi = nn.Parameter(torch.log(pi.to(dtype)), requires_grad=pi_requires_grad)
        # float32 draw widened: the reference's parameters are float32 (rate.py:51-53)
        self.upper_diag = nn.Parameter((0.01 * torch.randn(half)).to(dtype))
        if mode in ("default", "stationary", "pande"):
            self.lower_diag = nn.Parameter((0.01 * torch.randn(half)).to(dtype))
        self.register_buffer("mask", torch.as_tensor(mask).to(dtype))
        iu = torch.triu_indices(S, S, offset=1)
        self.register_buffer("_iu", iu)
        self.register_buffer("_il", torch.tril_indices(S, S, offset=-1))

        if initialization is not None and mode == "pande_reversible":
            init = np.asarray(initialization, dtype=np.float64)
            p0 = solve_stationery_dist(init)
            if np.any(np.abs(p0) < 1e-8):
                raise ValueError("Stationary distribution of initialization is degenerate.")
            if np.any(np.abs(self.mask.cpu().numpy() * init - init) > 1e-8):
                raise ValueError("initialization not compatible with mask")
            root = np.sqrt(p0)
            sym = (root[:, None] * init) / root[None, :]
            try:
                np.testing.assert_almost_equal(sym, sym.T, decimal=4)
            except AssertionError:
                warnings.warn("S and its transpose are not almost equal up to 4 decimal places.")
            rows, cols = iu[0].numpy(), iu[1].numpy()
            with np.errstate(divide="ignore"):
                logits = np.log(np.exp(sym[rows, cols]) - 1.0)  # softplus^-1, -inf if masked
            with torch.no_grad():
                self._pi.copy_(torch.tensor(np.log(p0), dtype=dtype))
                self.upper_diag.copy_(torch.tensor(logits, dtype=dtype))
            np.testing.assert_almost_equal(self().detach().cpu().numpy(), init, decimal=3)
        elif initialization is not None:
            raise ValueError(f"Parameter initialization not implemented for mode {mode}")

    # -- pieces -----------------------------------------------------------------
    def stationary(self) -> torch.Tensor:
        return torch.softmax(self._pi, dim=-1)

    def is_reversible(self) -> bool:
        """True when Q is reversible w.r.t. softmax(_pi) (=> symmetric eigen path)."""
        if self.mode not in REVERSIBLE_MODES:
            return False
        return bool(torch.equal(self.mask, self.mask.T))

    def _offdiag(self, with_lower: bool) -> torch.Tensor:
        S = self.num_states
        R = torch.zeros(S, S, dtype=self.upper_diag.dtype, device=self.upper_diag.device)
        R = R.index_put((self._iu[0], self._iu[1]), nn.functional.softplus(self.upper_diag))
        if with_lower:
            R = R.index_put((self._il[0], self._il[1]), nn.functional.softplus(self.lower_diag))
        else:
            R = R + R.T
        return R * self.mask

    def forward(self) -> torch.Tensor:
        mode = self.mode
        if mode == "default":
            R = self._offdiag(True)
            return R - torch.diag(R.sum(1))
        if mode in ("stationary_reversible", "stationary"):
            R = self._offdiag(mode == "stationary")
            pi = self.stationary()
            R = R + torch.diag(-(R @ pi) / pi)
            return R * pi[None, :]
        # pande / pande_reversible: Q = D^-1/2 R D^1/2 - diag(rowsum)
        R = self._offdiag(mode == "pande")
        root = self.stationary().sqrt()
        Q = R * (root[None, :] / root[:, None])
        return Q - torch.diag(Q.sum(1))
