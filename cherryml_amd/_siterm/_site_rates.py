"""SiteRM site-rate gather on the GPU: same call as the reference's Cython helper
`compute_optimal_site_rates` (cherryml/_siterm/fast_site_rates.pyx:8-47)."""
from typing import List

import numpy as np

from .. import _lib


def compute_optimal_site_rates(num_sites: int, cherries: list, log_mexps_tensor_w_gaps: np.ndarray,
                               site_rate_grid: List[float], site_rate_prior: List[float], device: int = 0) -> list:
    """cherries: [(x, y, t)] with x, y sequences of state indices (gaps already mapped to a state);
    log_mexps_tensor_w_gaps[rate, cherry, x, y].  Returns, per site, the rate of the grid with the
    largest log prior + summed log-likelihood (first maximum wins)."""
    tens = np.ascontiguousarray(log_mexps_tensor_w_gaps, dtype=np.float64)
    R, n, S, _ = tens.shape
    if n != len(cherries) or R != len(site_rate_grid) or R != len(site_rate_prior):
        raise ValueError("inconsistent shapes")
    cx = np.ascontiguousarray([c[0][:num_sites] for c in cherries], dtype=np.int8)
    cy = np.ascontiguousarray([c[1][:num_sites] for c in cherries], dtype=np.int8)
    lp = np.log(np.asarray(site_rate_prior, dtype=np.float64))
    best = np.zeros(num_sites, dtype=np.int32)
    rc = _lib.load().cb_site_rate_gather(device, S, R, n, num_sites, tens.ctypes.data, cx.ctypes.data, cy.ctypes.data,
                                         lp.ctypes.data, best.ctypes.data)
    _lib.check(rc, "cb_site_rate_gather")
    return [float(site_rate_grid[b]) for b in best]
