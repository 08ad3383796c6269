"""ORACLE -- TEST INFRASTRUCTURE ONLY (never imported by the product path).

CPU restatement of FastCherries' branch-length / site-rate estimation,
cherryml/phylogeny_estimation/FastCherries:

  log_bank                  io_helpers.cpp:150-174   log expm(t_i * rate_r * Q)   [T,R,S,S]
  initial_site_rates        branch_length_estimation.cpp:10-58   diversity bins
  get_branch_lengths        :60-103   per cherry, bisection on the grid index
  get_site_rates            :105-144  per site, bisection on the rate category (with the Gamma(3,1/3) log prior)
  ble                       :146-241  coordinate ascent of the two until the lengths stop moving
and of the SiteRM site-rate gather, cherryml/_siterm/fast_site_rates.pyx:8-47
  compute_optimal_site_rates.

Pinned (tests/test_oracle_golden.py::test_ble_*) on the known answers of the reference's own
tests/test_branch_length_estimation.cpp and on outputs of the reference itself compiled into
oracle/_ref/libref_ble.so (oracle/Makefile, oracle/ref_ble_shim.cpp); `ref_*` below call that
library when it is present.  Sequences are int arrays, -1 = gap / unknown."""
import ctypes
import os
import tempfile
from typing import Optional

import numpy as np
from scipy.linalg import expm


def log_bank(Q: np.ndarray, grid, rates) -> np.ndarray:
    Q = np.asarray(Q, dtype=np.float64)
    out = np.empty((len(grid), len(rates)) + Q.shape)
    for i, t in enumerate(grid):
        for r, rate in enumerate(rates):
            out[i, r] = np.log(expm(t * rate * Q))
    return out


def initial_site_rates(all_seqs: np.ndarray, weights, S: int) -> np.ndarray:
    """:10-58.  Sites sorted by the number of differing sequence pairs (ties: site index); site
    i of that order gets category rc, rc advancing while i >= round(weights[rc] * L)."""
    n, L = all_seqs.shape
    counts = np.zeros((L, S), dtype=np.int64)
    for j in range(L):
        col = all_seqs[:, j]
        col = col[col != -1]
        counts[j] = np.bincount(col, minlength=S)[:S]
    non_missing = counts.sum(axis=1)
    total = ((non_missing[:, None] - counts) * counts).sum(axis=1)
    order = sorted(range(L), key=lambda j: (int(total[j]), j))
    w = [int(round(x * L)) for x in weights]
    out = np.zeros(L, dtype=np.int32)
    rc = 0
    for i in range(L):
        if rc < len(w) and i >= w[rc]:
            rc += 1
        out[order[i]] = rc
    return out


def get_branch_lengths(cx, cy, logP, site_to_rate) -> np.ndarray:
    n, L = cx.shape
    T = logP.shape[0]
    out = np.zeros(n, dtype=np.int32)
    for c in range(n):
        valid = [i for i in range(L) if cx[c, i] != -1 and cy[c, i] != -1]
        low, high = 0, T - 1
        while low < high:
            mid = low + (high - low) // 2
            ll_m = ll_m1 = 0.0
            for i in valid:
                x, y, r = cx[c, i], cy[c, i], site_to_rate[i]
                ll_m += logP[mid, r, x, y] + logP[mid, r, y, x]
                ll_m1 += logP[mid + 1, r, x, y] + logP[mid + 1, r, y, x]
            if ll_m > ll_m1:
                high = mid
            else:
                low = mid + 1
        out[c] = low
    return out


def get_site_rates(cx, cy, logP, lengths_index, priors) -> np.ndarray:
    n, L = cx.shape
    R = len(priors)
    out = np.zeros(L, dtype=np.int32)
    for s in range(L):
        valid = [i for i in range(n) if cx[i, s] != -1 and cy[i, s] != -1]
        low, high = 0, R - 1
        while low < high:
            mid = low + (high - low) // 2
            ll_m, ll_m1 = priors[mid], priors[mid + 1]
            for i in valid:
                x, y, t = cx[i, s], cy[i, s], lengths_index[i]
                ll_m += logP[t, mid, x, y] + logP[t, mid, y, x]
                ll_m1 += logP[t, mid + 1, x, y] + logP[t, mid + 1, y, x]
            if ll_m > ll_m1:
                high = mid
            else:
                low = mid + 1
        out[s] = low
    return out


def rate_priors(rates) -> np.ndarray:
    """:199-203: log density of Gamma(shape 3, rate 3) up to a constant."""
    return np.array([2.0 * np.log(r) - 3.0 * r for r in rates])


def ble(cx, cy, all_seqs, logP, grid, rates, weights, max_iters: int):
    S = logP.shape[2]
    site_to_rate = initial_site_rates(all_seqs, weights, S)
    lengths = get_branch_lengths(cx, cy, logP, site_to_rate)
    priors = rate_priors(rates)
    match = False
    while not match and max_iters:
        site_to_rate = get_site_rates(cx, cy, logP, lengths, priors)
        new = get_branch_lengths(cx, cy, logP, site_to_rate)
        match = bool(np.array_equal(new, lengths))
        lengths = new
        max_iters -= 1
    return np.asarray(grid)[lengths], np.asarray(rates)[site_to_rate], lengths, site_to_rate


def compute_optimal_site_rates(cx, cy, log_mexps, site_rate_grid, site_rate_prior) -> np.ndarray:
    """fast_site_rates.pyx:8-47: log_mexps[rate, cherry, x, y]; per site the rate with the largest
    log prior + sum over cherries (first maximum wins).  States index the tensor directly (the
    caller maps gaps to an extra state)."""
    n, L = cx.shape
    out = np.zeros(L)
    for s in range(L):
        best, best_ll = None, None
        for r, rate in enumerate(site_rate_grid):
            ll = np.log(site_rate_prior[r])
            for c in range(n):
                ll += log_mexps[r, c, cx[c, s], cy[c, s]]
            if best_ll is None or ll > best_ll:
                best, best_ll = rate, ll
        out[s] = best
    return out


# ---------------------------------------------------------------- the compiled reference
_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libref_ble.so")


def ref_available() -> bool:
    return os.path.exists(_REF)


def _lib():
    return ctypes.CDLL(_REF)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def ref_log_bank(Q, grid, rates) -> np.ndarray:
    Q, grid, rates = _f64(Q), _f64(grid), _f64(rates)
    S = Q.shape[0]
    out = np.empty((len(grid), len(rates), S, S))
    with tempfile.NamedTemporaryFile("w", suffix=".txt", delete=False) as f:
        for row in Q:
            f.write(" ".join(repr(float(v)) for v in row) + "\n")
        path = f.name
    try:
        _lib().ref_log_bank(path.encode(), S, grid.ctypes.data_as(ctypes.c_void_p), len(grid),
                            rates.ctypes.data_as(ctypes.c_void_p), len(rates), out.ctypes.data_as(ctypes.c_void_p))
    finally:
        os.unlink(path)
    return out


def ref_get_branch_lengths(cx, cy, logP, grid, site_to_rate) -> np.ndarray:
    cx, cy, logP, grid, s2r = _i32(cx), _i32(cy), _f64(logP), _f64(grid), _i32(site_to_rate)
    n, L = cx.shape
    T, R, S, _ = logP.shape
    out = np.zeros(n, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _lib().ref_get_branch_lengths(S, T, R, p(logP), p(cx), p(cy), n, L, p(grid), p(s2r), p(out))
    return out


def ref_get_site_rates(cx, cy, logP, lengths_index, priors) -> np.ndarray:
    cx, cy, logP, li, pr = _i32(cx), _i32(cy), _f64(logP), _i32(lengths_index), _f64(priors)
    n, L = cx.shape
    T, R, S, _ = logP.shape
    out = np.zeros(L, dtype=np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _lib().ref_get_site_rates(S, T, R, p(logP), p(cx), p(cy), n, L, p(li), p(pr), p(out))
    return out


def ref_ble(cx, cy, all_seqs, logP, grid, rates, weights, max_iters: int):
    cx, cy, seqs, logP = _i32(cx), _i32(cy), _i32(all_seqs), _f64(logP)
    grid, rates, weights = _f64(grid), _f64(rates), _f64(weights)
    n, L = cx.shape
    T, R, S, _ = logP.shape
    lo, ro = np.zeros(n), np.zeros(L)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
    _lib().ref_ble(S, T, R, p(logP), p(cx), p(cy), n, L, p(seqs), seqs.shape[0], p(grid), p(rates), p(weights),
                   int(max_iters), p(lo), p(ro))
    return lo, ro
