"""Python face of cb_ble_* (csrc/ble.hip.h).  Reference:
cherryml/phylogeny_estimation/FastCherries/branch_length_estimation.cpp (get_branch_lengths :60-103,
get_site_rates :105-144, ble :146-241) and io_helpers.cpp:150-174 (the log-transition bank).
Sequences are integer arrays [n, L] with -1 for gaps / unknown states."""
from typing import Optional, Sequence, Tuple

import numpy as np

from .. import _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i8(a):
    a = np.asarray(a)
    if a.size and (a.min() < -1 or a.max() > 126):
        raise ValueError("state codes must be in [-1, 126]")
    return np.ascontiguousarray(a, dtype=np.int8)


def rate_priors(rate_categories) -> np.ndarray:
    """log density of Gamma(shape 3, rate 3) up to a constant (:199-203)."""
    r = _f64(rate_categories)
    return 2.0 * np.log(r) - 3.0 * r


def compute_log_transition_matrices(Q, quantization_points, rate_categories, device: int = 0,
                                    stationary_distribution: Optional[np.ndarray] = None) -> np.ndarray:
    """log expm(t_i * rate_r * Q) for every grid point and rate category: [T, R, S, S]
    (`read_rate_compute_log_transition_matrices`, io_helpers.cpp:150-174), computed by the expm bank
    of this library: spectral kernels when Q is reversible w.r.t. `stationary_distribution` (pass it
    to assert that), else the general scaling-and-squaring kernels."""
    Q, grid, rates = _f64(Q), _f64(quantization_points), _f64(rate_categories)
    S = Q.shape[0]
    out = np.empty((grid.size, rates.size, S, S))
    pi = None
    if stationary_distribution is not None:
        pi = _f64(stationary_distribution)
        flux = pi[:, None] * Q
        if not np.allclose(flux, flux.T, rtol=1e-9, atol=1e-12 * np.abs(flux).max()):
            raise ValueError("Q is not reversible with respect to the given stationary distribution")
    rc = _lib.load().cb_ble_log_bank(device, S, grid.size, rates.size, Q.ctypes.data,
                                     None if pi is None else pi.ctypes.data, grid.ctypes.data,
                                     rates.ctypes.data, out.ctypes.data)
    _lib.check(rc, "cb_ble_log_bank")
    return out


def _shapes(cx, cy, logP):
    cx, cy, logP = _i8(cx), _i8(cy), _f64(logP)
    if cx.shape != cy.shape or cx.ndim != 2 or logP.ndim != 4 or logP.shape[2] != logP.shape[3]:
        raise ValueError("cherries must be two [n, L] arrays and the bank [T, R, S, S]")
    return cx, cy, logP


def branch_lengths(cx, cy, log_transition_matrices, site_to_rate_index, device: int = 0) -> np.ndarray:
    """`get_branch_lengths` (:60-103): grid index of the ML total length of every cherry."""
    cx, cy, logP = _shapes(cx, cy, log_transition_matrices)
    T, R, S, _ = logP.shape
    n, L = cx.shape
    s2r = np.ascontiguousarray(site_to_rate_index, dtype=np.int32)
    out = np.zeros(n, dtype=np.int32)
    rc = _lib.load().cb_ble_branch_lengths(device, S, T, R, logP.ctypes.data, cx.ctypes.data, cy.ctypes.data, n, L,
                                           s2r.ctypes.data, out.ctypes.data)
    _lib.check(rc, "cb_ble_branch_lengths")
    return out


def site_rates(cx, cy, log_transition_matrices, lengths_index, priors, device: int = 0) -> np.ndarray:
    """`get_site_rates` (:105-144): rate-category index of every site."""
    cx, cy, logP = _shapes(cx, cy, log_transition_matrices)
    T, R, S, _ = logP.shape
    n, L = cx.shape
    li = np.ascontiguousarray(lengths_index, dtype=np.int32)
    pr = _f64(priors)
    out = np.zeros(L, dtype=np.int32)
    rc = _lib.load().cb_ble_site_rates(device, S, T, R, logP.ctypes.data, cx.ctypes.data, cy.ctypes.data, n, L,
                                       li.ctypes.data, pr.ctypes.data, out.ctypes.data)
    _lib.check(rc, "cb_ble_site_rates")
    return out


def estimate_branch_lengths_and_site_rates(cx, cy, all_sequences, log_transition_matrices,
                                           quantization_points: Sequence[float],
                                           rate_categories: Sequence[float],
                                           weights_for_initial_site_rates: Sequence[float],
                                           max_iters: int, device: int = 0, profile: Optional[dict] = None
                                           ) -> Tuple[np.ndarray, np.ndarray]:
    """`ble` (:146-241): (cherry lengths [n], site rates [L]) as values of the grid / categories.
    `profile` (a dict) receives `iterations` and `kernel_ms` (GPU time of the ascent, HIP events)."""
    cx, cy, logP = _shapes(cx, cy, log_transition_matrices)
    T, R, S, _ = logP.shape
    n, L = cx.shape
    seqs = _i8(all_sequences)
    grid, rates, w = _f64(quantization_points), _f64(rate_categories), _f64(weights_for_initial_site_rates)
    if seqs.ndim != 2 or seqs.shape[1] != L or grid.size != T or rates.size != R or w.size != R:
        raise ValueError("inconsistent shapes")
    li, ri = np.zeros(n, dtype=np.int32), np.zeros(L, dtype=np.int32)
    import ctypes
    iters, ms = ctypes.c_int(0), ctypes.c_double(0.0)
    rc = _lib.load().cb_ble(device, S, T, R, logP.ctypes.data, cx.ctypes.data, cy.ctypes.data, n, L, seqs.ctypes.data,
                            seqs.shape[0], rates.ctypes.data, w.ctypes.data, int(max_iters), li.ctypes.data,
                            ri.ctypes.data, ctypes.addressof(iters), ctypes.addressof(ms) if profile is not None else None)
    _lib.check(rc, "cb_ble")
    if profile is not None:
        profile["iterations"], profile["kernel_ms"] = iters.value, ms.value
    return grid[li], rates[ri]


def estimate_branch_lengths_and_site_rates_batch(families, log_transition_matrices, quantization_points: Sequence[float],
                                                 rate_categories: Sequence[float],
                                                 weights_for_initial_site_rates: Sequence[float], max_iters: int,
                                                 device: int = 0, profile: Optional[dict] = None):
    """`ble` (:146-241) for MANY families in one call (`cb_ble_batch`): `families` = [(cx, cy, all_sequences), ...]
    with per-family shapes [n_f, L_f], [n_f, L_f], [n_seqs_f, L_f]; the bank `log_transition_matrices` is shared
    (one rate matrix for the whole stage, as in the reference's `fast_cherries`) and uploaded once.  Returns
    [(cherry lengths [n_f], site rates [L_f]), ...], family by family what
    `estimate_branch_lengths_and_site_rates` returns."""
    import ctypes
    if not families:
        return []
    logP = _f64(log_transition_matrices)
    if logP.ndim != 4 or logP.shape[2] != logP.shape[3]:
        raise ValueError("log_transition_matrices must be [T,R,S,S]")
    T, R, S, _ = logP.shape
    grid, rates, w = _f64(quantization_points), _f64(rate_categories), _f64(weights_for_initial_site_rates)
    if grid.size != T or rates.size != R or w.size != R:
        raise ValueError("inconsistent shapes")
    xs, ys, ss, ns, Ls, nseq = [], [], [], [], [], []
    for cx, cy, seqs in families:
        cx, cy, seqs = _i8(cx), _i8(cy), _i8(seqs)
        if cx.ndim != 2 or cx.shape != cy.shape or seqs.ndim != 2 or seqs.shape[1] != cx.shape[1]:
            raise ValueError("every family needs cx, cy [n, L] and all_sequences [n_seqs, L]")
        xs.append(cx.reshape(-1)); ys.append(cy.reshape(-1)); ss.append(seqs.reshape(-1))
        ns.append(cx.shape[0]); Ls.append(cx.shape[1]); nseq.append(seqs.shape[0])
    X, Y, A = (np.ascontiguousarray(np.concatenate(v)) for v in (xs, ys, ss))
    n_arr, L_arr, q_arr = (np.ascontiguousarray(v, dtype=np.int32) for v in (ns, Ls, nseq))
    li, ri = np.zeros(int(n_arr.sum()), dtype=np.int32), np.zeros(int(L_arr.sum()), dtype=np.int32)
    it = np.zeros(len(families), dtype=np.int32)
    ms = ctypes.c_double(0.0)
    rc = _lib.load().cb_ble_batch(device, S, T, R, logP.ctypes.data, len(families), n_arr.ctypes.data, L_arr.ctypes.data,
                                  X.ctypes.data, Y.ctypes.data, A.ctypes.data, q_arr.ctypes.data, rates.ctypes.data,
                                  w.ctypes.data, int(max_iters), li.ctypes.data, ri.ctypes.data, it.ctypes.data,
                                  ctypes.addressof(ms) if profile is not None else None)
    _lib.check(rc, "cb_ble_batch")
    if profile is not None:
        profile["iterations"], profile["kernel_ms"] = it.tolist(), ms.value
    out, on, oL = [], 0, 0
    for n, L in zip(ns, Ls):
        out.append((grid[li[on:on + n]], rates[ri[oL:oL + L]]))
        on += n
        oL += L
    return out


class BleBank:
    """The log-transition bank of ONE rate matrix resident on the device, and FastCherries' coordinate ascent on it family after
    family (`cb_ble_bank_create` / `cb_ble_bank_run`).  The reference computes the bank once per run and calls `ble` per family
    (FastCherries main.cpp; _fast_cherries.py:191); `estimate_branch_lengths_and_site_rates` follows the per-family signature and
    re-uploads the 8 MB bank, re-allocates its workspace and passes over every sequence byte three times on the host on every
    call -- 15 ms around 1 ms of kernels for a 2048-cherry family.  Here: `bank = BleBank(logP, grid, rates)` once, then
    `bank.estimate(cx, cy, all_sequences, weights, max_iters)` per family, same results bit for bit."""

    def __init__(self, log_transition_matrices, quantization_points: Sequence[float], rate_categories: Sequence[float],
                 device: int = 0):
        import ctypes
        logP = _f64(log_transition_matrices)
        if logP.ndim != 4 or logP.shape[2] != logP.shape[3]:
            raise ValueError("log_transition_matrices must be [T,R,S,S]")
        self.grid, self.rates = _f64(quantization_points), _f64(rate_categories)
        T, R, S, _ = logP.shape
        if self.grid.size != T or self.rates.size != R:
            raise ValueError("inconsistent shapes")
        self.T, self.R, self.S, self.device = T, R, S, int(device)
        self._h = ctypes.c_void_p()
        _lib.check(_lib.load().cb_ble_bank_create(self.device, S, T, R, logP.ctypes.data, self.rates.ctypes.data, ctypes.byref(self._h)),
                   "cb_ble_bank_create")

    @classmethod
    def from_rate_matrix(cls, Q, quantization_points, rate_categories, device: int = 0, stationary_distribution=None) -> "BleBank":
        logP = compute_log_transition_matrices(Q, quantization_points, rate_categories, device=device,
                                               stationary_distribution=stationary_distribution)
        return cls(logP, quantization_points, rate_categories, device=device)

    def estimate(self, cx, cy, all_sequences, weights_for_initial_site_rates: Sequence[float], max_iters: int,
                 profile: Optional[dict] = None) -> Tuple[np.ndarray, np.ndarray]:
        """(cherry lengths [n], site rates [L]) as values of the grid / categories: `ble` (:146-241)"""
        import ctypes
        if self._h is None:
            raise _lib.CherryBankError("BleBank: closed")
        cx, cy = _i8(cx), _i8(cy)
        if cx.shape != cy.shape or cx.ndim != 2:
            raise ValueError("cherries must be two [n, L] arrays")
        n, L = cx.shape
        seqs = _i8(all_sequences)
        w = _f64(weights_for_initial_site_rates)
        if seqs.ndim != 2 or seqs.shape[1] != L or w.size != self.R:
            raise ValueError("inconsistent shapes")
        li, ri = np.zeros(n, dtype=np.int32), np.zeros(L, dtype=np.int32)
        iters, ms = ctypes.c_int(0), ctypes.c_double(0.0)
        rc = _lib.load().cb_ble_bank_run(self._h, cx.ctypes.data, cy.ctypes.data, n, L, seqs.ctypes.data, seqs.shape[0], w.ctypes.data,
                                         int(max_iters), li.ctypes.data, ri.ctypes.data, ctypes.addressof(iters),
                                         ctypes.addressof(ms) if profile is not None else None)
        _lib.check(rc, "cb_ble_bank_run")
        if profile is not None:
            profile["iterations"], profile["kernel_ms"] = iters.value, ms.value
        return self.grid[li], self.rates[ri]

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.load().cb_ble_bank_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
