"""The host side of the time-basis bank (csrc/tbasis.hip.h, include/cherrybank.h: cb_time_basis) -- the interpolative
decomposition over the branch-length grid -- checked on the CPU against its definition, in numpy long double, at points of the
spectrum the builder did not sample.  No GPU: the builder is host code of libcherrybank.so."""
import numpy as np
import pytest

from cherryml_amd.bank import time_basis

LD = np.longdouble
GRID = 0.03 * 1.1 ** np.arange(-64, 65.0)          # the reference's default quantisation grid (129 branch lengths)


def _g(x):
    """phi2(x) / x^2 = (e^x - 1 - x) / x^2 in long double"""
    xs = np.where(np.abs(x) < 0.5, x, 0)
    p = np.ones_like(xs)
    for k in range(26, 2, -1):
        p = 1 + p * xs / k
    with np.errstate(all="ignore"):
        big = (np.expm1(x) - x) / (x * x)
    return np.where(np.abs(x) < 0.5, 0.5 * p, big)


@pytest.mark.parametrize("rho_max", [2.0, 8.64, 40.0])
def test_decomposition_reproduces_both_families_between_the_sample_points(rho_max):
    tb = time_basis(GRID, rho_max)
    B, t = len(GRID), GRID.astype(LD)
    ns, nd, ng, kind = tb["ns"], tb["nd"], tb["ng"], tb["kind"]
    small = kind < 0
    # the short-branch buckets are a prefix, the long-branch ones numbered in order, and the rule is t rho_max <= 8
    assert np.array_equal(small, GRID * rho_max <= 8.0) and np.array_equal(kind[~small], np.arange(nd))
    assert ns <= 24 and ng <= 40 and ns + nd < B and ng < B / 3
    rng = np.random.default_rng(7)
    mu = -LD(rho_max) * np.concatenate([rng.random(1500), 10.0 ** (-6 * rng.random(1500)), [0.0, 1.0]]).astype(LD)
    x = t[:, None] * mu[None, :]
    psi = mu[None, :] ** 2 * _g(x)                  # phi2(t mu) / t^2
    r_s = np.abs(psi[small] - tb["Ls"][small].astype(LD) @ psi[tb["skel_s"]]).max() / np.abs(psi).max()
    E = np.exp(x)
    Lg = tb["Lg"].astype(LD) * t[tb["skel_g"]][None, :] / t[:, None]      # (Lg carries t_b / t_skeleton)
    r_g = np.abs(E - Lg @ E[tb["skel_g"]]).max()
    print(f"rho_max {rho_max}: ns {ns} nd {nd} ng {ng}; residuals psi {float(r_s):.1e} (relative), exp {float(r_g):.1e}")
    # (the forward family to rounding -- the O(t^2) entries of P_b rest on it --, the gradient family to 5e-14: its skeleton is
    # complete when the largest remaining row is 2e-15 of the largest, csrc/tbasis.hip.h CB_TB_TOL_G)
    assert r_s < 1e-15 and r_g < 5e-14
    assert tb["residuals"][0] < 1e-15 * float(np.abs(psi).max()) and tb["residuals"][1] < 3e-14
    # a skeleton bucket is itself; interpolation weights stay of order one
    assert np.array_equal(tb["Ls"][tb["skel_s"]], np.eye(ns)) and np.abs(tb["Ls"]).max() < 4.0
    assert np.allclose(Lg[tb["skel_g"]].astype(np.float64), np.eye(ng), atol=0, rtol=0) and np.abs(Lg).max() < 4.0
    assert np.all(tb["Ls"][~small] == 0.0)


def test_rank_grows_slowly_with_the_spectral_bound():
    ranks = [(time_basis(GRID, r)["ns"], time_basis(GRID, r)["ng"]) for r in (1.0, 4.0, 16.0)]
    print(ranks)
    assert all(a[1] <= b[1] for a, b in zip(ranks, ranks[1:])) and ranks[-1][1] <= 34 and max(r[0] for r in ranks) <= 18


def test_short_and_odd_grids():
    tb = time_basis([0.1], 3.0)
    assert (tb["ns"], tb["nd"], tb["ng"]) == (1, 0, 1) and tb["Lg"][0, 0] == 1.0
    tb = time_basis([0.5, 0.5, 0.7], 3.0)           # a repeated branch length: its second copy is the first one exactly
    assert tb["ng"] == 2 and np.allclose(tb["Lg"][1], tb["Lg"][0])
    tb = time_basis(np.geomspace(1e-3, 50.0, 40), 2.0)   # every bucket with t rho_max > 8 keeps its own product
    assert tb["nd"] == int((np.geomspace(1e-3, 50.0, 40) * 2.0 > 8.0).sum())


def test_unsupported_inputs_are_refused():
    with pytest.raises(NotImplementedError):
        time_basis(GRID[::-1], 3.0)                 # not ascending
    with pytest.raises(NotImplementedError):
        time_basis(GRID, float("nan"))
    with pytest.raises(NotImplementedError):
        time_basis(np.geomspace(1e-9, 1e6, 400), 1e3)   # needs more than 40 skeleton buckets
