#!/usr/bin/env python3
"""Golden vectors for FastCherries' branch-length / site-rate estimation.

  * the known answers of the reference's own tests
    (cherryml/phylogeny_estimation/FastCherries/tests/test_branch_length_estimation.cpp) as data:
    cherries, grid, rate categories, lengths -> expected indices; their lg.txt as an array;
  * random families evaluated by THE REFERENCE ITSELF, compiled by oracle/Makefile into
    oracle/_ref/libref_ble.so: get_branch_lengths, get_site_rates, ble;
  * the SiteRM site-rate gather evaluated by the reference's Cython module
    (cherryml/_siterm/fast_site_rates.pyx, built in the /tmp scratch copy).

Writes tests/golden/ble.npz.  Usage: python tests/golden/make_golden_ble.py (build container only)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)
FC = "/root/reference/cherryml/phylogeny_estimation/FastCherries"


def main():
    from oracle import ble_oracle as bo
    assert bo.ref_available(), "run `make -C oracle` first"
    Q = np.loadtxt(os.path.join(FC, "tests", "lg.txt"))
    grid129 = np.array([0.03 * 1.1 ** i for i in range(-64, 65)])
    # the test file prints its grid with 16-17 digits; take it from there (data), not from a formula
    src = open(os.path.join(FC, "tests", "test_branch_length_estimation.cpp")).read()
    lines = [ln for ln in src.split("\n") if "quantization_points = {" in ln]
    grids = [np.array([float(x) for x in ln[ln.index("{") + 1:ln.index("}")].split(",")]) for ln in lines]
    grid_bl, grid_sr = grids[0], grids[1]
    assert len(grid_bl) == 134 and len(grid_sr) == 129 and np.allclose(grid_sr, grid129, rtol=1e-12)
    out = dict(Q=Q, grid_bl=grid_bl, grid_sr=grid_sr)

    # ---- known answers (test_branch_lengths1-3, test_get_site_rates1-6)
    z8 = [0] * 8
    bl_cases = [
        ([[0,1,0,1,1,1,0,1],[0,0,0,0,0,1,0,1],[0,0,0,0,1,1,0,1],[0,1,0,1,1,1,0,1],[0,0,0,0,1,1,2,1]],
         [z8, z8, z8, z8, z8], [103, 81, 90, 103, 109]),
        ([[0,1,0,1,1,1,0,1],[0,0,0,0,0,1,0,1],[0,0,0,0,1,0,0,1],[0,1,0,1,1,1,0,1],[0,0,0,0,1,1,2,1]],
         [[0,1,0,0,0,0,0,0],[0,1,0,0,3,0,2,0],[0,0,0,0,3,0,2,0],[0,0,0,0,0,1,5,0],[0,0,0,0,0,0,5,0]], [98, 110, 89, 103, 103]),
        ([[0,0,0,0,0,0,0,0],[0,0,0,0,0,1,0,1],[0,0,0,0,0,0,5,0],[2,3,0,1,0,1,0,0],[0,5,0,0,1,1,2,0]],
         [[0,1,0,0,0,0,0,0],[0,1,0,0,3,0,2,0],[0,0,0,0,3,0,2,0],[0,0,0,0,0,1,5,0],[0,0,0,0,0,0,5,0]], [71, 110, 79, 120, 94]),
    ]
    rates_bl, s2r_bl = np.array([0.25, 0.7, 1.7, 4.0]), np.array([0, 3, 0, 3, 3, 3, 0, 3])
    bank_bl = bo.ref_log_bank(Q, grid_bl, rates_bl)
    for k, (x, y, exp) in enumerate(bl_cases):
        x, y = np.array(x), np.array(y)
        got = bo.ref_get_branch_lengths(x, y, bank_bl, grid_bl, s2r_bl)
        assert list(got) == exp, (k, got, exp)
        out.update({f"bl{k}_x": x, f"bl{k}_y": y, f"bl{k}_expected": np.array(exp)})
    out.update(rates_bl=rates_bl, s2r_bl=s2r_bl)

    ident = list(range(20))
    def mut(pos_list):
        rows = []
        for pos in pos_list:
            r = list(ident)
            for p in pos:
                r[p] = 0
            rows.append(r)
        return rows
    sr_cases = [
        ([[0,1,0,1,1,1,0,1],[0,0,0,0,0,1,0,1],[0,0,0,0,1,1,0,1],[0,1,0,1,1,1,0,1],[0,0,0,0,1,1,1,1],
          [0,1,0,1,1,1,0,1],[0,0,0,0,0,1,0,1],[0,0,0,0,1,1,0,1],[0,1,0,1,1,1,0,1],[0,0,0,0,1,1,0,1]], [z8] * 10,
         [1, 3, 1, 3, 3, 3, 2, 3]),
        ([[0,1,0,0,1,1,0,0],[0,0,0,0,0,1,0,0],[0,0,0,0,1,1,0,0],[0,1,0,0,1,1,0,0],[0,0,0,0,1,0,1,1],
          [0,1,0,0,1,0,0,1],[0,0,0,0,0,0,0,1],[0,0,0,0,1,0,0,1],[0,1,0,0,1,0,0,1],[0,0,0,0,1,1,0,0]], [z8] * 10,
         [1, 3, 1, 1, 3, 3, 2, 3]),
        ([[0,1,0,0,1,1,0,2],[0,0,0,1,0,1,0,2],[0,0,0,1,1,1,0,2],[1,1,0,1,1,1,0,2],[0,0,0,0,1,0,1,1],
          [0,1,0,0,1,0,0,1],[1,0,0,0,0,0,0,1],[0,0,0,0,1,0,0,1],[0,1,0,0,1,0,0,1],[1,0,0,0,1,1,0,0]],
         [[1,0,0,0,0,0,0,2],[0,0,0,1,0,0,0,2],[0,0,0,1,0,0,0,2],[1,0,0,1,0,0,0,2],[0,0,0,0,0,0,0,1],
          [0,0,0,0,0,0,0,1],[1,0,0,0,0,0,0,1],[0,0,0,0,0,0,0,1],[0,0,0,0,0,0,0,1],[1,0,0,0,0,0,0,0]],
         [2, 3, 1, 1, 3, 3, 2, 1]),
        ([[0,1,0,0,1,1,0,2],[0,0,0,1,0,1,0,2],[5,5,5,5,5,5,0,2],[1,1,0,1,1,1,0,2],[0,0,0,0,1,0,1,1],
          [0,1,0,0,1,0,0,1],[1,0,0,0,0,0,0,1],[0,0,0,0,1,0,0,1],[0,1,0,0,1,0,0,1],[1,0,0,0,1,1,0,0]],
         [[1,1,0,0,1,0,0,2],[0,0,0,1,0,0,0,2],[0,5,0,1,5,0,0,2],[1,1,0,1,1,0,0,2],[0,0,0,0,1,0,0,1],
          [0,1,0,0,1,0,0,1],[1,0,0,0,0,0,0,1],[0,0,0,0,1,0,0,1],[0,1,0,0,1,0,0,1],[1,0,0,0,1,0,0,0]],
         [2, 1, 2, 2, 1, 3, 2, 1]),
        ([ident] * 10, [ident] * 10, [1] * 20),
        (mut([[8], [], [3, 13], [], [4], [7], [18], [3], [], [10]]), [ident] * 10,
         [1, 1, 1, 2, 2, 1, 1, 2, 2, 1, 2, 1, 1, 2, 1, 1, 1, 1, 2, 1]),
    ]
    rates_sr = np.array([0.25, 0.6, 1.7, 4.0])
    lengths_sr = np.array([60, 20, 40, 60, 20, 40, 60, 20, 40, 60, 20, 40])
    bank_sr = bo.ref_log_bank(Q, grid_sr, rates_sr)
    for k, (x, y, exp) in enumerate(sr_cases):
        x, y = np.array(x), np.array(y)
        got = bo.ref_get_site_rates(x, y, bank_sr, lengths_sr[:len(x)], bo.rate_priors(rates_sr))
        assert list(got) == exp, (k, got, exp)
        out.update({f"sr{k}_x": x, f"sr{k}_y": y, f"sr{k}_expected": np.array(exp)})
    out.update(rates_sr=rates_sr, lengths_sr=lengths_sr)

    # ---- random families through the compiled reference
    rng = np.random.default_rng(11)
    for k, (n, L, R) in enumerate([(40, 60, 4), (25, 130, 8), (7, 33, 3)]):
        rates = np.geomspace(1.0 / R, float(R), R)
        weights = np.cumsum(rng.dirichlet(np.full(R, 4.0)))
        weights[-1] = 1.0
        bank = bo.ref_log_bank(Q, grid_sr, rates)
        true_rate = rng.choice(R, size=L)
        anc = rng.integers(0, 20, size=(n, L))
        t_idx = rng.integers(30, 110, size=n)
        cx, cy = anc.copy(), anc.copy()
        for c in range(n):
            for s in range(L):
                P = np.exp(bank[t_idx[c], true_rate[s], anc[c, s]])
                cy[c, s] = rng.choice(20, p=P / P.sum())
        gaps = rng.random((2, n, L)) < 0.08
        cx[gaps[0]] = -1
        cy[gaps[1]] = -1
        seqs = np.concatenate([cx, cy])
        s2r = rng.integers(0, R, size=L)
        li = rng.integers(0, len(grid_sr), size=n)
        bl = bo.ref_get_branch_lengths(cx, cy, bank, grid_sr, s2r)
        sr = bo.ref_get_site_rates(cx, cy, bank, li, bo.rate_priors(rates))
        ble_len, ble_rate = bo.ref_ble(cx, cy, seqs, bank, grid_sr, rates, weights, 50)
        out.update({f"rnd{k}_x": cx, f"rnd{k}_y": cy, f"rnd{k}_rates": rates, f"rnd{k}_weights": weights,
                    f"rnd{k}_s2r": s2r, f"rnd{k}_li": li, f"rnd{k}_bl": bl, f"rnd{k}_sr": sr,
                    f"rnd{k}_ble_lengths": ble_len, f"rnd{k}_ble_rates": ble_rate})
        print("random family", k, "ble lengths", ble_len[:5], "rates", ble_rate[:5])

    # ---- SiteRM site-rate gather through the reference's Cython module
    from make_golden import _prepare_scratch
    _prepare_scratch()
    from cherryml._siterm.fast_site_rates import compute_optimal_site_rates
    n, L, R, Sg = 9, 14, 5, 21
    grid_rates = list(np.geomspace(0.2, 5.0, R))
    prior = list(rng.dirichlet(np.full(R, 3.0)))
    tens = np.log(rng.dirichlet(np.full(Sg, 0.7), size=(R, n, Sg)))
    cx, cy = rng.integers(0, Sg, size=(n, L)), rng.integers(0, Sg, size=(n, L))
    cherries = [(list(map(int, cx[c])), list(map(int, cy[c])), 0.1) for c in range(n)]
    got = compute_optimal_site_rates(L, cherries, np.ascontiguousarray(tens), grid_rates, prior)
    out.update(gather_x=cx, gather_y=cy, gather_tensor=tens, gather_grid=np.array(grid_rates),
               gather_prior=np.array(prior), gather_expected=np.array(got))
    np.savez_compressed(os.path.join(HERE, "ble.npz"), **out)
    print("wrote", os.path.join(HERE, "ble.npz"))


if __name__ == "__main__":
    main()
