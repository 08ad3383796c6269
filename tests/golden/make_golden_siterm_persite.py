#!/usr/bin/env python3
"""Goldens for the reference's PER-SITE SiteRM path (`use_vectorized_cherryml_implementation=False`,
_siterm/_site_specific_rate_matrix.py:659-684): every site with counts goes through
`_quantized_transitions_mle` (:43-84) = RateMatrixLearner, "pande_reversible", initialisation
Q0 * rate_l, Adam lr 0.1, normalised loss, best iterate.  Made by RUNNING THE REFERENCE on the inputs
already committed in tests/golden/siterm_assembly.npz (the per-site count tensors the reference's own
assembly produced): as is (float32) and through its `train_quantization` in float64.

Written: tests/golden/siterm_persite.npz with <case>_res_f32 / <case>_res_f64 [L,S,S] (sites without
counts hold the prior Q0 * rate_l, as :655-658).   Usage: python tests/golden/make_golden_siterm_persite.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _prepare_scratch, _traj_reference_f64  # noqa: E402

CASES = ["t2_eq_cherry", "t2_some_missing", "rand_cherry"]


def main():
    z = dict(np.load(os.path.join(HERE, "siterm_assembly.npz"), allow_pickle=False))
    _prepare_scratch()
    import pandas as pd
    import torch
    import cherryml._siterm._site_specific_rate_matrix as M
    from cherryml.estimation._ratelearn.rate import RateMatrix
    from cherryml.estimation._ratelearn.trainer import train_quantization
    out = {}
    for c in CASES:
        counts, times, init = z[c + "_counts"], z[c + "_times"], z[c + "_init"]
        alphabet = [str(a) for a in z[c + "_alphabet"]]
        E = int(z[c + "_epochs"])
        L, _, S, _ = counts.shape
        res32, res64 = np.array(init), np.array(init)
        for l in range(L):
            live = [b for b in range(counts.shape[1]) if counts[l, b].sum() > 0]   # :667 the same filter
            if not live:
                continue
            cm = [(float(times[l, b]), pd.DataFrame(counts[l, b], index=alphabet, columns=alphabet)) for b in live]
            res32[l] = M._quantized_transitions_mle(count_matrices=cm, initialization=init[l], learning_rate=1e-1,
                                                    num_epochs=E, do_adam=True, loss_normalization=True,
                                                    return_best_iter=True,
                                                    rate_matrix_parameterization="pande_reversible").to_numpy()
            _, o64 = _traj_reference_f64(torch, RateMatrix, train_quantization, times[l, live], counts[l, live],
                                         np.ones((S, S)), init[l], E)
            res64[l] = o64["Q_best"]
        out[c + "_res_f32"], out[c + "_res_f64"] = res32, res64
        print(c, "sites", L, "max |f32 - f64|", np.abs(res32 - res64).max())
    np.savez_compressed(os.path.join(HERE, "siterm_persite.npz"), **out)
    print("wrote siterm_persite.npz")


if __name__ == "__main__":
    main()
