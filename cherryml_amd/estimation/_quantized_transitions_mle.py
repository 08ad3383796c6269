"""Stage function `quantized_transitions_mle` (reference:
cherryml/estimation/_quantized_transitions_mle.py:35-122): same keyword-only
signature, same output files (`result.txt`, `Q_best.txt`, `Q_last.txt`,
`Q_<2^k>.txt`, `df_res.txt`, `profiling.txt`), same return convention."""
import logging
import os
import tempfile
import time
from typing import Optional

import numpy as np

from .. import caching
from ..io import (read_count_matrices_arrays, read_mask_matrix, read_probability_distribution,
                  read_rate_matrix)
from ._ratelearn import RateMatrixLearner


@caching.cached_computation(
    output_dirs=["output_rate_matrix_dir"],
    exclude_args=["device", "OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS"],
    write_extra_log_files=True,
)
def quantized_transitions_mle(
    count_matrices_path: str,
    initialization_path: Optional[str],
    mask_path: Optional[str],
    output_rate_matrix_dir: Optional[str],
    stationary_distribution_path: Optional[str] = None,
    rate_matrix_parameterization: str = "pande_reversible",
    device: str = "cpu",
    learning_rate: float = 1e-1,
    num_epochs: int = 2000,
    do_adam: bool = True,
    loss_normalization: bool = True,
    OMP_NUM_THREADS: Optional[int] = 1,
    OPENBLAS_NUM_THREADS: Optional[int] = 1,
    return_best_iter: bool = True,
    bank_dtype: str = "f64",
):
    """`bank_dtype` is the one argument the reference does not have: "f64" (default; parity with the
    reference run in float64 to 1e-6) or "f32" -- the reference's own arithmetic (float32 matrix_exp,
    ratelearner.py:98,107) on the float32 MFMA, for more than 32 states."""
    start_time = time.time()
    logger = logging.getLogger(__name__)
    logger.info("Starting")
    assert device in ["cpu", "cuda"]
    q, C, states = read_count_matrices_arrays(count_matrices_path)
    stationary = (read_probability_distribution(stationary_distribution_path).to_numpy()
                  if stationary_distribution_path is not None else None)
    init = (read_rate_matrix(initialization_path).to_numpy()
            if initialization_path is not None else None)
    with tempfile.TemporaryDirectory() as tmp:
        mask2_path = None
        if mask_path is not None:  # hand the mask over in the learner's plain-matrix format
            mask2_path = os.path.join(tmp, "mask.txt")
            np.savetxt(mask2_path, read_mask_matrix(mask_path).to_numpy(), fmt="%d")
        learner = RateMatrixLearner(
            branches=[float(x) for x in q], mats=[C[b] for b in range(C.shape[0])], states=states,
            output_dir=output_rate_matrix_dir, stationnary_distribution=stationary,
            mask=mask2_path, rate_matrix_parameterization=rate_matrix_parameterization,
            device=device, initialization=init, bank_dtype=bank_dtype)
        learner.train(lr=learning_rate, num_epochs=num_epochs, do_adam=do_adam,
                      loss_normalization=loss_normalization, return_best_iter=return_best_iter)
    logger.info("Done!")
    with open(os.path.join(output_rate_matrix_dir, "profiling.txt"), "w") as f:
        f.write(f"Total time: {time.time() - start_time} seconds with "
                f"{OPENBLAS_NUM_THREADS} OPENBLAS_NUM_THREADS and {OMP_NUM_THREADS}"
                " OMP_NUM_THREADS\n")
