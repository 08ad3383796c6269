"""JTT-IPW closed-form initialiser (reference: cherryml/estimation/_jtt_ipw.py:32-125).
Host-side numpy like the reference's: O(B S^2), not on the hot path, but it is
the default initialisation of both pipelines (estimation_end_to_end/_cherry.py:353-367)."""
import logging
import os
import time
from typing import Optional

import numpy as np

from .. import caching
from ..io import read_count_matrices_arrays, read_mask_matrix, write_rate_matrix


def jtt_ipw_from_arrays(qtimes: np.ndarray, cmats: np.ndarray, mask: Optional[np.ndarray] = None,
                        use_ipw: bool = True, pseudocounts: float = 1e-8,
                        symmetrize_count_matrices: bool = True,
                        max_time: Optional[float] = None) -> np.ndarray:
    q = np.asarray(qtimes, dtype=np.float64)
    C = np.asarray(cmats, dtype=np.float64)
    if max_time is not None:
        keep = q <= max_time
        q, C = q[keep], C[keep]
    S = C.shape[-1]
    C = C + pseudocounts
    if symmetrize_count_matrices:  # a->b and b->a coalesced
        C = (C + np.swapaxes(C, 1, 2)) / 2.0
    if mask is not None:
        C = C * np.asarray(mask, dtype=np.float64)[None]
    hollow = 1.0 - np.eye(S)
    F = C.sum(axis=0)
    F_off = F * hollow
    ctp = F_off / F_off.sum(axis=1)[:, None]
    if use_ipw:
        mut = ((C * hollow[None]).sum(axis=2) / q[:, None]).sum(axis=0) / F.sum(axis=1)
    else:
        mut = F_off.sum(axis=1) / np.median(q) / F.sum(axis=1)
    res = mut[:, None] * ctp
    np.fill_diagonal(res, -mut)
    return res


def _normalized(Q: np.ndarray) -> np.ndarray:
    w, v = np.linalg.eig(Q.T)
    p = v[:, int(np.argmin(np.abs(w.real)))].real
    p = p / p.sum()
    return Q / (p @ -np.diag(Q))


@caching.cached_computation(output_dirs=["output_rate_matrix_dir"], write_extra_log_files=True)
def jtt_ipw(
    count_matrices_path: str,
    mask_path: Optional[str],
    use_ipw: bool,
    output_rate_matrix_dir: str,
    normalize: bool = False,
    max_time: Optional[float] = None,
    pseudocounts: float = 1e-8,
    symmetrize_count_matrices: bool = True,
) -> None:
    start = time.time()
    logging.getLogger(__name__).info("Starting")
    q, C, states = read_count_matrices_arrays(count_matrices_path)
    mask = read_mask_matrix(mask_path).to_numpy() if mask_path is not None else None
    res = jtt_ipw_from_arrays(q, C, mask, use_ipw, pseudocounts, symmetrize_count_matrices, max_time)
    if normalize:
        res = _normalized(res)
    write_rate_matrix(res, states, os.path.join(output_rate_matrix_dir, "result.txt"))
    with open(os.path.join(output_rate_matrix_dir, "profiling.txt"), "w") as f:
        f.write(f"Total time: {time.time() - start} seconds\n")
