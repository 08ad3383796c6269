"""JTT-IPW closed-form initialiser (reference: cherryml/estimation/_jtt_ipw.py:32-125), the default
initialisation of both pipelines (estimation_end_to_end/_cherry.py:353-367).

Everything that touches the [B, S, S] count tensor -- the two S x S sums `F = sum_b sym(C_b)` and
`R = sum_b sym(C_b) / t_b` the estimator is linear in -- is ONE streaming pass on the GPU
(`cb_jtt_ipw_stats`, csrc/counting.hip.h: 165 MB at 400 states); the closed form on top of them
(pseudocounts, mask, rates) is O(S^2) host arithmetic.  There is no CPU path for the tensor pass:
without the library or a GPU `jtt_ipw_statistics` raises `CherryBankError`."""
import logging
import os
import time
from typing import Optional, Tuple

import numpy as np

from .. import _lib, caching
from ..io import read_count_matrices_arrays, read_mask_matrix, write_rate_matrix


def jtt_ipw_statistics(qtimes: np.ndarray, cmats, unit: float = 1.0, symmetrize: bool = True,
                       device: int = 0) -> Tuple[np.ndarray, np.ndarray]:
    """(F, R) = (sum_b sym(C_b), sum_b sym(C_b) / t_b) * unit on the GPU.  `cmats` [B, S, S]: a numpy array (float64
    counts, or unsigned / signed 64-bit integers in units of `unit`) or a torch tensor resident on the device (float64 or
    int64: the histogram the counting kernels leave there)."""
    q = np.ascontiguousarray(np.asarray(qtimes, dtype=np.float64))
    B = q.shape[0]
    lib = _lib.load()
    if lib.cb_device_count() <= 0:
        raise _lib.CherryBankError("jtt_ipw: no HIP device visible; the count-tensor pass runs on the MI355X only")
    try:
        import torch
        is_tensor = isinstance(cmats, torch.Tensor)
    except ImportError:   # pragma: no cover
        is_tensor = False
    if is_tensor and cmats.is_cuda:
        import torch
        if cmats.dtype not in (torch.float64, torch.int64):
            raise ValueError("jtt_ipw_statistics: device counts must be float64 or int64")
        C = cmats.contiguous()
        S = int(C.shape[-1])
        assert C.numel() == B * S * S
        dev = C.device
        out = torch.empty((2, S, S), dtype=torch.float64, device=dev)
        tg = torch.from_numpy(q).to(dev)
        torch.cuda.synchronize(dev)   # the kernels run on HIP's default stream
        _lib.check(lib.cb_jtt_ipw_stats(dev.index or 0, S, B, C.data_ptr(), int(C.dtype == torch.float64), tg.data_ptr(),
                                        float(unit), int(bool(symmetrize)), _lib.CB_PTR_DEVICE, out[0].data_ptr(),
                                        out[1].data_ptr()), "cb_jtt_ipw_stats")
        torch.cuda.synchronize(dev)
        st = out.cpu().numpy()
        return st[0], st[1]
    C = np.asarray(cmats.cpu().numpy() if is_tensor else cmats)
    f64 = C.dtype.kind == "f"
    C = np.ascontiguousarray(C, dtype=np.float64 if f64 else np.uint64)
    S = C.shape[-1]
    assert C.shape == (B, S, S)
    F = np.empty((S, S))
    R = np.empty((S, S))
    _lib.check(lib.cb_jtt_ipw_stats(device, S, B, C.ctypes.data, int(f64), q.ctypes.data, float(unit),
                                    int(bool(symmetrize)), 0, F.ctypes.data, R.ctypes.data), "cb_jtt_ipw_stats")
    return F, R


def jtt_ipw_from_statistics(F_sum: np.ndarray, R_sum: np.ndarray, qtimes: np.ndarray, mask: Optional[np.ndarray] = None,
                            use_ipw: bool = True, pseudocounts: float = 1e-8) -> np.ndarray:
    """The estimator from F = sum_b sym(C_b) and R = sum_b sym(C_b) / t_b over the buckets `qtimes` (_jtt_ipw.py:66-110):
    the pseudocount enters every bucket once (and survives the symmetrisation unchanged), the mask multiplies entrywise."""
    q = np.asarray(qtimes, dtype=np.float64)
    S = F_sum.shape[0]
    m = np.ones((S, S)) if mask is None else np.asarray(mask, dtype=np.float64)
    hollow = 1.0 - np.eye(S)
    F = (np.asarray(F_sum, dtype=np.float64) + len(q) * pseudocounts) * m
    F_off = F * hollow
    ctp = F_off / F_off.sum(axis=1)[:, None]
    if use_ipw:
        R = (np.asarray(R_sum, dtype=np.float64) + pseudocounts * np.sum(1.0 / q)) * m
        mut = (R * hollow).sum(axis=1) / F.sum(axis=1)
    else:
        mut = F_off.sum(axis=1) / np.median(q) / F.sum(axis=1)
    res = mut[:, None] * ctp
    np.fill_diagonal(res, -mut)
    return res


def jtt_ipw_from_arrays(qtimes: np.ndarray, cmats, mask: Optional[np.ndarray] = None,
                        use_ipw: bool = True, pseudocounts: float = 1e-8,
                        symmetrize_count_matrices: bool = True,
                        max_time: Optional[float] = None) -> np.ndarray:
    q = np.asarray(qtimes, dtype=np.float64)
    if max_time is not None:
        keep = q <= max_time
        if not np.all(keep):
            idx = np.nonzero(keep)[0]
            q = q[idx]
            cmats = cmats[idx] if not hasattr(cmats, "is_cuda") else cmats[list(idx)]
    F, R = jtt_ipw_statistics(q, cmats, 1.0, symmetrize_count_matrices)
    return jtt_ipw_from_statistics(F, R, q, mask, use_ipw, pseudocounts)


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
