"""Stage functions `count_transitions` / `count_co_transitions` with the reference's
keyword signatures and output files (result.txt in the count-matrix format, profiling.txt).
Host: parsing + pairing (cherryml_amd.counting._host); device: cb_count_transitions /
cb_count_co_transitions (integer histogram, bit-exact).  With torch.distributed
initialised, families are dealt round-robin to the ranks (like the reference's MPI ranks,
_count_transitions.cpp:626) and the integer counts are all-reduced."""
import ctypes
import logging
import os
import time
from typing import List, Optional, Union

import numpy as np
import pandas as pd

from .. import _lib, caching
from ..io import write_count_matrices
from . import _host


class _Pair(ctypes.Structure):
    _fields_ = [("seq_a", ctypes.c_int64), ("seq_b", ctypes.c_int64), ("aux", ctypes.c_int64),
                ("n", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("len_a", ctypes.c_double), ("len_b", ctypes.c_double)]


PAIR_DTYPE = np.dtype([("seq_a", "<i8"), ("seq_b", "<i8"), ("aux", "<i8"), ("n", "<i4"),
                       ("reserved", "<i4"), ("len_a", "<f8"), ("len_b", "<f8")])
assert PAIR_DTYPE.itemsize == ctypes.sizeof(_Pair)


def _my_families(families: List[str]) -> List[str]:
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return [f for i, f in enumerate(families) if i % dist.get_world_size() == dist.get_rank()]
    except ImportError:  # pragma: no cover
        pass
    return list(families)


def _all_reduce_counts(counts: np.ndarray) -> np.ndarray:
    try:
        import torch
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" \
                else torch.device("cpu")
            t = torch.from_numpy(counts.astype(np.int64)).to(dev)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            return t.cpu().numpy().astype(np.uint64)
    except ImportError:  # pragma: no cover
        pass
    return counts


def _device_index() -> int:
    try:
        import torch
        if torch.cuda.is_available():
            return torch.cuda.current_device()
    except ImportError:  # pragma: no cover
        pass
    return 0


def _normalise_mode(edge_or_cherry: str) -> str:
    if edge_or_cherry.startswith("cherry++__"):
        return "cherry++"
    if edge_or_cherry not in ("edge", "cherry", "cherry++"):
        raise ValueError(f"unknown edge_or_cherry: {edge_or_cherry}")
    return edge_or_cherry


def _gather(tree_dir, msa_dir, families, amino_acids, mode, aux_dir, aux_reader, length_float32: bool = False):
    """Concatenate the families: encoded sequences, per-family aux arrays, pair records."""
    seq_chunks, aux_chunks, pair_rows = [], [], []
    seq_off = 0
    aux_off = 0
    for fam in families:
        names, children, root = _host.read_tree_arrays(os.path.join(tree_dir, fam + ".txt"), length_float32)
        msa = _host.read_msa(os.path.join(msa_dir, fam + ".txt"))
        pairs = _host.build_pairs(children, root, mode)
        used = sorted({p[0] for p in pairs} | {p[1] for p in pairs})
        row_of = {u: r for r, u in enumerate(used)}
        codes = _host.encode_msa(msa, [names[u] for u in used], amino_acids) if used else \
            np.zeros((0, 0), dtype=np.int8)
        L = codes.shape[1] if used else 0
        # aux: the family's site rates / contact pairs; n_per_pair: how many of them a pair walks
        aux, n_per_pair, n_aux = aux_reader(os.path.join(aux_dir, fam + ".txt"), L)
        for a, b, la, lb in pairs:
            pair_rows.append((seq_off + row_of[a] * L, seq_off + row_of[b] * L, aux_off, n_per_pair, 0, la, lb))
        seq_chunks.append(codes.reshape(-1))
        aux_chunks.append(aux)
        seq_off += codes.size
        aux_off += n_aux
    seqs = np.concatenate(seq_chunks) if seq_chunks else np.zeros(0, dtype=np.int8)
    pairs = np.array(pair_rows, dtype=PAIR_DTYPE) if pair_rows else np.zeros(0, dtype=PAIR_DTYPE)
    return np.ascontiguousarray(seqs), aux_chunks, pairs


def _rank() -> int:
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist.get_rank()
    except ImportError:  # pragma: no cover
        pass
    return 0


def _run_local_then_agree(local, name):
    """Run the rank-local part; under torch.distributed exchange success before the all-reduce,
    so a rank whose files are broken makes EVERY rank raise instead of stranding its peers."""
    from ..caching._cached import _dist_state, _raise_together
    if _dist_state()[1] == 1:
        local()
        return
    error = None
    try:
        local()
    except Exception as exc:
        error = f"{type(exc).__name__}: {exc}"
    _raise_together(error, name)


def _write_cpp_layout(path, grid, C, states):
    """The count-matrix file as the reference's C++ counters write it (counting/_count_transitions.cpp:524-548: an ofstream
    at its default precision -- six significant digits, `%g` -- for the quantisation points AND the counts, a tab in front
    of and behind the header row).  Lossy above 999 999.5 counts per bin; the reference's reader takes it as it is."""
    with open(path, "w") as f:
        f.write(f"{len(grid)} matrices\n{len(states)} states\n")
        head = "\t" + "".join(s + "\t" for s in states) + "\n"
        for b, q in enumerate(grid):
            f.write("%g\n" % q)
            f.write(head)
            for i, st in enumerate(states):
                f.write(st + "\t" + "\t".join("%g" % v for v in C[b, i]) + "\n")


def _write(output_dir, grid, counts, unit, states, start, num_processes, cpp_compat: bool = False):
    """Rank 0 only (every rank holds the same all-reduced counts); files appear atomically."""
    if _rank() != 0:
        return
    C = counts.astype(np.float64) * unit
    tmp = os.path.join(output_dir, "result.txt.tmp")
    if cpp_compat:
        _write_cpp_layout(tmp, grid, C, states)
    else:
        write_count_matrices([(float(q), pd.DataFrame(C[b], index=states, columns=states))
                              for b, q in enumerate(grid)], tmp)
    os.replace(tmp, os.path.join(output_dir, "result.txt"))
    tmp = os.path.join(output_dir, "profiling.txt.tmp")
    with open(tmp, "w") as f:
        f.write(f"Total time: {time.time() - start} seconds with {num_processes} processes.\n")
    os.replace(tmp, os.path.join(output_dir, "profiling.txt"))


@caching.cached_computation(
    exclude_args=["num_processes", "use_cpp_implementation", "cpp_command_line_prefix",
                  "cpp_command_line_suffix"],
    output_dirs=["output_count_matrices_dir"], write_extra_log_files=True, collective=True)
def count_transitions(
    tree_dir: str,
    msa_dir: str,
    site_rates_dir: str,
    families: List[str],
    amino_acids: List[str],
    quantization_points: List[Union[str, float]],
    edge_or_cherry: str,
    output_count_matrices_dir: Optional[str] = None,
    num_processes: int = 1,
    use_cpp_implementation: bool = True,
    cpp_command_line_prefix: str = "",
    cpp_command_line_suffix: str = "",
    cpp_compat: bool = False,
) -> None:
    """`use_cpp_implementation` is accepted and ignored: there is one implementation (the GPU), and by default it follows
    the reference's PYTHON counter (float64 branch lengths, `repr` digits in the file).  `cpp_compat=True` reproduces what
    the reference's default, the C++ binary, does differently: branch lengths rounded to float32 when the tree is read
    (_count_transitions.cpp:247, `std::stof`) -- a length near a bucket boundary can land in the neighbouring bucket --
    and the result file in the C++ writer's layout with six significant digits (:524-548)."""
    start = time.time()
    logging.getLogger(__name__).info(f"Starting on {len(families)} families")
    mode = _normalise_mode(edge_or_cherry)
    if _rank() == 0:
        os.makedirs(output_count_matrices_dir, exist_ok=True)
    grid = np.array(sorted(float(q) for q in quantization_points), dtype=np.float64)
    S, B = len(amino_acids), len(grid)

    def rates_reader(path, L):
        r = _host.read_site_rates(path)
        if len(r) < L:  # the reference indexes site_rates[0 .. L-1]; extra entries are ignored
            raise Exception(f"{path}: {len(r)} site rates for an MSA of {L} sites")
        return r, L, len(r)

    counts = np.zeros((B, S, S), dtype=np.uint64)

    def local():   # this rank's families; no collective inside
        seqs, aux_chunks, pairs = _gather(tree_dir, msa_dir, _my_families(families), amino_acids, mode,
                                          site_rates_dir, rates_reader, cpp_compat)
        rates = np.ascontiguousarray(np.concatenate(aux_chunks)) if aux_chunks else np.zeros(0)
        rc = _lib.load().cb_count_transitions(
            _device_index(), S, B, grid.ctypes.data, seqs.ctypes.data, seqs.size, rates.ctypes.data,
            rates.size, pairs.ctypes.data, len(pairs), int(mode != "edge"), 0, counts.ctypes.data)
        _lib.check(rc, "cb_count_transitions")

    _run_local_then_agree(local, "count_transitions")
    counts = _all_reduce_counts(counts)
    _write(output_count_matrices_dir, grid, counts, 1.0 if mode == "edge" else 0.5,
           list(amino_acids), start, num_processes, cpp_compat)


@caching.cached_computation(
    exclude_args=["num_processes", "use_cpp_implementation", "cpp_command_line_prefix",
                  "cpp_command_line_suffix"],
    output_dirs=["output_count_matrices_dir"], write_extra_log_files=True, collective=True)
def count_co_transitions(
    tree_dir: str,
    msa_dir: str,
    contact_map_dir: str,
    families: List[str],
    amino_acids: List[str],
    quantization_points: List[Union[str, float]],
    edge_or_cherry: str,
    minimum_distance_for_nontrivial_contact: int,
    output_count_matrices_dir: Optional[str] = None,
    num_processes: int = 1,
    use_cpp_implementation: bool = True,
    cpp_command_line_prefix: str = "",
    cpp_command_line_suffix: str = "",
    cpp_compat: bool = False,
) -> None:
    """As `count_transitions`: `cpp_compat=True` = float32 branch lengths (_count_co_transitions.cpp:245) and the C++
    writer's six-digit file; the default follows the reference's Python counter."""
    start = time.time()
    logging.getLogger(__name__).info(f"Starting on {len(families)} families")
    mode = _normalise_mode(edge_or_cherry)
    if _rank() == 0:
        os.makedirs(output_count_matrices_dir, exist_ok=True)
    grid = np.array(sorted(float(q) for q in quantization_points), dtype=np.float64)
    S, B = len(amino_acids), len(grid)
    mdnc = int(minimum_distance_for_nontrivial_contact)

    def contacts_reader(path, L):
        cm = _host.read_contact_map(path)
        i, j = np.nonzero(cm == 1)
        keep = (j - i >= mdnc) & (i < j)  # row-major order, like np.where in the reference
        ij = np.stack([i[keep], j[keep]], axis=1).astype(np.int32)
        if ij.size and L and ij.max() >= L:
            raise Exception(f"{path}: contact map larger than the MSA ({L} sites)")
        return ij.reshape(-1), ij.shape[0], ij.shape[0]

    counts = np.zeros((B, S * S, S * S), dtype=np.uint64)

    def local():   # this rank's families; no collective inside
        seqs, aux_chunks, pairs = _gather(tree_dir, msa_dir, _my_families(families), amino_acids, mode,
                                          contact_map_dir, contacts_reader, cpp_compat)
        contacts = np.ascontiguousarray(np.concatenate(aux_chunks)) if aux_chunks else np.zeros(0, dtype=np.int32)
        contacts = contacts.astype(np.int32)
        rc = _lib.load().cb_count_co_transitions(
            _device_index(), S, B, grid.ctypes.data, seqs.ctypes.data, seqs.size, contacts.ctypes.data,
            contacts.size // 2, pairs.ctypes.data, len(pairs), int(mode != "edge"), 0, counts.ctypes.data)
        _lib.check(rc, "cb_count_co_transitions")

    _run_local_then_agree(local, "count_co_transitions")
    counts = _all_reduce_counts(counts)
    states = [a + b for a in amino_acids for b in amino_acids]
    _write(output_count_matrices_dir, grid, counts, 0.5 if mode == "edge" else 0.25, states, start,
           num_processes, cpp_compat)
