"""The co-evolution estimation as ONE resident chain: count -> JTT-IPW -> optimise, nothing but file inputs and the
learned matrix crossing the host boundary.

The reference (and the mirrored pipeline `coevolution_end_to_end_with_cherryml_optimizer`) passes the 400-state count
tensor between its stages as an 84 MB text file (estimation_end_to_end/_cherry.py:514-574: `count_co_transitions` writes
`result.txt`, `jtt_ipw` and `quantized_transitions_mle` parse it again).  Here

  1. every rank counts ITS families (the reference's MPI dealing, counting/_count_co_transitions.cpp:626-628) into a
     device-resident integer histogram (`cb_count_co_transitions`, resident form);
  2. the JTT-IPW initialiser (estimation/_jtt_ipw.py:32-125) is linear in the counts up to its pseudocount, so it needs
     only two S x S sums over the buckets -- sum_b sym(C_b) and sum_b sym(C_b) / t_b -- which are all-reduced (2 x 1.28 MB
     instead of the 165 MB tensor);
  3. the counts go straight into `ShardedBank.from_rank_counts` (ONE reduce-scatter over the non-empty buckets: no rank
     ever holds the summed tensor) and the whole epoch loop runs from C on every rank (`train_pande_reversible`).

One process: the same chain without collectives.  The result equals the file-passing pipeline's (tests/test_gpu_demo_e2e.py).
"""
from typing import Dict, List, Optional, Union

import numpy as np

from .. import _lib
from ..counting import _host
from ..counting._stage import _device_index, _gather, _my_families, _normalise_mode, _run_local_then_agree


def jtt_ipw_from_reduced_statistics(F_sym: np.ndarray, R_sym: np.ndarray, grid: np.ndarray, mask: Optional[np.ndarray],
                                    pseudocounts: float = 1e-8) -> np.ndarray:
    """`jtt_ipw_from_arrays(use_ipw=True, symmetrize=True)` from F_sym = sum_b sym(C_b) and R_sym = sum_b sym(C_b) / t_b
    (what the ranks all-reduce instead of the tensor): estimation/_jtt_ipw.py::jtt_ipw_from_statistics."""
    from ..estimation import jtt_ipw_from_statistics
    return jtt_ipw_from_statistics(F_sym, R_sym, grid, mask, True, pseudocounts)


def _train_with_torch_glue(sharded, module, num_epochs: int, lr: float, do_adam: bool) -> Dict:
    """The reference's loop (trainer.py:156-218: loss, best iterate = Q BEFORE the step with strict <, snapshots at epochs
    1, 2, 4, ...) with the sharded bank's differentiable loss; every rank takes the same steps (the all-reduced loss and
    gradient are identical everywhere)."""
    import torch
    opt = (torch.optim.Adam if do_adam else torch.optim.SGD)(module.parameters(), lr=lr)
    losses, best, Q_best, Q, snaps = [], None, None, None, {}
    for epoch in range(num_epochs):
        opt.zero_grad()
        Q = module()
        loss = sharded.loss(Q, module.stationary(), normalize=True)[0]
        value = float(loss.item())
        if best is None or value < best:
            best, Q_best = value, Q.detach().cpu().numpy().copy()
        if (epoch & (epoch + 1)) == 0:
            snaps[epoch + 1] = Q.detach().cpu().numpy().copy()
        loss.backward()
        opt.step()
        losses.append(value)
    return dict(loss=np.array(losses), Q_best=Q_best, Q_last=None if Q is None else Q.detach().cpu().numpy().copy(),
                Q_pow2=snaps, upper_diag=module.upper_diag.detach().cpu().numpy().copy(),
                log_pi=module._pi.detach().cpu().numpy().copy())


def coevolution_fit_resident(
    tree_dir: str, msa_dir: str, contact_map_dir: str, families: List[str], amino_acids: List[str],
    quantization_points: List[Union[str, float]], edge_or_cherry: str, minimum_distance_for_nontrivial_contact: int,
    mask: Optional[np.ndarray] = None, num_epochs: int = 500, learning_rate: float = 0.1, do_adam: bool = True,
    bank_dtype: str = "f64",
) -> Dict:
    """Collective under torch.distributed (every rank calls it with the same arguments).  Returns
    dict(loss, Q_best, Q_last, Q_pow2, initialization, n_pairs, quantization_points) -- the same on every rank."""
    import torch
    import torch.distributed as dist
    from ..bank import CherryBank
    from ..distributed import ShardedBank
    from ..estimation._ratelearn._rate_matrix import RateMatrix

    if _lib.load().cb_device_count() <= 0:   # (a missing library raises CherryBankError inside load())
        raise _lib.CherryBankError("coevolution_fit_resident: no HIP device visible; cherryml_amd computes this path on the "
                                   "MI355X only and has no CPU fallback")
    mode = _normalise_mode(edge_or_cherry)
    grid = np.array(sorted(float(q) for q in quantization_points), dtype=np.float64)
    S1, B = len(amino_acids), len(grid)
    S = S1 * S1
    mdnc = int(minimum_distance_for_nontrivial_contact)
    on = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    device = _device_index()
    dev = torch.device("cuda", device)

    def contacts_reader(path, L):
        cm = _host.read_contact_map(path)
        i, j = np.nonzero(cm == 1)
        keep = (j - i >= mdnc) & (i < j)
        ij = np.stack([i[keep], j[keep]], axis=1).astype(np.int32)
        if ij.size and L and ij.max() >= L:
            raise Exception(f"{path}: contact map larger than the MSA ({L} sites)")
        return ij.reshape(-1), ij.shape[0], ij.shape[0]

    box = {}

    def local():   # this rank's families: parse on the host, histogram on the device, counts stay there
        seqs, aux_chunks, pairs = _gather(tree_dir, msa_dir, _my_families(families), amino_acids, mode, contact_map_dir,
                                          contacts_reader)
        contacts = (np.concatenate(aux_chunks) if aux_chunks else np.zeros(0, dtype=np.int32)).astype(np.int32)
        contacts = np.ascontiguousarray(np.concatenate([contacts, np.zeros(2, np.int32)]))
        d_counts = torch.zeros(B * S * S, dtype=torch.int64, device=dev)
        if len(pairs):
            d_seqs = torch.from_numpy(seqs if seqs.size else np.zeros(1, np.int8)).to(dev)
            d_contacts = torch.from_numpy(contacts).to(dev)
            d_pairs = torch.from_numpy(pairs.view(np.uint8)).to(dev)
            d_grid = torch.from_numpy(grid).to(dev)
            rc = _lib.load().cb_count_co_transitions(
                device, S1, B, d_grid.data_ptr(), d_seqs.data_ptr(), int(seqs.size), d_contacts.data_ptr(),
                contacts.size // 2, d_pairs.data_ptr(), len(pairs), int(mode != "edge"),
                _lib.CB_PTR_DEVICE, d_counts.data_ptr())
            _lib.check(rc, "cb_count_co_transitions")
            torch.cuda.synchronize(dev)
        box["counts"] = d_counts

    _run_local_then_agree(local, "coevolution_fit_resident")
    unit = 0.5 if mode == "edge" else 0.25
    counts_u64 = box.pop("counts")                                       # this rank's histogram (units of `unit`), resident
    # JTT-IPW from two S x S sums (all-reduced), not from the tensor: one streaming pass of cb_jtt_ipw_stats over the integers
    from ..estimation import jtt_ipw_statistics
    F_r, R_r = jtt_ipw_statistics(grid, counts_u64.reshape(B, S, S), unit, True)
    stats = torch.from_numpy(np.stack([F_r, R_r])).to(dev)
    C = (counts_u64.to(torch.float64) * unit).reshape(B, S, S)           # the sufficient statistics the bank is built from
    del counts_u64
    n_pairs = C.sum().reshape(1)
    if on:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
        dist.all_reduce(n_pairs, op=dist.ReduceOp.SUM)
    stats = stats.cpu().numpy()
    init = jtt_ipw_from_reduced_statistics(stats[0], stats[1], grid, mask)
    mk = np.ones((S, S)) if mask is None else np.asarray(mask, dtype=np.float64)
    mod = RateMatrix(num_states=S, mode="pande_reversible", mask=torch.tensor(mk), pi=torch.ones(S, dtype=torch.float64) / S,
                     pi_requires_grad=True, initialization=init)
    u0 = mod.upper_diag.detach().numpy().copy()
    p0 = mod._pi.detach().numpy().copy()
    if on:
        sharded = ShardedBank.from_rank_counts(grid, C, dtype=bank_dtype)
        del C
        try:
            try:   # raises on EVERY rank when any rank cannot make its raw RCCL communicator
                sharded.enable_in_library_allreduce()
                in_library = True
            except RuntimeError as exc:
                import warnings
                warnings.warn(f"coevolution_fit_resident: in-library all-reduce unavailable ({exc}); theta -> Q and the "
                              "optimiser step stay in torch, the collective is torch.distributed's")
                in_library = False
            if in_library:
                r = sharded.train_pande_reversible(u0, p0, mask=mask, num_epochs=num_epochs, lr=learning_rate, do_adam=do_adam)
            else:
                r = _train_with_torch_glue(sharded, mod.to(dev), num_epochs, learning_rate, do_adam)
        finally:
            sharded.close()
    else:
        with CherryBank(grid, C, device=device, dtype=bank_dtype) as bank:
            del C
            r = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=num_epochs, lr=learning_rate, do_adam=do_adam)
    r["initialization"] = init
    r["n_pairs"] = float(n_pairs.item())
    r["quantization_points"] = grid
    return r
