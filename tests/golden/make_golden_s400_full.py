#!/usr/bin/env python3
"""Reference goldens at the sizes the headline number is measured on: 400 states,
B = 129, and trajectories of >= 50 epochs -- produced by RUNNING THE REFERENCE (build container only).

  coevo_demo_full.npz   BASELINE.json config 3 as it really is: ALL 32 demo_data families through the
                        reference's `coevolution_end_to_end_with_cherryml_optimizer`
                        (estimation_end_to_end/_cherry.py:449-584: maximal matching, Python
                        `count_co_transitions`, masked JTT-IPW), then on that bank (sum C = 1 057 194,
                        43 of 129 buckets non-empty, 3.5 % dense; stored sparse)
                          * one float64 evaluation (loss, dL/dQ, dL/dtheta) of the epoch body
                            trainer.py:156-186 at the JTT-IPW initialisation,
                          * a float64 `train_quantization` trajectory of EPOCHS_DEMO epochs
                            (trainer.py:118-243; SURVEY 8c "f64 oracle recipe").
                        Only the non-empty buckets are handed to torch: an all-zero C_b multiplies
                        log P_b by 0 and adds exactly 0 to the loss and to every gradient.
  coevo_dense_eval.npz  the DENSE synthetic bank bench.py times (`bench.make_workload("coevo400")`,
                        B = 129, all buckets populated): one float64 evaluation at the JTT-IPW
                        initialisation.  The 165 MB bank is regenerated from its seed by the test; the
                        fixture holds the parameters, the outputs and checksums of the bank.
  coevo_dense_traj.npz  8 buckets of that bank (every 16th): a 60-epoch float64 trajectory and the
                        as-is float32 run of `quantized_transitions_mle` on the same counts.
  coevo_dense_traj_full.npz  the bench configuration itself: all 129 buckets, 60 epochs, float64.

Usage:  python tests/golden/make_golden_s400_full.py [demo] [dense_eval] [dense_traj] [dense_traj_full]   (~45 min for all)"""
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import (REF, _count_arrays, _prepare_scratch, _traj_reference_asis,  # noqa: E402
                         _traj_reference_f64)

EPOCHS_DEMO, EPOCHS_DENSE = 50, 60
THREADS = int(os.environ.get("GOLDEN_THREADS", "6"))


def _module_f64(torch, RateMatrix, S, mask, init):
    torch.manual_seed(0)
    return RateMatrix(num_states=S, mode="pande_reversible", pi=torch.ones(S, dtype=torch.float64) / S,
                      pi_requires_grad=True, initialization=init,
                      mask=torch.tensor(mask, dtype=torch.float)).double()


def _eval_f64_chunked(torch, RateMatrix, t, C, mask, init, chunk=12):
    """trainer.py:156-186 in float64 on the reference's RateMatrix; the buckets go through
    torch.matrix_exp + autograd in chunks (the [B,2S,2S] backward workspace of all 129 at once is ~10 GB):
    the loss and the gradients are sums over buckets, so only the order of those sums differs."""
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        module = _module_f64(torch, RateMatrix, C.shape[-1], mask, init)
        params = dict(upper_diag=module.upper_diag.detach().numpy().copy(), log_pi=module._pi.detach().numpy().copy())
        n = float(C.sum())
        Q = module()
        Q.retain_grad()
        total = 0.0
        for b0 in range(0, len(t), chunk):
            tt = torch.tensor(t[b0:b0 + chunk])
            CC = torch.tensor(C[b0:b0 + chunk])
            part = -(torch.log(torch.matrix_exp(tt[:, None, None] * Q)) * CC).sum() / n
            part.backward(retain_graph=True)
            total += float(part.item())
        out = dict(Q=Q.detach().numpy().copy(), loss=total, dQ=Q.grad.numpy().copy(),
                   d_upper=module.upper_diag.grad.numpy().copy(), d_log_pi=module._pi.grad.numpy().copy())
    finally:
        torch.set_default_dtype(old)
    return params, out


def _on_support(Q, mask):
    """masked rate matrices are stored on the mask's support + the diagonal"""
    keep = (mask != 0) | np.eye(len(mask), dtype=bool)
    return Q[keep]


def make_demo(torch, RateMatrix, train_quantization):
    from cherryml import caching
    from cherryml.estimation_end_to_end import coevolution_end_to_end_with_cherryml_optimizer
    from cherryml.io import read_count_matrices, read_mask_matrix, read_rate_matrix
    demo = os.path.join(REF, "demo_data")
    families = sorted(f[:-4] for f in os.listdir(os.path.join(demo, "msas")) if f.endswith(".txt"))
    mask_path = os.path.join("data", "mask_matrices", "aa_coevolution_mask.txt")
    cache = os.path.join(tempfile.gettempdir(), "_golden_demo_co_cache")
    caching.set_cache_dir(cache)
    t0 = time.time()
    r = coevolution_end_to_end_with_cherryml_optimizer(
        msa_dir=os.path.join(demo, "msas"), contact_map_dir=os.path.join(demo, "contact_maps"),
        minimum_distance_for_nontrivial_contact=7, coevolution_mask_path=mask_path, families=families,
        tree_estimator=None, initial_tree_estimator_rate_matrix_path=None, use_cpp_counting_implementation=False,
        optimizer_device="cpu", tree_dir=os.path.join(demo, "trees"), num_epochs=1,
        num_processes_tree_estimation=1, num_processes_counting=1, num_processes_optimization=THREADS)
    print(f"demo: reference pipeline (1 epoch) {time.time() - t0:.0f} s", flush=True)
    t, C = _count_arrays(read_count_matrices(os.path.join(r["count_matrices_dir_0"], "result.txt")))
    init = read_rate_matrix(os.path.join(r["jtt_ipw_dir_0"], "result.txt")).to_numpy()
    mask = read_mask_matrix(mask_path).to_numpy().astype(np.float64)
    live = np.flatnonzero(C.reshape(len(t), -1).any(axis=1))
    print(f"demo: {len(families)} families, sum C = {C.sum()}, non-empty buckets {live.size}, "
          f"non-zeros {np.count_nonzero(C)}", flush=True)
    t0 = time.time()
    params, ev = _eval_f64_chunked(torch, RateMatrix, t[live], C[live], mask, init)
    print(f"demo: f64 evaluation {time.time() - t0:.0f} s, loss {ev['loss']:.12f}", flush=True)
    t0 = time.time()
    _, tr = _traj_reference_f64(torch, RateMatrix, train_quantization, t[live], C[live], mask, init, EPOCHS_DEMO)
    print(f"demo: f64 trajectory of {EPOCHS_DEMO} epochs {time.time() - t0:.0f} s, loss {tr['loss'][0]:.10f} -> "
          f"{tr['loss'][-1]:.10f}", flush=True)
    nz = np.nonzero(C)
    quarter = np.rint(C[nz] * 4.0)
    assert np.array_equal(quarter * 0.25, C[nz]) and quarter.max() < 2 ** 32      # counts are multiples of 1/4
    np.savez_compressed(
        os.path.join(HERE, "coevo_demo_full.npz"), families=np.array(families), t=t,
        C_shape=np.array(C.shape), C_b=nz[0].astype(np.uint8), C_i=nz[1].astype(np.uint16), C_j=nz[2].astype(np.uint16),
        C_quarters=quarter.astype(np.uint32), mask_packed=np.packbits(mask.astype(bool)),
        init_support=_on_support(init, mask), upper_diag=params["upper_diag"], log_pi=params["log_pi"],
        loss_f64=np.float64(ev["loss"]), dQ_f64=ev["dQ"], d_upper_f64=ev["d_upper"], d_log_pi_f64=ev["d_log_pi"],
        Q_support_f64=_on_support(ev["Q"], mask), epochs=np.int64(EPOCHS_DEMO), traj_loss_f64=tr["loss"],
        traj_Q_best_support_f64=_on_support(tr["Q_best"], mask), traj_Q_last_support_f64=_on_support(tr["Q_last"], mask),
        traj_Q_1_support_f64=_on_support(tr["Q_1"], mask), traj_Q_2_support_f64=_on_support(tr["Q_2"], mask))
    print("wrote coevo_demo_full.npz", flush=True)


def _dense_bank():
    import bench
    wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
    from cherryml_amd.estimation import jtt_ipw_from_arrays
    # the evaluation point: the pipelines' default initialiser (this repository's host implementation of
    # jtt_ipw, itself pinned on the reference's goldens; any point would do for a single evaluation)
    init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
    return wl, init


def _bank_checksums(C):
    return dict(C_sum=np.float64(C.sum()), C_bucket_sums=C.reshape(C.shape[0], -1).sum(1),
                C_probe=C[::16, ::37, ::41].copy())


def make_dense_eval(torch, RateMatrix, train_quantization):
    wl, init = _dense_bank()
    t0 = time.time()
    params, ev = _eval_f64_chunked(torch, RateMatrix, wl["t"], wl["C"], wl["mask"], init)
    print(f"dense: f64 evaluation of 129 buckets {time.time() - t0:.0f} s, loss {ev['loss']:.12f}", flush=True)
    np.savez_compressed(
        os.path.join(HERE, "coevo_dense_eval.npz"), upper_diag=params["upper_diag"], log_pi=params["log_pi"],
        init_support=_on_support(init, wl["mask"]), loss_f64=np.float64(ev["loss"]), dQ_f64=ev["dQ"],
        d_upper_f64=ev["d_upper"], d_log_pi_f64=ev["d_log_pi"], Q_support_f64=_on_support(ev["Q"], wl["mask"]),
        **_bank_checksums(wl["C"]))
    print("wrote coevo_dense_eval.npz", flush=True)


def make_dense_traj(torch, RateMatrix, train_quantization):
    import pandas as pd
    from cherryml.estimation import quantized_transitions_mle
    from cherryml.io import read_rate_matrix, write_count_matrices, write_rate_matrix
    wl, init = _dense_bank()
    sel = np.arange(4, 129, 16)                       # 8 buckets, t from 1e-4 to 1.3
    t, C, mask = wl["t"][sel], wl["C"][sel], wl["mask"]
    t0 = time.time()
    params, tr = _traj_reference_f64(torch, RateMatrix, train_quantization, t, C, mask, init, EPOCHS_DENSE)
    print(f"dense: f64 trajectory of {EPOCHS_DENSE} epochs on {len(sel)} buckets {time.time() - t0:.0f} s, loss "
          f"{tr['loss'][0]:.10f} -> {tr['loss'][-1]:.10f}", flush=True)
    states = [a + b for a in "ARNDCQEGHILKMFPSTWYV" for b in "ARNDCQEGHILKMFPSTWYV"]
    with tempfile.TemporaryDirectory() as d:
        cpath, ipath, mpath = (os.path.join(d, f) for f in ("counts.txt", "init.txt", "mask.txt"))
        write_count_matrices([(float(t[b]), pd.DataFrame(C[b], index=states, columns=states)) for b in range(len(t))], cpath)
        write_rate_matrix(init, states, ipath)
        pd.DataFrame(mask.astype(int), index=states, columns=states).to_csv(mpath, sep=" ")
        t0 = time.time()
        asis, _ = _traj_reference_asis(quantized_transitions_mle, read_rate_matrix, cpath, ipath, mpath, EPOCHS_DENSE)
        print(f"dense: as-is float32 run {time.time() - t0:.0f} s, loss {asis['loss'][0]:.7f} -> {asis['loss'][-1]:.7f}",
              flush=True)
    kw = dict(sel=sel, epochs=np.int64(EPOCHS_DENSE), upper_diag0=params["upper_diag"], log_pi0=params["log_pi"],
              init_support=_on_support(init, mask), loss_f64=tr["loss"], loss_f32=asis["loss"], **_bank_checksums(wl["C"]))
    for k in ("Q_best", "Q_last", "Q_1", "Q_2"):
        kw[k + "_support_f64"] = _on_support(tr[k], mask)
        kw[k + "_support_f32"] = _on_support(asis[k], mask)
    np.savez_compressed(os.path.join(HERE, "coevo_dense_traj.npz"), **kw)
    print("wrote coevo_dense_traj.npz", flush=True)


def make_dense_traj_full(torch, RateMatrix, train_quantization):
    """The bench configuration itself: all 129 buckets of the dense bank, EPOCHS_DENSE epochs, float64."""
    wl, init = _dense_bank()
    t0 = time.time()
    params, tr = _traj_reference_f64(torch, RateMatrix, train_quantization, wl["t"], wl["C"], wl["mask"], init,
                                     EPOCHS_DENSE)
    print(f"dense: f64 trajectory of {EPOCHS_DENSE} epochs on all 129 buckets {time.time() - t0:.0f} s, loss "
          f"{tr['loss'][0]:.10f} -> {tr['loss'][-1]:.10f}", flush=True)
    kw = dict(epochs=np.int64(EPOCHS_DENSE), upper_diag0=params["upper_diag"], log_pi0=params["log_pi"],
              init_support=_on_support(init, wl["mask"]), loss_f64=tr["loss"], **_bank_checksums(wl["C"]))
    for k in ("Q_best", "Q_last", "Q_1", "Q_2"):
        kw[k + "_support_f64"] = _on_support(tr[k], wl["mask"])
    np.savez_compressed(os.path.join(HERE, "coevo_dense_traj_full.npz"), **kw)
    print("wrote coevo_dense_traj_full.npz", flush=True)


def main():
    which = sys.argv[1:] or ["demo", "dense_eval", "dense_traj", "dense_traj_full"]
    _prepare_scratch()
    import torch

    import cherryml  # noqa: F401
    from cherryml.estimation._ratelearn.rate import RateMatrix
    from cherryml.estimation._ratelearn.trainer import train_quantization
    torch.set_num_threads(THREADS)
    for name in which:
        {"demo": make_demo, "dense_eval": make_dense_eval, "dense_traj": make_dense_traj,
         "dense_traj_full": make_dense_traj_full}[name](
            torch, RateMatrix, train_quantization)


if __name__ == "__main__":
    main()
