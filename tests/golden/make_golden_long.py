#!/usr/bin/env python3
"""Reference goldens at the reference's own HORIZON ("missing 1"): the reference optimises for 500 epochs
(co-evolution, estimation_end_to_end/_cherry.py:463, _cherryml_public_api.py:56) or 2000 (_quantized_transitions_mle.py:49),
the longest pinned trajectory so far was 100 / 60.  Produced by RUNNING THE REFERENCE (build container only):

  long_s64.npz      64 states, B = 129, noise-free symmetric bank of a random reversible model, 2000 epochs of
                    `train_quantization` in float64 (trainer.py:118-243) from a start at 0.3 x the generating rates:
                    max |Q_ii| grows 3x, so the time basis this build runs the bank in (csrc/tbasis.hip.h) leaves
                    the range it was built for and is REPLACED on the way -- without any test hook.
  long_s400_b32.npz 32 buckets of the bench bank (`bench.make_workload("coevo400")`, every 4th bucket from 2), 400 states,
                    500 epochs in float64: the co-evolution horizon on a bank that runs in the time basis.

Both banks are regenerated from their seeds by the tests; the fixtures hold parameters, outputs and checksums.
Usage:  python tests/golden/make_golden_long.py [s64] [s400]      (s64: ~10 min, s400: hours)"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

EPOCHS_S64, EPOCHS_S400 = 2000, 500
S400_SEL = np.arange(2, 129, 4)
THREADS = int(os.environ.get("GOLDEN_THREADS", "4"))


def s64_bank():
    """(t, C, Q_true, pi, init): shared with tests/test_gpu_long_horizon.py, which regenerates the bank from here"""
    import bench
    rng = np.random.default_rng(64)
    S = 64
    pi = rng.dirichlet(np.full(S, 4.0))
    ex = rng.gamma(0.6, 1.0, size=(S, S))
    ex = np.triu(ex, 1)
    ex = ex + ex.T
    Q = ex * pi[None, :]
    Q -= np.diag(Q.sum(1))
    Q /= -(pi * np.diag(Q)).sum()
    t, C = bench.reversible_bank(Q, pi, 2.0e6, rng)
    return t, C, Q, pi, 0.3 * Q


def _checks(C):
    return dict(C_sum=np.float64(C.sum()), C_bucket_sums=C.reshape(C.shape[0], -1).sum(1), C_probe=C[::16, ::7, ::5].copy())


def _ref():
    from make_golden import _prepare_scratch
    _prepare_scratch()   # the reference (a scratch copy with its extension built) on sys.path
    import torch
    import cherryml  # noqa: F401
    torch.set_num_threads(THREADS)
    from cherryml.estimation._ratelearn.rate import RateMatrix
    from cherryml.estimation._ratelearn.trainer import train_quantization
    return torch, RateMatrix, train_quantization


def _traj(torch, RateMatrix, train_quantization, t, C, mask, init, num_epochs, lr, upper_diag=None, log_pi=None):
    """SURVEY 8c's float64 recipe (make_golden._traj_reference_f64) with the learning rate as a parameter
    (`quantized_transitions_mle(learning_rate=...)`, _quantized_transitions_mle.py:48) and every power-of-two snapshot kept"""
    from torch.utils.data import TensorDataset
    S = C.shape[-1]
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(0)
        module = RateMatrix(num_states=S, mode="pande_reversible", pi=torch.ones(S, dtype=torch.float64) / S,
                            pi_requires_grad=True, initialization=init, mask=torch.tensor(mask, dtype=torch.float)).double()
        if upper_diag is not None:
            module.upper_diag.data.copy_(torch.tensor(upper_diag))
            module._pi.data.copy_(torch.tensor(log_pi))
        params = dict(upper_diag=module.upper_diag.detach().numpy().copy(), log_pi=module._pi.detach().numpy().copy())
        ds = TensorDataset(torch.tensor(t), torch.tensor(C))
        opt = torch.optim.Adam(module.parameters(), lr=lr)
        df, Qd = train_quantization(rate_module=module, quantized_dataset=ds, num_epochs=num_epochs, Q_true=None, optimizer=opt,
                                    loss_normalization=True, return_best_iter=True)
    finally:
        torch.set_default_dtype(old)
    out = dict(loss=df.loss.to_numpy().astype(np.float64))
    for k, v in Qd.items():
        out[k] = np.asarray(v, dtype=np.float64)
    return params, out


def make_s64():
    """Four reference runs: lr = 0.1 (the reference's default, _quantized_transitions_mle.py:48) and lr = 0.02, each with a TWIN
    whose starting parameters differ by 1e-14 relative.  At lr = 0.1 the noise-free optimum is reached after ~300 epochs and
    Adam's constant step then bounces around it (the loss moves by 1e-6 from epoch to epoch): the twins say how far two
    float64 evaluations of the SAME recipe drift apart there -- the yardstick the GPU test uses beside the absolute bars."""
    torch, RateMatrix, train_quantization = _ref()
    t, C, Q, pi, init = s64_bank()
    mask = np.ones((64, 64))
    kw = dict(epochs=np.int64(EPOCHS_S64), Q_true=Q, **_checks(C))
    rng = np.random.default_rng(1)
    for tag, lr in (("", 0.1), ("_lr002", 0.02)):
        t0 = time.time()
        params, tr = _traj(torch, RateMatrix, train_quantization, t, C, mask, init, EPOCHS_S64, lr)
        print(f"s64 lr {lr}: {EPOCHS_S64} epochs {time.time() - t0:.0f} s, loss {tr['loss'][0]:.10f} -> {tr['loss'][-1]:.10f}; "
              f"max|Q_ii| {np.abs(np.diag(init)).max():.3f} -> {np.abs(np.diag(tr['Q_last'])).max():.3f}", flush=True)
        if tag == "":
            kw.update(upper_diag0=params["upper_diag"], log_pi0=params["log_pi"])
            u1 = params["upper_diag"] * (1.0 + 1e-14 * rng.standard_normal(params["upper_diag"].shape))
        _, tw = _traj(torch, RateMatrix, train_quantization, t, C, mask, init, EPOCHS_S64, lr, upper_diag=u1, log_pi=params["log_pi"])
        dl = np.abs(tw["loss"] - tr["loss"]) / np.abs(tr["loss"])
        print(f"   twin (start moved by 1e-14): loss curves differ by {dl[:200].max():.1e} (epochs < 200), {dl[:500].max():.1e} (< 500), "
              f"{dl.max():.1e} (all); Q_best {np.linalg.norm(tw['Q_best'] - tr['Q_best']) / np.linalg.norm(tr['Q_best']):.1e}, "
              f"Q_last {np.linalg.norm(tw['Q_last'] - tr['Q_last']) / np.linalg.norm(tr['Q_last']):.1e}", flush=True)
        kw["lr" + tag] = np.float64(lr)
        kw["upper_diag0_twin"] = u1
        kw["loss_f64" + tag] = tr["loss"]
        kw["loss_twin_f64" + tag] = tw["loss"]
        for k in ("Q_best", "Q_last", "Q_1", "Q_2", "Q_256", "Q_1024"):
            kw[k + "_f64" + tag] = tr[k]
        for k in ("Q_best", "Q_last", "Q_256", "Q_1024"):
            kw[k + "_twin_f64" + tag] = tw[k]
    np.savez_compressed(os.path.join(HERE, "long_s64.npz"), **kw)
    print("wrote long_s64.npz", flush=True)


def make_s400():
    from make_golden import _traj_reference_f64
    from make_golden_s400_full import _on_support
    import bench
    torch, RateMatrix, train_quantization = _ref()
    wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
    # the starting point of coevo_dense_traj_full.npz (the masked JTT-IPW initialiser of the whole bank; this repository's
    # jtt_ipw runs on the MI355X only, so the matrix is taken from the fixture the GPU run left)
    z = np.load(os.path.join(HERE, "coevo_dense_traj_full.npz"))
    keep = (wl["mask"] != 0) | np.eye(400, dtype=bool)
    init = np.zeros((400, 400))
    init[keep] = z["init_support"]
    t, C, mask = wl["t"][S400_SEL], wl["C"][S400_SEL], wl["mask"]
    t0 = time.time()
    params, tr = _traj_reference_f64(torch, RateMatrix, train_quantization, t, C, mask, init, EPOCHS_S400)
    print(f"s400: {EPOCHS_S400} epochs on {len(S400_SEL)} buckets {time.time() - t0:.0f} s, loss {tr['loss'][0]:.10f} -> "
          f"{tr['loss'][-1]:.10f}", flush=True)
    # the TWIN: the same recipe from a start moved by 1e-14 relative -- how far two float64 evaluations of the reference drift
    # apart over this horizon (the yardstick for Q_last, which sits at the end of 500 Adam steps)
    fin = np.isfinite(params["upper_diag"])
    u1 = params["upper_diag"].copy()
    u1[fin] *= 1.0 + 1e-14 * np.random.default_rng(2).standard_normal(int(fin.sum()))
    _, tw = _traj_reference_f64(torch, RateMatrix, train_quantization, t, C, mask, init, EPOCHS_S400, upper_diag=u1,
                                log_pi=params["log_pi"])
    dl = np.abs(tw["loss"] - tr["loss"]) / np.abs(tr["loss"])
    rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)  # noqa: E731
    print(f"   twin (start moved by 1e-14): loss curves differ by {dl.max():.1e}; Q_best {rel(tw['Q_best'], tr['Q_best']):.1e}, "
          f"Q_last {rel(tw['Q_last'], tr['Q_last']):.1e}", flush=True)
    kw = dict(sel=S400_SEL, epochs=np.int64(EPOCHS_S400), upper_diag0=params["upper_diag"], log_pi0=params["log_pi"],
              init_support=_on_support(init, mask), loss_f64=tr["loss"], C_sum=np.float64(wl["C"].sum()),
              C_bucket_sums=wl["C"].reshape(129, -1).sum(1), C_probe=wl["C"][::16, ::37, ::41].copy())
    for k in ("Q_best", "Q_last", "Q_1", "Q_2"):
        kw[k + "_support_f64"] = _on_support(tr[k], mask)
    kw["loss_twin_f64"] = tw["loss"]
    for k in ("Q_best", "Q_last"):
        kw[k + "_twin_support_f64"] = _on_support(tw[k], mask)
    np.savez_compressed(os.path.join(HERE, "long_s400_b32.npz"), **kw)
    print("wrote long_s400_b32.npz", flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["s64", "s400"]
    if "s64" in which:
        make_s64()
    if "s400" in which:
        make_s400()
