#!/usr/bin/env python3
"""Golden vectors for the four NON-default parameterisations of the reference's RateMatrix
(cherryml/estimation/_ratelearn/rate.py:98-128, 190-218: "default", "pande", "stationary",
"stationary_reversible"), produced by RUNNING THE REFERENCE in float64 (the recipe of make_golden.py):

  modes.npz   per mode m and case c (a 3-state toy bank, and the reference's 20-state test bank with its own
              non-symmetric random mask):
      <m>_<c>_upper / _lower / _log_pi       the parameters the reference's module was given
      <m>_<c>_Q, _loss, _dQ, _d_upper, _d_lower, _d_log_pi
                                             one evaluation of the epoch body (trainer.py:156-186)
      <m>_<c>_traj_loss, _Q_best, _Q_last    30 epochs of the reference's train_quantization (Adam, lr 0.05)

Only works in the build container (needs /root/reference).  Usage: python tests/golden/make_golden_modes.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import _count_arrays, _prepare_scratch  # noqa: E402


def main():
    _prepare_scratch()
    import torch
    from torch.utils.data import TensorDataset

    import cherryml  # noqa: F401
    from cherryml.estimation._ratelearn.rate import RateMatrix
    from cherryml.estimation._ratelearn.trainer import train_quantization
    from cherryml.io import read_count_matrices, read_mask_matrix

    torch.set_num_threads(4)
    torch.set_default_dtype(torch.float64)
    tid = "tests/test_input_data"
    cases = {
        "toy3": (f"{tid}/matrices_toy.txt", None),
        "s20": (f"{tid}/matrices_small/matrices_by_quantized_branch_length.txt", f"{tid}/20x20_random_mask.txt"),
    }
    out = {}
    rng = np.random.default_rng(20261003)
    for cname, (cpath, mpath) in cases.items():
        t, C = _count_arrays(read_count_matrices(cpath))
        S = C.shape[-1]
        mask = read_mask_matrix(mpath).to_numpy().astype(np.float64) if mpath else np.ones((S, S))
        out[f"{cname}_t"], out[f"{cname}_C"], out[f"{cname}_mask"] = t, C, mask
        half = S * (S - 1) // 2
        for mode in ("default", "pande", "stationary", "stationary_reversible"):
            upper = rng.normal(0.0, 0.5, half)
            lower = rng.normal(0.0, 0.5, half)
            log_pi = rng.normal(0.0, 0.3, S)

            def make():
                m = RateMatrix(num_states=S, mode=mode, pi=torch.ones(S) / S, pi_requires_grad=True,
                               initialization=None, mask=torch.tensor(mask, dtype=torch.float)).double()
                m.upper_diag.data.copy_(torch.tensor(upper))
                if hasattr(m, "lower_diag"):
                    m.lower_diag.data.copy_(torch.tensor(lower))
                m._pi.data.copy_(torch.tensor(log_pi))
                return m

            key = f"{mode}_{cname}"
            module = make()
            has_lower = hasattr(module, "lower_diag")
            Q = module()
            Q.retain_grad()
            CC = torch.tensor(C)
            loss = -(torch.log(torch.matrix_exp(torch.tensor(t)[:, None, None] * Q)) * CC).sum() / CC.sum()
            loss.backward()
            out[f"{key}_upper"], out[f"{key}_log_pi"] = upper, log_pi
            if has_lower:
                out[f"{key}_lower"] = lower
                out[f"{key}_d_lower"] = module.lower_diag.grad.numpy().copy()
            out[f"{key}_Q"] = Q.detach().numpy().copy()
            out[f"{key}_loss"] = np.float64(loss.item())
            out[f"{key}_dQ"] = Q.grad.numpy().copy()
            out[f"{key}_d_upper"] = module.upper_diag.grad.numpy().copy()
            # ("default" and "pande" do not use _pi: autograd leaves its gradient at None)
            out[f"{key}_d_log_pi"] = (module._pi.grad.numpy().copy() if module._pi.grad is not None else np.zeros(S))
            module = make()
            opt = torch.optim.Adam(module.parameters(), lr=0.05)
            df, Qd = train_quantization(rate_module=module, quantized_dataset=TensorDataset(torch.tensor(t), torch.tensor(C)),
                                        num_epochs=30, Q_true=None, optimizer=opt, loss_normalization=True,
                                        return_best_iter=True)
            out[f"{key}_traj_loss"] = df.loss.to_numpy().astype(np.float64)
            out[f"{key}_Q_best"] = np.asarray(Qd["Q_best"], dtype=np.float64)
            out[f"{key}_Q_last"] = np.asarray(Qd["Q_last"], dtype=np.float64)
            print(f"{key}: loss {loss.item():.12f}; 30 epochs {out[key + '_traj_loss'][0]:.6f} -> {out[key + '_traj_loss'][-1]:.6f}")
    np.savez_compressed(os.path.join(HERE, "modes.npz"), **out)


if __name__ == "__main__":
    main()
