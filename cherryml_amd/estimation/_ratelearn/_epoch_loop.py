"""The optimiser's epoch loop (reference: cherryml/estimation/_ratelearn/
trainer.py:118-243, `train_quantization`) with the expm bank, log-likelihood
contraction and gradient evaluated by libcherrybank on the MI355X.

What changes with respect to the reference, and nothing else:
  * (t, C) are uploaded ONCE into a CherryBank (the reference re-sends them
    every epoch, trainer.py:164-167);
  * trainer.py:170-177 + the matrix_exp part of `loss.backward()` become one
    call `bank_loss(Q, pi, bank)` (HIP); the parameterisation and the optimiser
    step stay torch;
  * arithmetic is float64.
"""
import logging
import time
from typing import Dict, Optional, Tuple

import numpy as np
import pandas as pd
import torch

from ..._autograd import bank_loss
from ...bank import CherryBank


def _snapshot(Q: torch.Tensor) -> np.ndarray:
    return Q.detach().cpu().numpy().copy()


class LazyOptimizer:
    """The reference's optimiser choice (`ratelearner.py:123-130`: Adam or plain SGD, default
    hyper-parameters) as a description.  The device-side loop only needs `kind` and `lr`; a real
    torch optimiser is built by `materialize()` when the torch loop runs (constructing torch.optim.Adam
    imports torch._dynamo: 0.6 s of a 2.8 s co-evolution stage)."""

    def __init__(self, params, lr: float, do_adam: bool):
        self.params, self.lr, self.do_adam = list(params), float(lr), bool(do_adam)

    def materialize(self):
        cls = torch.optim.Adam if self.do_adam else torch.optim.SGD
        return cls(params=self.params, lr=self.lr)


def _fusable(rate_module, optimizer, Q_true, m) -> bool:
    """The whole loop can run on the device without torch (cb_train_pande_reversible: one kernel
    for S <= 32, a C-driven kernel sequence for larger S) when nothing but the reference's
    standard configuration is asked for."""
    if rate_module.mode != "pande_reversible":
        return False
    if Q_true is not None or m != 1.0 or not rate_module._pi.requires_grad:
        return False
    if isinstance(optimizer, LazyOptimizer):
        return True
    if len(optimizer.param_groups) != 1 or optimizer.state:
        return False
    g = optimizer.param_groups[0]
    if isinstance(optimizer, torch.optim.Adam):
        return (tuple(g["betas"]) == (0.9, 0.999) and g["eps"] == 1e-8 and g["weight_decay"] == 0
                and not g["amsgrad"] and not g.get("maximize", False))
    if isinstance(optimizer, torch.optim.SGD):
        return (g["momentum"] == 0 and g["weight_decay"] == 0 and not g["nesterov"]
                and not g.get("maximize", False))
    return False


def _train_fused(rate_module, bank, optimizer, num_epochs, loss_normalization, return_best_iter):
    if isinstance(optimizer, LazyOptimizer):
        lr, do_adam = optimizer.lr, optimizer.do_adam
    else:
        lr, do_adam = optimizer.param_groups[0]["lr"], isinstance(optimizer, torch.optim.Adam)
    start = time.time()
    up0, pi0, mask0 = (rate_module.upper_diag.detach().cpu().numpy(), rate_module._pi.detach().cpu().numpy(),
                       rate_module.mask.detach().cpu().numpy())
    t_call = time.time()
    r = bank.train_pande_reversible(up0, pi0, mask=mask0, num_epochs=num_epochs, lr=lr, do_adam=do_adam,
                                    normalize=loss_normalization)
    with torch.no_grad():  # leave the module at the final parameters, like the torch loop does
        rate_module.upper_diag.copy_(torch.as_tensor(r["upper_diag"]))
        rate_module._pi.copy_(torch.as_tensor(r["log_pi"]))
    E = num_epochs
    # `time` (trainer.py:207-217: seconds since the start, at the end of every epoch): the loop never returns to the
    # host between epochs, so the DEVICE stamps its 100 MHz wall clock at the end of every epoch's parameter step
    # (cb_train_epoch_times: seconds since the C call was entered); `lead` = what passed on the host before that
    lead = t_call - start
    rows = [(0.0, 0.0, float(r["loss"][e]), lead + float(r["time"][e]), e, 0.0, 0.0) for e in range(E)]
    Q_dict = {f"Q_{k}": v.copy() for k, v in r["Q_pow2"].items()}
    if E > 0:
        Q_dict["Q_best"] = r["Q_best"].copy()
        Q_dict["Q_last"] = r["Q_last"].copy()
        Q_dict["result"] = (r["Q_best"] if return_best_iter else r["Q_last"]).copy()
    return rows, Q_dict


def train_quantization(rate_module, quantized_dataset, m=1.0, lr=1e-1, num_epochs=2000,
                       Q_true=None, optimizer=None, loss_normalization: bool = True,
                       return_best_iter: bool = True,
                       bank: Optional[CherryBank] = None,
                       fused: Optional[bool] = None, bank_dtype: str = "f64") -> Tuple[pd.DataFrame, Dict]:
    """Full-batch optimisation of `rate_module` on a TensorDataset(qtimes, cmats).

    Returns (df_res, Q_dict) exactly like the reference: per-epoch rows
    (nuc_norm, frob_norm, loss, time, epoch, frob_norm_diag, frob_norm_offdiag)
    and Q at epochs 1,2,4,..., "Q_best", "Q_last", "result".
    `bank_dtype` ("f64" / "f32", num_states > 32 only): element type of the bank products when the
    bank is built here (CherryBank's `dtype`; the reference computes them in float32).
    """
    logger = logging.getLogger(__name__)
    params = [p for p in rate_module.parameters()]
    device = params[0].device
    if device.type != "cuda":
        raise RuntimeError("cherryml_amd.train_quantization runs on the MI355X only: move the "
                           "rate module to device='cuda' (there is no CPU fallback)")
    reversible = rate_module.is_reversible()
    if optimizer is None:
        optimizer = torch.optim.SGD(rate_module.parameters(), lr=lr, momentum=0.0, weight_decay=0)
    own_bank = bank is None
    if own_bank:
        qtimes, cmats = quantized_dataset.tensors
        bank = CherryBank(qtimes.detach().cpu().numpy().astype(np.float64),
                          cmats.detach().cpu().numpy().astype(np.float64),
                          device=device.index or 0, dtype=bank_dtype)
    logger.info(f"Training for {num_epochs} epochs")
    Q_dict: Dict[str, np.ndarray] = {}
    rows = []
    best_loss, Q_best, Q = None, None, None
    start = time.time()
    use_fused = (reversible and _fusable(rate_module, optimizer, Q_true, m)) if fused is None else fused
    try:
        if use_fused:
            # the WHOLE loop (theta -> Q, bank, gradient, best-iterate bookkeeping, Adam) runs
            # on the device: S <= 32 is launch-latency bound and takes one to three launches per
            # epoch; larger S saves the torch glue (0.3 ms of a 4.4 ms epoch at S = 400).
            rows, Q_dict = _train_fused(rate_module, bank, optimizer, num_epochs,
                                        loss_normalization, return_best_iter)
            num_epochs_torch = 0
        else:
            num_epochs_torch = num_epochs
            if isinstance(optimizer, LazyOptimizer):
                optimizer = optimizer.materialize()
        for epoch in range(num_epochs_torch):
            optimizer.zero_grad()
            Q = rate_module()
            # non-reversible Q (non-symmetric mask, modes default/pande/stationary): general path
            loss = bank_loss(Q, rate_module.stationary() if reversible else None, bank,
                             normalize=loss_normalization)[0]
            if m != 1.0:
                loss = loss / m
            loss_value = float(loss.item())
            if best_loss is None or loss_value < best_loss:  # strict <, Q before the step
                best_loss, Q_best = loss_value, _snapshot(Q)
            if (epoch & (epoch + 1)) == 0:
                Q_dict[f"Q_{epoch + 1}"] = _snapshot(Q)
            loss.backward()
            optimizer.step()
            frob = frob_d = frob_o = nuc = 0.0
            if Q_true is not None:
                dif = (Q.detach() - torch.as_tensor(Q_true, device=Q.device, dtype=Q.dtype)) ** 2
                frob = float(torch.sqrt(dif.sum()))
                frob_d = float(torch.sqrt(dif.diag().sum()))
                frob_o = float(torch.sqrt((dif - torch.diag(dif.diag())).sum()))
            rows.append((nuc, frob, loss_value, time.time() - start, epoch, frob_d, frob_o))
    finally:
        if own_bank:
            bank.close()
    df_res = pd.DataFrame(rows, columns=["nuc_norm", "frob_norm", "loss", "time", "epoch",
                                         "frob_norm_diag", "frob_norm_offdiag"])
    logger.info(f"Total time = {time.time() - start}")
    if num_epochs > 0 and not use_fused:
        Q_dict["Q_best"] = Q_best.copy()
        Q_dict["Q_last"] = _snapshot(Q)
        Q_dict["result"] = Q_best.copy() if return_best_iter else _snapshot(Q)
    return df_res, Q_dict
