"""`RateMatrixLearner` (reference: cherryml/estimation/_ratelearn/ratelearner.py:34-184):
same constructor, `.train(...)`, `.get_learnt_rate_matrix()`, attributes
`df_res`, `Q_dict`, and the same set of output files."""
import logging
import os
from typing import List, Optional

import numpy as np
import pandas as pd
import torch
from torch.utils.data import TensorDataset

from ...io import write_rate_matrix
from ._epoch_loop import LazyOptimizer, train_quantization
from ._rate_matrix import RateMatrix


class RateMatrixLearner:
    def __init__(self, branches: List[float], mats: List[np.ndarray], states: List[str],
                 output_dir: str, stationnary_distribution: str, device: str, mask: str = None,
                 rate_matrix_parameterization="pande_reversible",
                 initialization: Optional[np.ndarray] = None,
                 skip_writing_to_output_dir: bool = False, bank_dtype: str = "f64"):
        self.branches = branches
        self.mats = mats
        self.states = states
        self.output_dir = None if skip_writing_to_output_dir else output_dir
        self.stationnary_distribution = stationnary_distribution
        self.mask = mask
        self.rate_matrix_parameterization = rate_matrix_parameterization
        self.device = device
        self.initialization = initialization
        self.skip_writing_to_output_dir = skip_writing_to_output_dir
        self.bank_dtype = bank_dtype   # extra to the reference: "f32" = its own float32 arithmetic on the f32 MFMA (S > 32)
        self.lr = None
        self.do_adam = None
        self.df_res = None
        self.Q_dict = None
        self.trained = False

    def train(self, lr=1e-1, num_epochs=2000, do_adam: bool = True,
              loss_normalization: bool = False, return_best_iter: bool = True):
        logger = logging.getLogger(__name__)
        logger.info(f"Starting, outdir: {self.output_dir}")
        from ..._device import resolve_device
        resolve_device(self.device, "RateMatrixLearner")   # "cpu" and "cuda" both mean the MI355X (see _device.py)
        torch.manual_seed(0)  # ratelearner.py:77
        if not self.skip_writing_to_output_dir:
            os.makedirs(self.output_dir, exist_ok=True)
        S = int(np.asarray(self.mats[0]).shape[0])
        self.n_states = S
        qtimes = torch.tensor(np.asarray(self.branches, dtype=np.float64))
        cmats = torch.tensor(np.asarray(self.mats, dtype=np.float64))
        self.quantized_data = TensorDataset(qtimes, cmats)

        src = self.stationnary_distribution
        if src is None:
            pi = np.full(S, 1.0 / S)
        elif isinstance(src, str):
            pi = pd.read_csv(src, header=None, index_col=None).values.squeeze()
        else:  # the stage function hands over an array (_quantized_transitions_mle.py:82-85)
            pi = np.asarray(src, dtype=np.float64).squeeze()
        self.pi = torch.tensor(pi, dtype=torch.float64)
        if self.mask is not None:
            mask_mat = pd.read_csv(self.mask, sep=r"\s+", header=None, index_col=None).values
        else:
            mask_mat = np.ones((S, S))
        self.mask_mat = torch.tensor(mask_mat, dtype=torch.float64)

        self.mat_module = RateMatrix(
            num_states=S, mode=self.rate_matrix_parameterization, pi=self.pi,
            pi_requires_grad=src is None, initialization=self.initialization,
            mask=self.mask_mat).to(device="cuda")
        self.lr, self.do_adam = lr, do_adam
        optim = LazyOptimizer(self.mat_module.parameters(), lr=lr, do_adam=do_adam)   # torch's, built only if the torch loop runs
        self.df_res, self.Q_dict = train_quantization(
            rate_module=self.mat_module, quantized_dataset=self.quantized_data,
            num_epochs=num_epochs, Q_true=None, optimizer=optim,
            loss_normalization=loss_normalization, return_best_iter=return_best_iter,
            bank_dtype=self.bank_dtype)
        self.trained = True
        if not self.skip_writing_to_output_dir:
            self.process_results()

    def process_results(self):
        for key, value in self.Q_dict.items():
            write_rate_matrix(value, self.states, os.path.join(self.output_dir, key + ".txt"))
        self.df_res.to_csv(os.path.join(self.output_dir, "df_res.txt"))
        try:  # the reference also drops a loss plot; optional here (matplotlib may be absent)
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
            plt.figure(figsize=(5, 4))
            self.df_res.loss.plot()
            plt.xscale("log")
            plt.ylabel("Negative likelihood", fontsize=13)
            plt.xlabel("# of iterations", fontsize=13)
            plt.tight_layout()
            plt.savefig(os.path.join(self.output_dir, "training_plot.png"))
            plt.close()
        except Exception:  # pragma: no cover
            pass

    def get_learnt_rate_matrix(self) -> pd.DataFrame:
        if not self.trained:
            raise ValueError("Model should be trained first!")
        return pd.DataFrame(self.Q_dict["result"], columns=self.states, index=self.states)
