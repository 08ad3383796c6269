"""dL/dQ of the bucket-sum-first form against the per-bucket third product (CB_BANK_K3=1) for several ranges of the near-pair
series (CB_PHI_Z: |t dlam / 2| <= z), on the reference's real bank and the bench bank.  python profiles/tools/r5_phi_sweep.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["CB_TEST_HOOKS"] = "1"
import bench  # noqa: E402
import torch  # noqa: E402
import cherryml_amd  # noqa: E402
from cherryml_amd import CherryBank  # noqa: E402
from cherryml_amd.estimation import jtt_ipw_from_arrays  # noqa: E402
from conftest import relerr  # noqa: E402

for name in ("coevo400_demo", "coevo400"):
    wl = bench.make_workload(name, 0, np.random.default_rng(0))
    init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
    mod = cherryml_amd.RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                                  pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
    Q, pi = mod().detach().numpy(), mod.stationary().detach().numpy()
    os.environ["CB_BANK_K3"] = "1"
    with CherryBank(wl["t"], wl["C"]) as bank:
        l0, d0 = bank.loss_grad(Q, pi)
    os.environ["CB_BANK_K3"] = "0"
    for z in ("0.02", "0.05", "0.1", "0.2", "0.3", "0.5"):
        os.environ["CB_PHI_Z"] = z
        with CherryBank(wl["t"], wl["C"]) as bank:
            l1, d1 = bank.loss_grad(Q, pi)
        print(name, "|z| <=", z, "rel. Frobenius of dL/dQ vs per-bucket:", f"{relerr(d1[0], d0[0]):.3e}", flush=True)
    os.environ.pop("CB_PHI_Z")
