"""cfg-5 bank (sum C = 1e8), 100 epochs from the JTT-IPW start, in the arithmetic modes and tile forms given on the command
line: which of them end with a finite loss curve.   python profiles/tools/r5_cfg5_repro.py f64:2 f64:1 mixed:2 ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["CB_TEST_HOOKS"] = "1"
import bench  # noqa: E402
import torch  # noqa: E402
from cherryml_amd import CherryBank, RateMatrix  # noqa: E402
from cherryml_amd.estimation import jtt_ipw_from_arrays  # noqa: E402

rng = np.random.default_rng(5)
Q, pi, mask = bench.coevolution_truth(rng)
t, C = bench.reversible_bank(Q, pi, 1.0e8, rng)
init = jtt_ipw_from_arrays(t, C, mask)
mod = RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(mask),
                 pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
u0, p0 = mod.upper_diag.detach().numpy().copy(), mod._pi.detach().numpy().copy()
ref = None
for spec in sys.argv[1:]:
    dtype, kg, *rest = spec.split(":")
    os.environ["CB_BANK_KG"] = kg
    if rest and rest[0] == "unfused":
        os.environ["CB_BANK_UNFUSED"] = "1"
    else:
        os.environ.pop("CB_BANK_UNFUSED", None)
    try:
        with CherryBank(t, C, dtype=dtype) as bank:
            r = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=100, lr=0.1)
            c = bank.eigh_counters()
        if ref is None:
            ref = r["loss"]
        print(spec, "ok: final loss", r["loss"][-1], "max |dloss| vs first spec", float(np.max(np.abs(r["loss"] - ref))), c, flush=True)
    except Exception as exc:
        print(spec, "FAILED:", exc, flush=True)

# single evaluations at the start point: loss and dL/dQ per form against float64 four-wave tiles
if os.environ.get("CB_REPRO_EVAL"):
    Q0 = init
    w, v = np.linalg.eig(Q0.T)
    p = v[:, int(np.argmin(np.abs(w.real)))].real
    p = p / p.sum()
    base = None
    for spec in os.environ["CB_REPRO_EVAL"].split():
        dtype, kg, *rest = spec.split(":")
        os.environ["CB_BANK_KG"] = kg
        if rest and rest[0] == "unfused":
            os.environ["CB_BANK_UNFUSED"] = "1"
        else:
            os.environ.pop("CB_BANK_UNFUSED", None)
        with CherryBank(t, C, dtype=dtype) as bank:
            loss, dQ = bank.loss_grad(Q0, p, normalize=True)
        if base is None:
            base = (loss[0], dQ[0])
        print("eval", spec, "loss", loss[0], "rel dloss", abs(loss[0] - base[0]) / abs(base[0]), "rel ddQ",
              float(np.linalg.norm(dQ[0] - base[1]) / np.linalg.norm(base[1])), "finite", bool(np.all(np.isfinite(dQ))), flush=True)
