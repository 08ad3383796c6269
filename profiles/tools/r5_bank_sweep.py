"""The bank launch against the number of live buckets, in its four forms (four- / eight-wave tiles x one persistent launch /
three launches): where the shape-based choice of large_eval (cherrybank.hip) comes from.
    python profiles/tools/r5_bank_sweep.py [epochs] > gpurun_out/r5_bank_sweep.json"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["CB_TEST_HOOKS"] = "1"
import bench  # noqa: E402
import torch  # noqa: E402
import cherryml_amd  # noqa: E402
from cherryml_amd.estimation import jtt_ipw_from_arrays  # noqa: E402

E = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(0)
wl = bench.make_workload("coevo400", 0, rng)
init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
mod = cherryml_amd.RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                              pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
u0, p0 = mod.upper_diag.detach().numpy().copy(), mod._pi.detach().numpy().copy()
out = {}
for nb in (4, 8, 12, 17, 24, 32, 43, 56, 72, 96, 129):
    sel = np.linspace(0, 128, nb).round().astype(int)
    row = {}
    with cherryml_amd.CherryBank(wl["t"][sel], wl["C"][sel], device=0) as bank:
        for form in ("kg1_fused", "kg1_unfused", "kg2_fused", "kg2_unfused"):
            os.environ["CB_BANK_KG"] = form[2]
            if form.endswith("unfused"):
                os.environ["CB_BANK_UNFUSED"] = "1"
            else:
                os.environ.pop("CB_BANK_UNFUSED", None)
            bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=8, lr=0.1)
            bank.profile(True)
            bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=E, lr=0.1)
            tm = bank.timing_means()
            bank.profile(False)
            row[form] = round(tm["k1"] + tm["k2"] + tm["k3"], 4)
    row["best"] = min(row, key=row.get)
    out[nb] = row
    print(nb, row, file=sys.stderr, flush=True)
print(json.dumps(out, indent=1))
