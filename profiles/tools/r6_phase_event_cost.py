"""Round 6: what the phase events of bench.py's timed window cost.  The headline's 20 timed epochs with cb_profile on (six
completion events per epoch, as bench.py runs them to report phase_ms) and off, alternating, same process, same optimisation
(prewarm 30, warm-up 5, then 20 resumed epochs).  Run on the GPU box from the repo root."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import bench, cherryml_amd
from cherryml_amd.estimation import jtt_ipw_from_arrays
wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
mod = cherryml_amd.RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                              pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
u0 = mod.upper_diag.detach().numpy().copy(); p0 = mod._pi.detach().numpy().copy()
out = {"with_events_ms": [], "without_events_ms": []}
with cherryml_amd.CherryBank(wl["t"], wl["C"]) as bank:
    bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=30, lr=0.1)
    for rep in range(6):
        on = rep % 2 == 0
        bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=5, lr=0.1)
        bank.profile(on)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=20, lr=0.1, resume=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20 * 1e3
        bank.profile(False)
        out["with_events_ms" if on else "without_events_ms"].append(round(dt, 4))
print(json.dumps(out))
