"""wall time per epoch of the C-driven 400-state loop without profile markers (CB_TWO_STREAM experiment)"""
import sys, time
import numpy as np
sys.path.insert(0, ".")
import bench, torch, cherryml_amd
from cherryml_amd.estimation import jtt_ipw_from_arrays
wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
mod = cherryml_amd.RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                              pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
u0, p0 = mod.upper_diag.detach().numpy().copy(), mod._pi.detach().numpy().copy()
with cherryml_amd.CherryBank(wl["t"], wl["C"]) as bank:
    bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=5, lr=0.1)
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        r = bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=200, lr=0.1, resume=True)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(f"ms per epoch {dt / 200 * 1e3:.4f}  final loss {r['loss'][-1]:.12f}")
