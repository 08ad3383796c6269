import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np, torch
import bench, cherryml_amd
from cherryml_amd.estimation import jtt_ipw_from_arrays
wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
mod = cherryml_amd.RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                              pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
u0, p0 = mod.upper_diag.detach().numpy().copy(), mod._pi.detach().numpy().copy()
bank = cherryml_amd.CherryBank(wl["t"], wl["C"], device=0)
call = lambda E, resume=False: bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=E, lr=0.1, resume=resume)
call(30)
for prof in (False, True, False, True, False, True):
    call(5)
    bank.profile(prof)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    call(20, True)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    bank.profile(False)
    print("profile", prof, "ms/epoch", round(dt / 20 * 1e3, 4), flush=True)
