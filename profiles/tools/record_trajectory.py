"""Record the parameters of the first epochs of the bench optimisation (one resumed epoch at a time) so that eigensolver
variants can be prototyped in numpy on the REAL sequence of matrices: gpurun_out/r3_params.npz."""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import bench, cherryml_amd
from cherryml_amd.estimation import jtt_ipw_from_arrays
rng = np.random.default_rng(0)
wl = bench.make_workload("coevo400", 0, rng)
init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
mod = cherryml_amd.RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                              pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
u0 = mod.upper_diag.detach().numpy().copy(); p0 = mod._pi.detach().numpy().copy()
E = int(sys.argv[1]) if len(sys.argv) > 1 else 26
ups, pis = [u0.astype(np.float32)], [p0]
with cherryml_amd.CherryBank(wl["t"], wl["C"]) as bank:
    r = bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=1, lr=0.1)
    for e in range(1, E):
        ups.append(r["upper_diag"].copy()); pis.append(r["log_pi"].copy())
        r = bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=1, lr=0.1, resume=True)
np.savez_compressed("gpurun_out/r3_params.npz", upper=np.array(ups, dtype=np.float64), log_pi=np.array(pis), mask=np.packbits(wl["mask"].astype(np.uint8)))
print("saved", len(ups))
