import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np
from conftest import load_golden
from cherryml_amd import CherryBank
sys.path.insert(0, ".")
import bench
rng = np.random.default_rng(0)
Q, pi, mask = bench.coevolution_truth(rng)
d = np.sqrt(pi); A = d[:, None] * Q / d[None, :]; A = 0.5 * (A + A.T)
bank = CherryBank(np.ones(2), np.ones((2, 400, 400)))
t0 = time.time(); lam, U = bank.eigh(A); print("time", time.time() - t0, "sweeps", bank.last_sweeps())
U = U[0]; lam = lam[0]
print("orth", np.abs(U.T @ U - np.eye(400)).max(), "resid", np.abs(A @ U - U * lam).max())
