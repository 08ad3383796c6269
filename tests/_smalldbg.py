import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, "tests")
import numpy as np, bench
from cherryml_amd import CherryBank
rng = np.random.default_rng(0)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = bench.make_workload("siterm", L, rng) if L > 1 else bench.make_workload("lg20", 1, rng)
bank = CherryBank(wl["t"], wl["C"])
if L > 1:
    Q = wl["init"]; pi = np.stack([bench.stationary(q) for q in Q[:2]]); pi = np.tile(pi, (L // 2 + 1, 1))[:L]
else:
    Q = bench.lg_matrix()[None] * 0.9; pi = bench.stationary(Q[0])[None]
for _ in range(2):
    loss, dQ = bank.loss_grad(Q, pi)
print("loss", loss[:2])
