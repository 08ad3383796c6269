"""How do K1 / K2 / K3 scale with the number of buckets (tile-count quantisation over the 256 CUs x 4
workgroup slots)?  Per-kernel HIP-event times of cb_loss_grad on the first B buckets of the bench bank."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from cherryml_amd import CherryBank
wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
from cherryml_amd.estimation import jtt_ipw_from_arrays
Q = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
pi = bench.stationary(Q)
dtype = sys.argv[1] if len(sys.argv) > 1 else "f64"
for B in (41, 64, 82, 96, 112, 122, 123, 126, 129):
    with CherryBank(wl["t"][:B], wl["C"][:B], dtype=dtype) as bank:
        for _ in range(3):
            bank.loss_grad(Q, pi)
        bank.profile(True)
        for _ in range(10):
            bank.loss_grad(Q, pi)
        tm = bank.timing_means()
    print(dtype, "B", B, "tiles k2", B * 25, "rounds", round(B * 25 / 1024, 2),
          {k: round(tm[k] * 1e3, 1) for k in ("k1", "k2", "k3")}, "us; per bucket k2", round(tm["k2"] * 1e3 / B, 3))
