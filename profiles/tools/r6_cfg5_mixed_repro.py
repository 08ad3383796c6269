"""Round 6 debugging aid: BASELINE config 5's bank in mixed precision, the epochs around a stalled planned eigensolve."""
import numpy as np, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, torch
from cherryml_amd import CherryBank, RateMatrix
from cherryml_amd.estimation import jtt_ipw_from_arrays
rng = np.random.default_rng(5)
Q, pi, mask = bench.coevolution_truth(rng)
t, C = bench.reversible_bank(Q, pi, 1.0e8, rng)
init = jtt_ipw_from_arrays(t, C, mask)
mod = RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(mask), pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
u0, p0 = mod.upper_diag.detach().numpy().copy(), mod._pi.detach().numpy().copy()
for E in [int(a) for a in sys.argv[1:]] or [100]:
    try:
        with CherryBank(t, C, dtype="mixed") as bank:
            r = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
            print("E", E, "ok; last losses", r["loss"][-4:], "finite Q_last", np.isfinite(r["Q_last"]).all(), bank.eigh_counters(), flush=True)
    except Exception as e:
        print("E", E, "FAILED", str(e)[:80], flush=True)
