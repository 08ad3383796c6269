import os, sys, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
from cherryml_amd import CherryBank
from conftest import load_golden
wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
z = load_golden("coevo_dense_traj.npz")
u0, p0, mask = z["upper_diag0"], z["log_pi0"], wl["mask"]
for nb in (24, 32, 40):
    sel = np.linspace(0, 128, nb).round().astype(int)
    t, C = wl["t"][sel], wl["C"][sel]
    res = {}
    for tb in ("0", "1"):
        os.environ["CB_BANK_TB"] = tb
        with CherryBank(t, C) as bank:
            bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=30, lr=0.1)
            t0 = time.time()
            r = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=100, lr=0.1, resume=True)
            dt = (time.time() - t0) / 100 * 1e3
            res[tb] = (dt, r["loss"][-1], bank.last_bank_form(), bank.time_basis_info() if tb == "1" else None)
    print(nb, "buckets: per-bucket forms %.4f ms (%s) | time basis %.4f ms %s | loss diff %.1e" % (
        res["0"][0], {k: v for k, v in res["0"][2].items() if v}, res["1"][0], res["1"][3], abs(res["0"][1] - res["1"][1]) / abs(res["0"][1])))
