"""Round 6: timing of tb_ew<16, 48> -- the elementwise kernel of the time basis with 33..48 gradient skeleton buckets (wider
spectral ranges than the bench optimisation ever reaches) -- at 400 states, beside the <16, 32> form of the headline: the bench
bank evaluated at k x its starting rate matrix.  Run on the GPU box from the repo root: python profiles/tools/r6_tb_ew48.py"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import bench
from conftest import load_golden
from cherryml_amd import CherryBank
wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
z = load_golden("coevo_dense_eval.npz")
keep = (wl["mask"] != 0) | np.eye(400, dtype=bool)
Q = np.zeros((400, 400)); Q[keep] = z["Q_support_f64"]
p = np.exp(z["log_pi"] - z["log_pi"].max()); pi = p / p.sum()
out = []
for k in (1.0, 4.0, 10.0, 30.0):
    with CherryBank(wl["t"], wl["C"]) as bank:
        bank.loss_grad(k * Q, pi)                      # builds the basis for this spectral range
        bank.profile(True)
        for _ in range(20):
            bank.loss_grad(k * Q, pi)
        tm = bank.timing_means()
        bank.profile(False)
        info, form = bank.time_basis_info(), bank.last_bank_form()
    rec = dict(scale=k, time_basis=form["time_basis"], forward_skeleton=info["forward_skeleton"], direct=info["direct"],
               gradient_skeleton=info["gradient_skeleton"], rho_max=info["rho_max"],
               tb_ew_form="<%d, %d>" % (16 if info["forward_skeleton"] <= 16 else 24, 32 if info["gradient_skeleton"] <= 32 else 48),
               forward_products_ms=tm.get("k1"), tb_ew_plus_mirror_ms=tm.get("k2"), gradient_products_ms=tm.get("k3"))
    out.append(rec)
    print(rec, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(round="r06", what=__doc__, runs=out), open("gpurun_out/r06_tb_ew48_timing.json", "w"), indent=1)
