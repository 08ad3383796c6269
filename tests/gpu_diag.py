"""First-contact diagnostic on a GPU box: prints errors stage by stage."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from conftest import load_golden, relerr
from oracle import ratelearn_oracle as orc
from cherryml_amd import CherryBank

def pi_of(lp):
    p = np.exp(lp - lp.max()); return p / p.sum()

for case in ["toy3_init", "s20_symmask", "s400_mask"]:
    g = load_golden(f"eval_{case}.npz")
    Q, pi = g["Q_f64"], pi_of(g["log_pi"])
    S = Q.shape[0]
    d = np.sqrt(pi); A = d[:, None] * Q / d[None, :]; A = 0.5 * (A + A.T)
    try:
        bank = CherryBank(g["t"], g["C"])
        print(case, "n", bank.total_counts, "ref", g["C"].sum())
        t0 = time.time(); lam, U = bank.eigh(A); t1 = time.time()
        lam, U = lam[0], U[0]
        print(case, "eigh: orth", np.abs(U.T @ U - np.eye(S)).max(), "resid", np.abs(A @ U - U * lam[None]).max(),
              "lam err", np.abs(np.sort(lam) - np.linalg.eigvalsh(A)).max(), f"time {t1-t0:.4f}s")
        P = bank.expm_bank(Q, pi)[0]
        ref = orc.expm_bank(Q, g["t"])
        print(case, "expm: abs", np.abs(P - ref).max(), "rel", np.abs(P / ref - 1).max())
        t0 = time.time(); loss, dQ = bank.loss_grad(Q, pi); t1 = time.time()
        print(case, "loss", loss[0], "ref", float(g["loss_f64"]), "rel", abs(loss[0] - g["loss_f64"]) / abs(g["loss_f64"]),
              "dQ rel", relerr(dQ[0], g["dQ_f64"]), f"time {t1-t0:.4f}s")
        for _ in range(3):
            t0 = time.time(); loss, dQ = bank.loss_grad(Q, pi); t1 = time.time()
            print(case, f"  repeat time {t1-t0:.4f}s loss {loss[0]!r}")
        bank.close()
    except Exception as e:
        import traceback; traceback.print_exc()
