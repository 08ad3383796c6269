"""Round 5: accuracy of the bucket sum BEFORE the last product (csrc/large_bank.hip.h, ky_reduce_loss / kphi_combine) on the
bench bank, in float64 numpy, against an 80-bit evaluation of the same formula the per-bucket kernels compute:
per-bucket divided differences (today) vs differences of the one-sided sums (far pairs) + the tanh(z)/z series (near pairs),
for several thresholds and term counts.   python profiles/tools/accumulated_phi_model.py   (CPU, ~2 minutes)"""
import sys, time
import numpy as np
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
from eigen_model import phi2, divided_difference
from cherryml_amd.estimation._jtt_ipw import jtt_ipw_from_statistics

rng = np.random.default_rng(0)
wl = bench.make_workload("coevo400", 0, rng)
t_all, C_all, mask = wl["t"], wl["C"], wl["mask"]
sel = np.arange(0, 129, 4)          # 33 buckets spread over the grid (longdouble reference is slow)
t, C = t_all[sel], C_all[sel]
S = 400
n = C.sum()

def setup(Q):
    w, v = np.linalg.eig(Q.T)
    pi = np.real(v[:, np.argmin(np.abs(w))]); pi = pi / pi.sum()
    d = np.sqrt(pi)
    A = d[:, None] * Q / d[None, :]
    A = 0.5 * (A + A.T)
    lam, U = np.linalg.eigh(A)
    return A, lam, U

def run(Q, label):
    A, lam, U = setup(Q)
    Gs = []
    for b in range(len(t)):
        Pt = np.eye(S) + t[b] * A + (U * phi2(t[b] * lam)) @ U.T
        Gs.append(-C[b] / Pt / n)
    # (a) per bucket, float64
    Ma = np.zeros((S, S))
    Ts = []
    for b in range(len(t)):
        T = Gs[b] @ U
        Ts.append(T)
        Ma += (U.T @ T) * divided_difference(lam, t[b])
    # (b) accumulated, float64
    E = np.exp(t[:, None] * lam[None, :])          # [B, S]
    Y1 = np.zeros((S, S)); Y2 = np.zeros((S, S)); Y3 = np.zeros((S, S))
    for b in range(len(t)):
        Y1 += Ts[b] * E[b][None, :]
        Y2 += Ts[b] * (t[b] * E[b])[None, :]
        Y3 += Ts[b] * (t[b] ** 3 * E[b])[None, :]
    L1 = Y1.T @ U; L2 = Y2.T @ U; L3 = Y3.T @ U
    dl = lam[:, None] - lam[None, :]
    out = {}
    ak = [1.0, -1.0/3, 2.0/15, -17.0/315, 62.0/2835, -1382.0/155925, 21844.0/6081075, -929569.0/638512875]
    Lk = []
    for k in range(len(ak)):
        Yk = np.zeros((S, S))
        for b in range(len(t)):
            Yk += Ts[b] * (t[b] ** (2 * k + 1) * E[b])[None, :]
        Lk.append(Yk.T @ U)
    with np.errstate(divide="ignore", invalid="ignore"):
        quot = (L1 - L1.T) / dl
    for K in (4, 6, 8):
        for delta in (5e-3, 1e-2, 2e-2, 5e-2):
            near = np.abs(dl) < delta
            tay = np.zeros((S, S))
            for k in range(K):
                tay += ak[k] * (dl * dl / 4.0) ** k * 0.5 * (Lk[k] + Lk[k].T)
            out[(K, delta)] = np.where(near, tay, quot)
    # reference: longdouble
    t0 = time.time()
    Ul = U.astype(np.longdouble); laml = lam.astype(np.longdouble)
    Mr = np.zeros((S, S), dtype=np.longdouble)
    for b in range(len(t)):
        W = Ul.T @ (Gs[b].astype(np.longdouble) @ Ul)
        x = np.longdouble(t[b]) * laml
        dx = x[:, None] - x[None, :]
        Eb = np.exp(x)
        with np.errstate(divide="ignore", invalid="ignore"):
            phi = np.where(np.abs(dx) > 1e-9, (Eb[:, None] - Eb[None, :]) / (laml[:, None] - laml[None, :]),
                           np.longdouble(t[b]) * np.exp((x[:, None] + x[None, :]) / 2) * (1 + dx * dx / 24))
        Mr += W * phi
    dAr = (Ul @ Mr @ Ul.T)
    def err(M):
        dA = U @ M @ U.T
        return float(np.linalg.norm((dA - dAr).astype(np.float64)) / np.linalg.norm(dAr.astype(np.float64))), \
               float(np.linalg.norm((M - Mr).astype(np.float64)) / np.linalg.norm(Mr.astype(np.float64)))
    gaps = np.sort(np.abs(np.diff(lam)))
    print(label, "smallest gaps", gaps[:5], "median gap", np.median(gaps), "sigma", np.abs(np.diag(A)).max(), f"(ref {time.time()-t0:.0f} s)")
    print("  per-bucket (today)   : dA err %.2e, M err %.2e" % err(Ma))
    for (K, delta), Mb in out.items():
        print("  accumulated, %d terms, near<%g: dA err %.2e, M err %.2e  (near pairs %d)" % ((K, delta) + err(Mb) + (int((np.abs(dl) < delta).sum() - S) // 2,)))

Cs = 0.5 * (C_all + C_all.transpose(0, 2, 1))
init = jtt_ipw_from_statistics(Cs.sum(0), (Cs / t_all[:, None, None]).sum(0), t_all, mask, True, 1e-8)
run(init, "JTT-IPW start")
Qt, pit, _ = bench.coevolution_truth(np.random.default_rng(3))
run(Qt, "generating model (perturbed product model)")
