"""Round 5: numpy prototype of the bank in a time basis (csrc/tbasis.hip.h) on the bench bank -- the numbers behind EXPERIMENTS
section 13.  (a) interpolative decomposition (long double, pivoted Gram-Schmidt) of the two families over the grid; (b) P_b,
G_b, loss and M = dL/dA-in-the-eigenbasis through the basis, in float64, against the per-bucket formula (tests/eigen_model.py);
(c) both against a long-double Taylor evaluation of P_b for a few short branches.
python profiles/tools/tb_prototype.py [coevo400|coevo400_demo]      (CPU, ~2 minutes)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from eigen_model import divided_difference, phi2  # noqa: E402
from cherryml_amd.estimation._jtt_ipw import jtt_ipw_from_statistics  # noqa: E402

LD = np.longdouble
S = 400


def phi2_ld(x):
    x = np.asarray(x, dtype=LD)
    small = np.abs(x) < 0.5
    xs = np.where(small, x, LD(0))
    acc = np.zeros_like(xs)
    for k in range(24, 1, -1):
        acc = (acc + 1) * xs / k if k > 2 else (acc + 1) * xs * xs / 2
    return np.where(small, acc, np.expm1(x) - x)


def id_rows(F, rmax, tol):
    """rows of F [B, N] -> skeleton rows and L [B, R] with F ~ L F[skel] (what tbasis::id_rows does in C++)"""
    B, N = F.shape
    W, Qs, Cq, skel = F.copy(), np.zeros((rmax, N), dtype=LD), np.zeros((B, rmax), dtype=LD), []
    scale = None
    for r in range(rmax):
        nr = (W * W).sum(1)
        p = int(np.argmax(nr))
        scale = scale or np.sqrt(nr[p])
        if np.sqrt(nr[p]) <= tol * scale:
            break
        q = W[p] / np.sqrt(nr[p])
        for _ in range(2):
            if r:
                q = q - Qs[:r].T @ (Qs[:r] @ q)
            q = q / np.sqrt((q * q).sum())
        Qs[r] = q
        c = W @ q
        Cq[:, r] = c
        W = W - np.outer(c, q)
        skel.append(p)
    R = len(skel)
    T = Cq[skel][:, :R]
    L = np.zeros((B, R), dtype=LD)
    for r in range(R - 1, -1, -1):
        L[:, r] = (Cq[:, r] - L[:, r + 1:] @ T[r + 1:, r]) / T[r, r]
    return skel, L, float(np.abs(F - L @ F[skel]).max())


def setup(Q):
    w, v = np.linalg.eig(Q.T)
    pi = np.real(v[:, np.argmin(np.abs(w))])
    pi = pi / pi.sum()
    d = np.sqrt(pi)
    A = d[:, None] * Q / d[None, :]
    A = 0.5 * (A + A.T)
    lam, U = np.linalg.eigh(A)
    return A, lam, U


def run(Q, t, C, label, growth=3.0):
    B, n = len(t), C.sum()
    A, lam, U = setup(Q)
    two_sigma = 2.0 * np.abs(np.diag(A)).max()
    rho_max = growth * two_sigma
    tl = t.astype(LD)
    mu = -LD(rho_max) * np.concatenate([[LD(0)], np.logspace(-6, 0, 512).astype(LD)])
    small = t * rho_max <= 8.0
    Fs = phi2_ld(tl[small][:, None] * mu[None, :]) / (tl[small][:, None] ** 2)
    sk_s, L_s, res_s = id_rows(Fs, 24, 1e-17)
    Fg = np.exp(tl[:, None] * mu[None, :])
    sk_g, L_g, res_g = id_rows(Fg, 40, 1e-16)
    ns, ng, nd = len(sk_s), len(sk_g), int((~small).sum())
    print(f"{label}: rho {np.abs(lam).max():.3f}, 2 sigma {two_sigma:.3f}, rho_max {rho_max:.2f} | ns {ns} (residual {res_s:.1e}) "
          f"nd {nd} ng {ng} ({res_g:.1e}) | tiles {(ns + nd) * 15 + ng * 40} vs {B * 40}")
    L_s = L_s.astype(np.float64)
    L_gs = (L_g * tl[:, None] / tl[sk_g][None, :]).astype(np.float64)      # carries t_b / t_skeleton
    ts = t[small]
    Psi = [(U * (phi2(ts[r] * lam) / ts[r] ** 2)) @ U.T for r in sk_s]
    idx_s = np.cumsum(small) - 1
    Gh = [np.zeros((S, S)) for _ in range(ng)]
    Mref, lossref, loss, perr, gerr = np.zeros((S, S)), 0.0, 0.0, 0.0, 0.0
    split_ref = t * 2.0 * np.abs(np.diag(A)).max() <= 1.0                 # the per-bucket kernels' rule
    for b in range(B):
        Pr = (np.eye(S) + t[b] * A + (U * phi2(t[b] * lam)) @ U.T) if split_ref[b] else (U * np.exp(t[b] * lam)) @ U.T
        nz = C[b] != 0
        lossref -= (C[b][nz] * np.log(Pr[nz])).sum() / n
        Gr = np.where(nz, -C[b] / np.where(nz, Pr, 1.0) / n, 0.0)
        Mref += (U.T @ Gr @ U) * divided_difference(lam, t[b])
        if small[b]:
            acc = np.zeros((S, S))
            for r in range(ns):
                acc += L_s[idx_s[b], r] * Psi[r]
            Pt = np.eye(S) + t[b] * (A + t[b] * acc)
        else:
            Pt = (U * np.exp(t[b] * lam)) @ U.T
        if nz.any():
            perr = max(perr, (np.abs(Pt - Pr) / np.abs(Pr))[nz].max())
        loss -= (C[b][nz] * np.log(Pt[nz])).sum() / n
        G = np.where(nz, -C[b] / np.where(nz, Pt, 1.0) / n, 0.0)
        gerr = max(gerr, np.linalg.norm(G - Gr) / np.linalg.norm(Gr))
        for r in range(ng):
            Gh[r] += L_gs[b, r] * G
    M = np.zeros((S, S))
    for r in range(ng):
        M += (U.T @ Gh[r] @ U) * divided_difference(lam, t[sk_g[r]])
    dA, dAr = U @ M @ U.T, U @ Mref @ U.T
    print(f"   vs per-bucket float64: P (counted entries) {perr:.1e} | G_b {gerr:.1e} | loss {abs(loss - lossref) / abs(lossref):.1e} | "
          f"dL/dA {np.linalg.norm(dA - dAr) / np.linalg.norm(dAr):.1e}")
    # long-double Taylor of exp(t A) for a few short branches: both forms against it
    Al = A.astype(LD)
    for b in (0, B // 4, B // 2):
        if not small[b]:
            continue
        term = np.eye(S, dtype=LD)
        P = np.eye(S, dtype=LD)
        for k in range(1, 40):
            term = (term @ Al) * (LD(t[b]) / k)
            P = P + term
            if float(np.abs(term).max()) < 1e-25:
                break
        Pold = np.eye(S) + t[b] * A + (U * phi2(t[b] * lam)) @ U.T
        acc = np.zeros((S, S))
        for r in range(ns):
            acc += L_s[idx_s[b], r] * Psi[r]
        Pnew = np.eye(S) + t[b] * (A + t[b] * acc)
        Pt = P.astype(np.float64)
        nz = C[b] != 0
        Gt, Go, Gn = np.where(nz, C[b] / Pt, 0), np.where(nz, C[b] / Pold, 0), np.where(nz, C[b] / Pnew, 0)
        print(f"   bucket {b} (t {t[b]:.1e}) vs long-double Taylor: G_b per-bucket {np.linalg.norm(Go - Gt) / np.linalg.norm(Gt):.1e}, "
              f"time basis {np.linalg.norm(Gn - Gt) / np.linalg.norm(Gt):.1e}")


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "coevo400"
    wl = bench.make_workload(name, 0, np.random.default_rng(0))
    t, C, mask = wl["t"], wl["C"], wl["mask"]
    live = C.sum((1, 2)) > 0
    t, C = t[live], C[live]
    Cs = 0.5 * (C + C.transpose(0, 2, 1))
    t0 = time.time()
    run(jtt_ipw_from_statistics(Cs.sum(0), (Cs / t[:, None, None]).sum(0), t, mask, True, 1e-8), t, C, "JTT-IPW start")
    Qt, _, _ = bench.coevolution_truth(np.random.default_rng(3))
    run(Qt, t, C, "perturbed product model")
    print(f"({time.time() - t0:.0f} s)")
