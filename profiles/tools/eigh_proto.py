"""numpy prototype of the warm-started 400-state eigensolver schedule (eigh_large_host.hip.h) on the RECORDED parameter
trajectory of the bench optimisation (gpurun_out/r3_params.npz, written by record_trajectory.py on the GPU box).
Used to choose the sweep schedule before any kernel is written: it prints, per epoch, the cosine each sweep starts from
and a launch count under a simple cost model.  Not product, not oracle."""
import sys
import numpy as np

JB = 8


def build_A(upper, log_pi, mask):
    S = log_pi.shape[0]
    iu = np.triu_indices(S, 1)
    R = np.zeros((S, S))
    R[iu] = np.logaddexp(0.0, upper)
    R = (R + R.T) * mask
    pi = np.exp(log_pi - log_pi.max())
    pi /= pi.sum()
    root = np.sqrt(pi)
    Q = R * (root[None, :] / root[:, None])
    A = R.copy()
    A[np.arange(S), np.arange(S)] = -Q.sum(1)
    return A


def cosmax(Gam):
    d = np.diag(Gam)
    c = np.abs(Gam) / np.sqrt(np.outer(d, d))
    np.fill_diagonal(c, 0.0)
    return c.max()


def build_X(Gam, band):
    n = Gam.shape[0]
    d = np.diag(Gam)
    den = d[None, :] - d[:, None]
    np.fill_diagonal(den, 1.0)
    X = Gam / den
    np.fill_diagonal(X, 0.0)
    bi = np.arange(n) // JB
    far = np.abs(bi[:, None] - bi[None, :]) > band
    return X, X * far, far


def expm_antisym(X):
    # exact enough: scaling and squaring of a Taylor polynomial
    nrm = np.abs(X).sum(1).max()
    s = max(0, int(np.ceil(np.log2(max(nrm, 1e-300) / 0.05))))
    Y = X / 2.0 ** s
    R = np.eye(X.shape[0])
    T = np.eye(X.shape[0])
    for k in range(1, 14):
        T = T @ Y / k
        R = R + T
    for _ in range(s):
        R = R @ R
    return R


def group_diag(G, cols):
    """the within pass of one group: the columns made exactly orthogonal, kept in descending-norm order"""
    P = G[:, cols]
    w, V = np.linalg.eigh(P.T @ P)
    V = V[:, ::-1]
    G[:, cols] = P @ V


def cross_round(G, ci, cj):
    """one two-sided Jacobi sweep over the 64 cross pairs of blocks (ci, cj) on their 16 x 16 Gram matrix"""
    cols = np.concatenate([ci, cj])
    P = G[:, cols]
    Gm = P.T @ P
    R = np.eye(16)
    for a in range(8):
        for b in range(8, 16):
            # Brent-Luk order does not matter for the model
            p, q = a, b
            apq = Gm[p, q]
            if abs(apq) <= 1e-300:
                continue
            tau = (Gm[q, q] - Gm[p, p]) / (2.0 * apq)
            t = np.sign(tau) / (abs(tau) + np.sqrt(1.0 + tau * tau)) if tau != 0 else 1.0
            c = 1.0 / np.sqrt(1.0 + t * t)
            s = t * c
            J = np.eye(16)
            J[p, p] = c; J[q, q] = c; J[p, q] = s; J[q, p] = -s
            Gm = J.T @ Gm @ J
            R = R @ J
    G[:, cols] = P @ R


def band_pass(G, shift, band, nb):
    for w in range(2):
        odd = ((shift + w) & 1) and nb > 2
        for g in range(nb // 2):
            if odd:
                b0, b1 = 2 * g + 1, (2 * g + 2) % nb
            else:
                b0, b1 = 2 * g, 2 * g + 1
            cols = np.concatenate([np.arange(b0 * JB, b0 * JB + JB), np.arange(b1 * JB, b1 * JB + JB)])
            group_diag(G, cols)
    for k in range(2, band + 1):
        for par in range(2):
            for w in range(((nb + 2 * k - 1) // (2 * k)) * k):
                bi = (w // k) * 2 * k + par * k + (w % k)
                bj = bi + k
                if bj >= nb:
                    continue
                cross_round(G, np.arange(bi * JB, bi * JB + JB), np.arange(bj * JB, bj * JB + JB))


def window_pass(G, width, offset):
    """exact diagonalisation of every `width`-column window starting at offset (mod width)"""
    n = G.shape[1]
    start = offset
    if offset > 0:
        group_diag(G, np.arange(0, offset))
    while start < n:
        cols = np.arange(start, min(start + width, n))
        group_diag(G, cols)
        start += width


def solve(Ap, U_prev, variant, log):
    """returns (U, lam', list of (kind, cos) per sweep, launch count)"""
    n = Ap.shape[0]
    nb = n // JB
    band = variant.get("band", 3)
    trigger = 3e-4
    G = Ap @ U_prev
    launches = 3  # sigma, warm product, (theta->A not counted)
    hist = []
    for it in range(14):
        Gam = G.T @ G
        c = cosmax(Gam)
        X, Xf, far = build_X(Gam, band)
        rs = np.abs(X).sum(1).max()
        rsf = np.abs(Xf).sum(1).max()
        masked = c > trigger or rs > 0.5
        if masked and rsf > 12.0:
            hist.append(("refused", c))
            break
        if masked:
            Xu = Xf
            if variant.get("second_order"):
                # Schrieffer-Wolff second order for the far generator: [D, X2] = -([N + F/2, X1])_far
                d = np.diag(Gam)
                E = Gam - np.diag(d)
                N = E * (~far)
                F = E * far
                M = N + 0.5 * F
                Cm = M @ Xu - Xu @ M
                den = d[None, :] - d[:, None]
                np.fill_diagonal(den, 1.0)
                X2 = -(Cm / den) * far
                np.fill_diagonal(X2, 0.0)
                X2 = 0.5 * (X2 - X2.T)
                Xu = Xu + variant.get("so_scale", 1.0) * X2
                launches += 2
            R = expm_antisym(Xu)
            G = G @ R
            if variant.get("near") == "windows":
                w = variant.get("width", 32)
                window_pass(G, w, 0)
                window_pass(G, w, w // 2)
                launches += 2
            else:
                band_pass(G, it, band, nb)
                launches += 2 + 2 * (band - 1)
            # Gram+build, X2, X3X4, poly, R, NS1, NS2, GR (+ squaring when the far norm is above 0.5)
            launches += variant.get("masked_gemm_launches", 9) + (1 if rsf > 0.5 else 0)
            hist.append(("M", c, rs, rsf))
            continue
        R = expm_antisym(X)
        G = G @ R
        if rs <= 1e-5:
            launches += 4
        elif rs <= 2e-3:
            launches += 6
        else:
            sq = int(np.ceil(np.log2(rs / 0.075)))
            launches += 7 + sq + (2 if sq > 2 else 0)
        hist.append(("L", c, rs, rsf))
        if c <= 1e-8:
            break
    nrm = np.linalg.norm(G, axis=0)
    order = np.argsort(-nrm, kind="stable")
    U = -(G / nrm)[:, order]
    return U, nrm[order], hist, launches + 2


def main():
    d = np.load("gpurun_out/r3_params.npz")
    mask = np.unpackbits(d["mask"]).reshape(400, 400).astype(np.float64)
    E = d["upper"].shape[0]
    variants = {
        "current": {},
        "second_order": {"second_order": True},
        "windows32": {"near": "windows", "width": 32},
        "windows32+so": {"near": "windows", "width": 32, "second_order": True},
        "band2": {"band": 2},
    }
    pick = sys.argv[1:] or list(variants)
    for name in pick:
        v = variants[name]
        A = build_A(d["upper"][0], d["log_pi"][0], mask)
        sig = np.abs(np.diag(A)).max()
        lam, U = np.linalg.eigh(A - sig * np.eye(400))
        U = U[:, np.argsort(lam)]
        tot = 0
        print(f"== {name}")
        for e in range(1, E):
            A = build_A(d["upper"][e], d["log_pi"][e], mask)
            sig = np.abs(np.diag(A)).max()
            Ap = A - sig * np.eye(400)
            U, nrm, hist, launches = solve(Ap, U, v, False)
            res = np.abs(U.T @ A @ U - np.diag(np.diag(U.T @ A @ U))).max() / sig
            orth = np.abs(U.T @ U - np.eye(400)).max()
            if e >= 5:
                tot += launches
            print(f"  epoch {e:2d}: " + " ".join(f"{h[0]}{h[1]:.1e}" for h in hist) + f"  launches {launches}  resid {res:.1e} orth {orth:.1e}")
        print(f"  launches, epochs 5..{E - 1}: {tot}  ({tot / (E - 5):.1f} per solve)")


if __name__ == "__main__":
    main()
