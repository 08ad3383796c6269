"""Round-6 numpy model of the planned warm eigensolve (eigh_planned.hip.h) on the RECORDED Adam trajectory of the bench
optimisation (gpurun_out/r3_params.npz, profiles/tools/record_trajectory.py).  Models what the device does today -- Jacobi
angles (atan) in the generator, the second-order generator of lge_so, band = 2 blocks of 8 columns, the schedule
band / far rotation / band / all-pairs sweeps, the final rule c |X|^2 <= 2e-14 -- and the round-6 candidates that cut the
number of DEPENDENT launches.  Prints per epoch the sweep sequence and a launch count.  Not product, not oracle."""
import sys
import numpy as np

JB = 8
SKIP = 1e-16
NORM = "rowsum"


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


def stats_X(Gam, band, angles=True):
    """lge_gram's epilogue: generator (Jacobi angles), far mask, cosine, row sums"""
    n = Gam.shape[0]
    d = np.diag(Gam)
    ab = np.outer(d, d)
    g2 = Gam * Gam
    live = g2 > ab * SKIP * SKIP
    np.fill_diagonal(live, False)
    c = np.sqrt((g2 / ab)[live].max()) if live.any() else 0.0
    den = d[None, :] - d[:, None]
    den[den == 0.0] = 1e-300
    X = 0.5 * np.arctan(2.0 * Gam / den) if angles else Gam / den
    X = X * live
    bi = np.arange(n) // JB
    far = np.abs(bi[:, None] - bi[None, :]) > band
    Xf = X * far
    if NORM == "fro":   # (round 6: would a single global sum do instead of 400 row sums?  |X|_2 <= |X|_F)
        return X, Xf, far, c, np.sqrt((X * X).sum()), np.sqrt((Xf * Xf).sum())
    if NORM == "cmp":
        print(f"      rowsum {np.abs(X).sum(1).max():.2e} fro {np.sqrt((X * X).sum()):.2e} two {np.linalg.norm(X, 2):.2e} | far: rowsum "
              f"{np.abs(Xf).sum(1).max():.2e} fro {np.sqrt((Xf * Xf).sum()):.2e} two {np.linalg.norm(Xf, 2):.2e}")
    return X, Xf, far, c, np.abs(X).sum(1).max(), np.abs(Xf).sum(1).max()


def so_correct(Gam, X1):
    d = np.diag(Gam)
    E = Gam - np.diag(d)
    C = E @ X1
    C = C + C.T
    den = d[None, :] - d[:, None]
    den[den == 0.0] = np.inf
    X2 = 0.5 * C / den
    X2[X1 == 0.0] = 0.0
    return X1 + X2


def expm_antisym(X):
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


def jacobi16(Gm, pairs, R):
    for p, q in pairs:
        apq = Gm[p, q]
        if apq * apq <= Gm[p, p] * Gm[q, q] * SKIP * SKIP:
            continue
        d = Gm[q, q] - Gm[p, p]
        th = 0.5 * np.arctan2(2.0 * apq, d) if d != 0 else np.pi / 4 * np.sign(apq)
        if th > np.pi / 4:
            th -= np.pi / 2
        if th < -np.pi / 4:
            th += np.pi / 2
        c, s = np.cos(th), np.sin(th)
        J = np.eye(Gm.shape[0])
        J[p, p] = c; J[q, q] = c; J[p, q] = s; J[q, p] = -s
        Gm[:] = J.T @ Gm @ J
        R[:] = R @ J


ALL16 = [(p, q) for p in range(16) for q in range(p + 1, 16)]
CROSS16 = [(p, q) for p in range(8) for q in range(8, 16)]


def group_rot(G, cols, inner, exact=False):
    P = G[:, cols]
    Gm = P.T @ P
    if exact:
        w, V = np.linalg.eigh(Gm)
        V = V[:, ::-1]
        # keep the columns where they were (largest overlap): eigh's order is by eigenvalue, the device keeps positions
        G[:, cols] = P @ V
        return
    R = np.eye(len(cols))
    if inner > 0:
        for _ in range(inner):
            jacobi16(Gm, ALL16 if len(cols) == 16 else [(p, q) for p in range(len(cols)) for q in range(p + 1, len(cols))], R)
    else:
        jacobi16(Gm, CROSS16, R)
    G[:, cols] = P @ R


def band_pass(G, shift, band, inner, exact=False, tilings=2):
    n = G.shape[1]
    nb = n // JB
    for w in range(tilings):
        odd = ((shift + w) & 1) and nb > 2
        for g in range(nb // 2):
            b0, b1 = (2 * g + 1, (2 * g + 2) % nb) if odd else (2 * g, 2 * g + 1)
            cols = np.concatenate([np.arange(b0 * JB, b0 * JB + JB), np.arange(b1 * JB, b1 * JB + JB)])
            group_rot(G, cols, inner, exact)
    for k in range(2, band + 1):
        for par in range(2):
            for w in range(((nb + 2 * k - 1) // (2 * k)) * k):
                bi = (w // k) * 2 * k + par * k + (w % k)
                bj = bi + k
                if bj >= nb:
                    continue
                cols = np.concatenate([np.arange(bi * JB, bi * JB + JB), np.arange(bj * JB, bj * JB + JB)])
                group_rot(G, cols, 0)


def solve_current(Ap, U_prev, v):
    """the device's schedule today; returns U, hist, launches (as enqueued, spare included)"""
    G = Ap @ U_prev
    band = v.get("band", 2)
    launches = 2
    hist = []
    shift = 0
    if v.get("lead", True):
        band_pass(G, shift, band, v.get("lead_inner", 2), v.get("exact", False), v.get("lead_tilings", 2))
        shift += 1
        launches += v.get("lead_tilings", 2) * band
    for it in range(12):
        Gam = G.T @ G
        X, Xf, far, c, rs, rsf = stats_X(Gam, band)
        masked = c > v.get("trigger", 3e-4) or rs > v.get("rs_max", 1.0)
        if it >= v.get("force_L_from", 99):
            masked = False
        so = (it >= 1) and not masked and c > 1e-8 and rs <= v.get("so_max", 1e9)
        final = (not masked) and (c <= 1e-8 or (so and c * rs * rs <= 2e-14))
        if masked:
            Xu = Xf
        elif so:
            Xu = so_correct(Gam, X)
        else:
            Xu = X
        G = G @ expm_antisym(Xu)
        rsu = rsf if masked else rs
        order = 2 if rsu <= 1e-5 else 4 if rsu <= 2e-3 else 8 if rsu <= 0.06 else 12
        launches += {2: 5, 4: 6, 8: 8, 12: 9}[order] + (1 if it >= 1 else 0)
        if masked:
            if v.get("post", True):
                band_pass(G, shift, band, v.get("post_inner", 1), v.get("exact", False), v.get("post_tilings", 2))
                shift += 1
                launches += v.get("post_tilings", 2) * band
        hist.append(("M" if masked else "L", c, rs, rsf, order))
        if final:
            break
    launches += 6 + 2
    return G, hist, launches


def windows(n, width, offset):
    out = []
    if offset > 0:
        out.append(np.arange(0, offset))
    s = offset
    while s < n:
        out.append(np.arange(s, min(s + width, n)))
        s += width
    return out


def solve_gamma_w(Ap, U_prev, v):
    """candidate: every sweep = ONE Gram; the near pairs of a tiling (width-column windows, offset alternating) are rotated
    EXACTLY from the diagonal blocks of Gamma (W, block diagonal), the generator of all other pairs is taken from
    W^T Gamma W (second order when asked), and G <- G W exp(X) is one product.  No band pass touches G."""
    n = Ap.shape[0]
    width = v.get("width", 16)
    G = Ap @ U_prev
    hist = []
    launches = 1
    for it in range(12):
        Gam = G.T @ G
        d0 = np.diag(Gam)
        c_all = np.abs(Gam) / np.sqrt(np.outer(d0, d0))
        np.fill_diagonal(c_all, 0.0)
        W = np.eye(n)
        offs = v.get("offsets", [0, width // 2])
        wins = windows(n, width, offs[it % len(offs)])
        inwin = np.zeros((n, n), dtype=bool)
        for cols in wins:
            B = Gam[np.ix_(cols, cols)]
            w_, V = np.linalg.eigh(B)
            V = V[:, ::-1]
            # keep columns near their old places: permute so that the largest component sits on the diagonal (sorted spectrum)
            W[np.ix_(cols, cols)] = V
            inwin[np.ix_(cols, cols)] = True
        Gp = W.T @ Gam @ W
        d = np.diag(Gp)
        ab = np.outer(d, d)
        live = (Gp * Gp > ab * SKIP * SKIP) & ~inwin
        if it < v.get("far_sweeps", 0):
            bi = np.arange(n) // JB
            live &= np.abs(bi[:, None] - bi[None, :]) > v.get("band", 2)
        c = np.sqrt(((Gp * Gp) / ab)[live].max()) if live.any() else 0.0
        den = d[None, :] - d[:, None]
        den[den == 0.0] = 1e-300
        X = 0.5 * np.arctan(2.0 * Gp / den) * live
        rs = np.abs(X).sum(1).max()
        use_so = v.get("so", True) and c > 1e-8
        if use_so:
            E = (Gp - np.diag(d)) * ~inwin if v.get("so_excl", True) else Gp - np.diag(d)
            C = E @ X
            C = C + C.T
            den2 = den.copy()
            X2 = 0.5 * C / den2
            X2[~live] = 0.0
            Xu = X + X2
        else:
            Xu = X
        cin = c_all[inwin].max()
        final = (c <= 1e-8 or (use_so and c * rs * rs <= 2e-14)) and cin <= v.get("cin_final", 1e-8)
        G = G @ (W @ expm_antisym(Xu))
        order = 2 if rs <= 1e-5 else 4 if rs <= 2e-3 else 7 if rs <= 0.06 else 12
        # gram, wgen(+decide), [so], exp chain, GR
        launches += 2 + (1 if use_so else 0) + {2: 1, 4: 2, 7: 3, 12: 4}[order] + (1 if rs > 0.5 else 0) + (1 if rs > 1.0 else 0) + 1
        hist.append(("W", c, rs, cin, order))
        if final:
            break
    launches += 2
    return G, hist, launches


def win_rot(B, mode):
    """rotation of one window's Gram block: 'exact' or k cyclic Jacobi sweeps (k = int)"""
    m = B.shape[0]
    if mode == "exact":
        w_, V = np.linalg.eigh(B)
        return V[:, ::-1]
    Gm = B.copy()
    R = np.eye(m)
    pairs = [(p, q) for p in range(m) for q in range(p + 1, m)]
    for _ in range(int(mode)):
        jacobi16(Gm, pairs, R)
    return R


def solve_gs(Ap, U_prev, v):
    """general Gamma-space sweep: one Gram per sweep, a LIST of window stages (offset, width, mode) whose rotations are found on
    the band of Gamma and applied two-sided there, then the generator of exp(X) from the rotated Gamma over `pairs` =
    'all' | 'out' (outside the last stage's windows) ; sweep s < far_sweeps: far pairs only."""
    n = Ap.shape[0]
    G = Ap @ U_prev
    hist = []
    launches = 1
    plan = v["stages"]           # list (per sweep, last entry repeats) of lists of (offset, width, mode)
    for it in range(12):
        Gam = G.T @ G
        stages = plan[min(it, len(plan) - 1)]
        W = np.eye(n)
        Gp = Gam.copy()
        inwin = np.zeros((n, n), dtype=bool)
        for (off, width, mode) in stages:
            Ws = np.eye(n)
            inwin[:] = False
            for cols in windows(n, width, off):
                Ws[np.ix_(cols, cols)] = win_rot(Gp[np.ix_(cols, cols)], mode)
                inwin[np.ix_(cols, cols)] = True
            Gp = Ws.T @ Gp @ Ws
            W = W @ Ws
        d = np.diag(Gp)
        ab = np.outer(d, d)
        live = Gp * Gp > ab * SKIP * SKIP
        np.fill_diagonal(live, False)
        if v.get("pairs", "all") == "out":
            live &= ~inwin
        if it < v.get("far_sweeps", 0):
            bi = np.arange(n) // JB
            live &= np.abs(bi[:, None] - bi[None, :]) > v.get("band", 2)
        c = np.sqrt(((Gp * Gp) / ab)[live].max()) if live.any() else 0.0
        den = d[None, :] - d[:, None]
        den[den == 0.0] = 1e-300
        X = 0.5 * np.arctan(2.0 * Gp / den) * live
        rs = np.abs(X).sum(1).max()
        call = np.abs(Gp) / np.sqrt(ab)
        np.fill_diagonal(call, 0.0)
        cmax_all = call.max()
        use_so = v.get("so", True) and c > 1e-8 and it >= v.get("so_from", 0)
        if use_so:
            E = Gp - np.diag(d)
            C = E @ X
            C = C + C.T
            X2 = 0.5 * C / den
            X2[~live] = 0.0
            Xu = X + X2
        else:
            Xu = X
        final = (cmax_all <= 1e-8 or (use_so and cmax_all * rs * rs <= 2e-14 and c == cmax_all))
        G = G @ (W @ expm_antisym(Xu))
        order = 2 if rs <= 1e-5 else 4 if rs <= 2e-3 else 7 if rs <= 0.06 else 12
        launches += 2 + (1 if use_so else 0) + {2: 1, 4: 2, 7: 3, 12: 4}[order] + (1 if rs > 0.5 else 0) + (1 if rs > 1.0 else 0) + 1
        hist.append(("S", c, rs, cmax_all, order))
        if final:
            break
    launches += 2
    return G, hist, launches


def finish(G):
    nrm = np.linalg.norm(G, axis=0)
    order = np.argsort(-nrm, kind="stable")
    return -(G / nrm)[:, order]


def main():
    d = np.load("gpurun_out/r3_params.npz")
    mask = np.unpackbits(d["mask"]).reshape(400, 400).astype(np.float64)
    E = d["upper"].shape[0]
    variants = {
        "current": (solve_current, {}),
        "cur_b1": (solve_current, {"band": 1}),
        "b1_so03": (solve_current, {"band": 1, "so_max": 0.3}),
        "b1_so03_trig1e-3": (solve_current, {"band": 1, "so_max": 0.3, "trigger": 1e-3}),
        "b1_so05_trig1e-3": (solve_current, {"band": 1, "so_max": 0.5, "trigger": 1e-3}),
        "b1_so1_trig1e-3": (solve_current, {"band": 1, "so_max": 1.0, "trigger": 1e-3}),
        "b1_trig6e-4": (solve_current, {"band": 1, "trigger": 6e-4}),
        "b1_trig1e-3": (solve_current, {"band": 1, "trigger": 1e-3}),
        "b1_trig1e-3_rs2": (solve_current, {"band": 1, "trigger": 1e-3, "rs_max": 2.0}),
        "b1_lead1": (solve_current, {"band": 1, "lead_tilings": 1}),
        "b1_post1": (solve_current, {"band": 1, "post_tilings": 1}),
        "b1_lead1_post1": (solve_current, {"band": 1, "lead_tilings": 1, "post_tilings": 1}),
        "b1_lead1_li3": (solve_current, {"band": 1, "lead_tilings": 1, "lead_inner": 3}),
        "cur_forceL1": (solve_current, {"force_L_from": 1}),
        "cur_forceL0": (solve_current, {"force_L_from": 0}),
        "cur_nopost_forceL1": (solve_current, {"force_L_from": 1, "post": False}),
        "cur_nopost_forceL1_li3": (solve_current, {"force_L_from": 1, "post": False, "lead_inner": 3}),
        "cur_b1_nopost_forceL1": (solve_current, {"force_L_from": 1, "post": False, "band": 1}),
        "cur_forceL0_nolead": (solve_current, {"force_L_from": 0, "lead": False}),
        "cur_forceL0_li3": (solve_current, {"force_L_from": 0, "lead_inner": 3}),
        "cur_forceL0_b1": (solve_current, {"force_L_from": 0, "band": 1}),
        "cur_forceL1_li3": (solve_current, {"force_L_from": 1, "lead_inner": 3}),
        "cur_forceL1_x": (solve_current, {"force_L_from": 1, "exact": True}),
        "cur_trig1e-3": (solve_current, {"trigger": 1e-3, "rs_max": 2.0}),
        "cur_b3": (solve_current, {"band": 3}),
        "cur_nopost": (solve_current, {"post": False}),
        "cur_li1": (solve_current, {"lead_inner": 1}),
        "cur_li3": (solve_current, {"lead_inner": 3, "post_inner": 2}),
        "cur_exact": (solve_current, {"exact": True}),
        "gw16": (solve_gamma_w, {"width": 16}),
        "gw16_noso": (solve_gamma_w, {"width": 16, "so": False}),
        "gw32": (solve_gamma_w, {"width": 32}),
        "gw24": (solve_gamma_w, {"width": 24}),
        "gw48": (solve_gamma_w, {"width": 48}),
        "gsA1": (solve_gs, {"stages": [[(0, 16, 1)], [(8, 16, 1)], [(0, 16, 1)], [(8, 16, 1)]], "far_sweeps": 1}),
        "gsA2": (solve_gs, {"stages": [[(0, 16, 2)], [(8, 16, 1)], [(0, 16, 1)], [(8, 16, 1)]], "far_sweeps": 1}),
        "gsAB1": (solve_gs, {"stages": [[(0, 16, 1), (8, 16, 1)], [(0, 16, 1), (8, 16, 1)]], "far_sweeps": 1}),
        "gsAB2": (solve_gs, {"stages": [[(0, 16, 2), (8, 16, 2)], [(0, 16, 1), (8, 16, 1)]], "far_sweeps": 1}),
        "gsABx": (solve_gs, {"stages": [[(0, 16, "exact"), (8, 16, "exact")]], "far_sweeps": 1}),
        "gsAB2all": (solve_gs, {"stages": [[(0, 16, 2), (8, 16, 2)], [(0, 16, 1), (8, 16, 1)]], "far_sweeps": 0}),
        "gs32x": (solve_gs, {"stages": [[(0, 32, "exact"), (16, 32, "exact")]], "far_sweeps": 1}),
        "gs32x1": (solve_gs, {"stages": [[(0, 32, "exact")], [(16, 32, "exact")]], "far_sweeps": 1}),
        "gwf16": (solve_gamma_w, {"width": 16, "far_sweeps": 1}),
        "gwf32": (solve_gamma_w, {"width": 32, "far_sweeps": 1}),
        "gwf48": (solve_gamma_w, {"width": 48, "far_sweeps": 1}),
        "gwf32b1": (solve_gamma_w, {"width": 32, "far_sweeps": 1, "band": 1}),
        "gwf32b3": (solve_gamma_w, {"width": 32, "far_sweeps": 1, "band": 3}),
    }
    global NORM
    if sys.argv[1:] and sys.argv[1].startswith("norm="):
        NORM = sys.argv.pop(1)[5:]
    pick = sys.argv[1:] or list(variants)
    for name in pick:
        fn, v = variants[name]
        A = build_A(d["upper"][0], d["log_pi"][0], mask)
        sig = np.abs(np.diag(A)).max()
        lam, U = np.linalg.eigh(A - sig * np.eye(400))
        U = U[:, np.argsort(lam)]
        tot = 0
        nsw = 0
        print(f"== {name}")
        for e in range(1, E):
            A = build_A(d["upper"][e], d["log_pi"][e], mask)
            sig = np.abs(np.diag(A)).max()
            Ap = A - sig * np.eye(400)
            G, hist, launches = fn(Ap, U, v)
            U = finish(G)
            S = U.T @ A @ U
            res = np.abs(S - np.diag(np.diag(S))).max() / sig
            orth = np.abs(U.T @ U - np.eye(400)).max()
            if e >= 5:
                tot += launches
                nsw += len(hist)
            print(f"  epoch {e:2d}: " + " ".join(f"{h[0]}{h[4]}:{h[1]:.0e}/{h[2]:.0e}/{h[3]:.0e}" for h in hist) + f"  launches {launches}  resid {res:.1e} orth {orth:.1e}")
        print(f"  epochs 5..{E - 1}: launches {tot / (E - 5):.1f} per solve, sweeps {nsw / (E - 5):.2f}")


if __name__ == "__main__":
    main()
