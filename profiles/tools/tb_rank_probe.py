"""Round 5: how many independent buckets does the bench bank have?  Singular values of E[b, i] = exp(t_b lam_i) and of the seven
weightings of the accumulated form stacked beside it, at the JTT-IPW start of the bench bank (numpy, CPU, seconds).
python profiles/tools/tb_rank_probe.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cherryml_amd.estimation._jtt_ipw import jtt_ipw_from_statistics  # noqa: E402

wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
t, C, mask = wl["t"], wl["C"], wl["mask"]
Cs = 0.5 * (C + C.transpose(0, 2, 1))
Q = jtt_ipw_from_statistics(Cs.sum(0), (Cs / t[:, None, None]).sum(0), t, mask, True, 1e-8)
w, v = np.linalg.eig(Q.T)
pi = np.real(v[:, np.argmin(np.abs(w))])
pi /= pi.sum()
d = np.sqrt(pi)
A = d[:, None] * Q / d[None, :]
lam = np.linalg.eigvalsh(0.5 * (A + A.T))
E = np.exp(t[:, None] * lam[None, :])
print("t", t.min(), t.max(), "lam", lam.min(), lam.max())
for name, M in (("E", E), ("E and t^(2k+1) E, k = 0..5", np.concatenate([E] + [t[:, None] ** (2 * k + 1) * E for k in range(6)], 1))):
    s = np.linalg.svd(M, compute_uv=False)
    print(name, {r: float("%.1e" % (s[r] / s[0])) for r in (8, 12, 16, 20, 24, 28, 32)})
