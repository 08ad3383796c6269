"""BASELINE.json config 4 (SiteRM, B = 129, N = 20, L = 5000) pinned where it is measured: the reference's own
`quantized_transitions_mle_vectorized_over_sites` (float64) was run on 32 sites of the bench tensor and on 24 sites of a
21-state problem with non-symmetric counts (tests/golden/make_golden_siterm_cfg4.py -> siterm_cfg4.npz); here those sites
sit INSIDE the full batches, so the kernels compared are the ones `bench.py --workload siterm` launches --
`sp_bank<5, true, true>` (three workgroups per CU, symmetric counts) -- and `sp_bank<6, false, *>`.
Tolerances: loss curves 1e-8 relative, learned matrices 1e-6 relative Frobenius (BASELINE.json's bar)."""
import os
import sys

import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def test_cfg4_sites_inside_the_5000_site_batch_match_the_reference():
    import bench
    from cherryml_amd import CherryBank
    from cherryml_amd._siterm._vectorized import _invert
    z = load_golden("siterm_cfg4.npz")
    sel = z["sites20"]
    wl = bench.make_workload("siterm", 5000, np.random.default_rng(0))
    # the tensor regenerated on this machine is the one the reference saw (entries down to 1e-19 keep ~1e-13 relative)
    assert np.allclose(wl["C"][:2], z["banks20"], rtol=1e-10, atol=1e-300)
    assert np.allclose(wl["t"][sel], z["times20"], rtol=1e-14, atol=0) and np.array_equal(wl["init"][sel], z["init20"])
    wl["C"][sel] = z["banks20"][sel % 2]          # exactly the reference's inputs at the compared sites
    wl["t"][sel] = z["times20"]
    E = z["lpeps20"].shape[0]
    th0, Th0 = _invert(wl["init"])
    with CherryBank(wl["t"], wl["C"]) as bank:
        r = bank.train_siterm(th0, Th0, E, lr=0.1)
        assert bank.last_kernel_form() == 1511      # sp_bank<TS = 5, symmetric, three workgroups per CU>: the bench's form
    got = r["loss_per_epoch_per_site"][:, sel]
    assert np.allclose(got, z["lpeps20"], rtol=1e-8, atol=0), np.abs(got / z["lpeps20"] - 1).max()
    worst = max(relerr(r["res"][l], z["res20"][k]) for k, l in enumerate(sel))
    assert worst < 1e-6, worst
    # the mirrored function on the 32 sites alone (two workgroups per CU: sp_bank<5, true, false>) agrees too
    from cherryml_amd import quantized_transitions_mle_vectorized_over_sites as qvec
    small = qvec(wl["C"][sel], wl["t"][sel], num_epochs=E, initialization=z["init20"], device="cuda")
    assert np.allclose(small["loss_per_epoch_per_site"], z["lpeps20"], rtol=1e-8, atol=0)
    assert max(relerr(small["res"][k], z["res20"][k]) for k in range(len(sel))) < 1e-6


def test_cfg4_21_states_nonsymmetric_counts_inside_a_large_batch_match_the_reference():
    from make_golden_siterm_cfg4 import nonsymmetric_21_state_problem
    from cherryml_amd import CherryBank
    from cherryml_amd._siterm._vectorized import _invert
    z = load_golden("siterm_cfg4.npz")
    T, C, Q0 = nonsymmetric_21_state_problem()
    assert np.allclose(C.sum(axis=(2, 3)), z["counts21_bucket_sums"], rtol=1e-11, atol=0)
    assert np.allclose(T, z["times21"], rtol=1e-14, atol=0) and np.allclose(Q0, z["init21"], rtol=1e-12, atol=1e-300)
    assert np.abs(C - C.transpose(0, 1, 3, 2)).max() > 1e-3       # really not symmetric
    n, reps = len(T), 60
    where = 7 + np.arange(n) * reps                               # the golden sites scattered through 1440 sites
    order = np.arange(n * reps) % n
    order[where] = np.arange(n)
    E = z["lpeps21"].shape[0]
    th0, Th0 = _invert(z["init21"][order])
    with CherryBank(T[order], C[order]) as bank:
        r = bank.train_siterm(th0, Th0, E, lr=0.1)
        form = bank.last_kernel_form()
    assert form // 100 == 16 and (form // 10) % 10 == 0, form     # sp_bank<TS = 6, not symmetric, *>
    assert np.allclose(r["loss_per_epoch_per_site"][:, where], z["lpeps21"], rtol=1e-8, atol=0)
    worst = max(relerr(r["res"][l], z["res21"][k]) for k, l in enumerate(where))
    assert worst < 1e-6, worst
