"""Parity WHERE THE HEADLINE NUMBER IS MEASURED: 400 states, B = 129, trajectories of
50-60 epochs -- against outputs of the reference itself (tests/golden/make_golden_s400_full.py: the
reference's `coevolution_end_to_end_with_cherryml_optimizer` on all 32 demo_data families, and its
`train_quantization` / epoch body in float64 on the demo bank and on the dense bank bench.py times).
This is the first place the warm-started hybrid eigensolver is checked against the reference beyond
epoch 3.  Tolerances: loss 1e-12, dL/dQ 1e-10, learned Q 1e-6 relative Frobenius (BASELINE.json).
Also the float32 bank (cb_create(dtype = CB_F32)): the cfg-5 fp64 / fp32 sweep.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


def _mask_of(z):
    return np.unpackbits(z["mask_packed"])[:160000].reshape(400, 400).astype(np.float64)


def _from_support(vals, mask):
    keep = (mask != 0) | np.eye(len(mask), dtype=bool)
    Q = np.zeros(mask.shape)
    Q[keep] = vals
    return Q


def _demo_bank(z):
    C = np.zeros(tuple(z["C_shape"]))
    C[z["C_b"].astype(np.int64), z["C_i"].astype(np.int64), z["C_j"].astype(np.int64)] = z["C_quarters"] * 0.25
    return z["t"], C


def _pi_of(log_pi):
    p = np.exp(log_pi - log_pi.max())
    return p / p.sum()


@pytest.fixture(scope="module")
def demo():
    z = load_golden("coevo_demo_full.npz")
    t, C = _demo_bank(z)
    assert C.sum() == 1057194.0 and np.count_nonzero(C) == 730864
    return z, t, C, _mask_of(z)


@pytest.fixture(scope="module")
def dense():
    import bench
    wl = bench.make_workload("coevo400", 0, np.random.default_rng(0))
    return wl


def _check_bank_is_the_goldens(wl, z):
    """the 165 MB dense bank is regenerated from its seed: make sure it is the one the reference saw"""
    C = wl["C"]
    assert np.allclose(C.sum(), z["C_sum"], rtol=1e-13)
    assert np.allclose(C.reshape(C.shape[0], -1).sum(1), z["C_bucket_sums"], rtol=1e-12)
    assert np.allclose(C[::16, ::37, ::41], z["C_probe"], rtol=1e-11, atol=1e-300)


@pytest.mark.parametrize("dtype,tol_loss,tol_grad", [("f64", 1e-12, 1e-10), ("mixed", 1e-12, 2e-5), ("f32", 2e-6, 2e-3)])
def test_demo_bank_single_evaluation_vs_reference(demo, dtype, tol_loss, tol_grad):
    """(loss, dL/dQ) on the REAL config-3 bank (all 32 families, 43 live buckets, 3.5 % dense) at the
    JTT-IPW initialisation; float32 bank: the stated f32 tolerance."""
    from cherryml_amd import CherryBank
    z, t, C, mask = demo
    Q = _from_support(z["Q_support_f64"], mask)
    with CherryBank(t, C, dtype=dtype) as bank:
        assert int(bank.live_buckets[0]) == 43
        loss, dQ = bank.loss_grad(Q, _pi_of(z["log_pi"]))
    el, eg = abs(loss[0] - float(z["loss_f64"])) / abs(float(z["loss_f64"])), relerr(dQ[0], z["dQ_f64"])
    print(f"demo bank {dtype}: loss rel. err {el:.2e}, dL/dQ rel. Frobenius {eg:.2e}")
    assert el < tol_loss and eg < tol_grad


@pytest.mark.parametrize("dtype,tol_loss,tol_grad", [("f64", 1e-12, 1e-10), ("mixed", 1e-12, 2e-5), ("f32", 2e-6, 2e-3)])
def test_dense_bench_bank_single_evaluation_vs_reference(dense, dtype, tol_loss, tol_grad):
    """the bank bench.py times (B = 129, every bucket populated)"""
    from cherryml_amd import CherryBank
    z = load_golden("coevo_dense_eval.npz")
    _check_bank_is_the_goldens(dense, z)
    Q = _from_support(z["Q_support_f64"], dense["mask"])
    with CherryBank(dense["t"], dense["C"], dtype=dtype) as bank:
        loss, dQ = bank.loss_grad(Q, _pi_of(z["log_pi"]))
    el, eg = abs(loss[0] - float(z["loss_f64"])) / abs(float(z["loss_f64"])), relerr(dQ[0], z["dQ_f64"])
    print(f"dense bank {dtype}: loss rel. err {el:.2e}, dL/dQ rel. Frobenius {eg:.2e}")
    assert el < tol_loss and eg < tol_grad


def _train(t, C, mask, u0, p0, E, dtype="f64"):
    from cherryml_amd import CherryBank
    with CherryBank(t, C, dtype=dtype) as bank:
        return bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)


def test_demo_bank_50_epoch_trajectory_vs_reference(demo):
    """50 epochs of the reference's `train_quantization` (float64) on the real bank: loss curve, Q_1, Q_2,
    Q_best, Q_last.  49 warm-started eigensolves in a row, checked against torch.matrix_exp + autograd."""
    z, t, C, mask = demo
    E = int(z["epochs"])
    r = _train(t, C, mask, z["upper_diag"], z["log_pi"], E)
    assert np.allclose(r["loss"], z["traj_loss_f64"], rtol=1e-9, atol=0)
    for key, got in (("Q_1", r["Q_pow2"][1]), ("Q_2", r["Q_pow2"][2]), ("Q_best", r["Q_best"]), ("Q_last", r["Q_last"])):
        want = _from_support(z[f"traj_{key}_support_f64"], mask)
        e = relerr(got, want)
        print(f"demo bank, {E} epochs: {key} rel. Frobenius {e:.2e}")
        assert e < 1e-6, key


def test_dense_bank_60_epoch_trajectories_vs_reference(dense):
    """8 buckets of the bench bank, 60 epochs: float64 against the reference in float64 (1e-6), and the
    float32 bank against BOTH reference runs (its own float32 arithmetic, and float64) -- the distances
    the cfg-5 sweep reports."""
    z = load_golden("coevo_dense_traj.npz")
    _check_bank_is_the_goldens(dense, z)
    sel, E, mask = z["sel"], int(z["epochs"]), dense["mask"]
    t, C = dense["t"][sel], dense["C"][sel]
    r64 = _train(t, C, mask, z["upper_diag0"], z["log_pi0"], E)
    assert np.allclose(r64["loss"], z["loss_f64"], rtol=1e-9, atol=0)
    for key in ("Q_best", "Q_last"):
        e = relerr(r64[key], _from_support(z[f"{key}_support_f64"], mask))
        print(f"dense sub-bank f64: {key} to the f64 reference {e:.2e}")
        assert e < 1e-6
    r32 = _train(t, C, mask, z["upper_diag0"], z["log_pi0"], E, dtype="f32")
    assert np.allclose(r32["loss"], z["loss_f64"], rtol=5e-6, atol=0)
    d_ref = relerr(_from_support(z["Q_last_support_f32"], mask), _from_support(z["Q_last_support_f64"], mask))
    for key in ("Q_best", "Q_last"):
        e64 = relerr(r32[key], _from_support(z[f"{key}_support_f64"], mask))
        e32 = relerr(r32[key], _from_support(z[f"{key}_support_f32"], mask))
        print(f"dense sub-bank f32 bank: {key} to the f64 reference {e64:.2e}, to the reference's own float32 run "
              f"{e32:.2e} (reference f32 vs f64: {d_ref:.2e})")
    # Q_last follows the trajectory itself (Q_best also depends on which epoch wins a flat argmin)
    assert relerr(r32["Q_last"], _from_support(z["Q_last_support_f64"], mask)) < 1e-3
    assert relerr(r32["Q_last"], _from_support(z["Q_last_support_f32"], mask)) < 1e-3
    # CB_MIXED (P_b, loss, G_b in float64; the two gradient products on the float32 MFMA) anchored to the REFERENCE's runs
    # too, not only to our own f64 bank: it must sit closer to the float64 reference than the reference's own
    # float32 arithmetic does
    rmx = _train(t, C, mask, z["upper_diag0"], z["log_pi0"], E, dtype="mixed")
    assert np.allclose(rmx["loss"], z["loss_f64"], rtol=1e-7, atol=0)
    for key in ("Q_best", "Q_last"):
        e64 = relerr(rmx[key], _from_support(z[f"{key}_support_f64"], mask))
        e32 = relerr(rmx[key], _from_support(z[f"{key}_support_f32"], mask))
        print(f"dense sub-bank mixed bank: {key} to the f64 reference {e64:.2e}, to the reference's own float32 run {e32:.2e}")
    emx = relerr(rmx["Q_last"], _from_support(z["Q_last_support_f64"], mask))
    assert emx < 1e-4 and emx < 10 * d_ref + 1e-6


def test_dense_bank_full_60_epoch_trajectory_vs_reference(dense):
    """THE bench configuration: all 129 buckets, 60 epochs, against the reference in float64."""
    z = load_golden("coevo_dense_traj_full.npz")
    _check_bank_is_the_goldens(dense, z)
    E, mask = int(z["epochs"]), dense["mask"]
    r = _train(dense["t"], dense["C"], mask, z["upper_diag0"], z["log_pi0"], E)
    assert np.allclose(r["loss"], z["loss_f64"], rtol=1e-9, atol=0)
    for key, got in (("Q_1", r["Q_pow2"][1]), ("Q_2", r["Q_pow2"][2]), ("Q_best", r["Q_best"]), ("Q_last", r["Q_last"])):
        e = relerr(got, _from_support(z[f"{key}_support_f64"], mask))
        print(f"dense bank B = 129, {E} epochs: {key} rel. Frobenius {e:.2e}")
        assert e < 1e-6, key


def test_cfg5_fp64_vs_fp32_sweep():
    """BASELINE.json config 5: co-evolution 400 x 400, 10k synthetic families (sum C ~ 1e8, B = 129), the
    same optimisation in the three arithmetic modes.  Reported: rel. Frobenius(Q_mode, Q_f64) along the
    trajectory; asserted: the loss curves agree to float32 accuracy and the learned matrices to 1e-3 (the
    reference's own float32 arithmetic sits 1e-5 .. 1e-3 from its float64 run)."""
    import bench
    rng = np.random.default_rng(5)
    Q, pi, mask = bench.coevolution_truth(rng)
    t, C = bench.reversible_bank(Q, pi, 1.0e8, rng)
    from cherryml_amd import CherryBank, RateMatrix
    from cherryml_amd.estimation import jtt_ipw_from_arrays
    import torch
    init = jtt_ipw_from_arrays(t, C, mask)
    mod = RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(mask),
                     pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
    u0, p0 = mod.upper_diag.detach().numpy().copy(), mod._pi.detach().numpy().copy()
    E = 100
    runs = {}
    for dtype in ("f64", "mixed", "f32"):
        with CherryBank(t, C, dtype=dtype) as bank:
            runs[dtype] = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
    a = runs["f64"]
    assert a["loss"][-1] < a["loss"][0]
    for dtype, tol_loss, tol_q in (("mixed", 1e-8, 1e-4), ("f32", 5e-6, 1e-3)):
        b = runs[dtype]
        assert np.all(np.isfinite(b["loss"]))
        dl = np.max(np.abs(a["loss"] - b["loss"]) / np.abs(a["loss"]))
        sweep = {k: float(f"{relerr(b['Q_pow2'][k], a['Q_pow2'][k]):.1e}") for k in sorted(a["Q_pow2"])}
        print(f"cfg-5 sweep (sum C = 1e8, {E} epochs) {dtype} vs f64: max rel. loss difference {dl:.2e}; rel. Frobenius of Q at "
              f"epochs {sweep}; Q_last {relerr(b['Q_last'], a['Q_last']):.2e}, Q_best {relerr(b['Q_best'], a['Q_best']):.2e}; "
              f"to the generating Q: f64 {relerr(a['Q_best'], Q):.3e}, {dtype} {relerr(b['Q_best'], Q):.3e}")
        assert dl < tol_loss
        assert relerr(b["Q_last"], a["Q_last"]) < tol_q and relerr(b["Q_best"], a["Q_best"]) < tol_q


def test_float32_modes_of_a_single_small_bank_vs_both_reference_runs():
    """cb_create's dtype at S <= 32 (VERDICT r2 "missing 3"): float32 is the reference's OWN arithmetic for the LG bank
    (ratelearner.py:98,107).  A single bank of any size takes CB_F32 / CB_MIXED through the tile kernels of the 400-state
    path (LD = 32 at 20 states); batches of sites stay float64 and refuse.  Reported and asserted here: one evaluation and
    the 100-epoch LG trajectory of the reference (tests/golden/traj_lgbank.npz, B = 129) in the three modes, against the
    reference run in float64 AND as is (float32)."""
    from cherryml_amd import CherryBank
    from oracle import ratelearn_oracle as orc
    g = load_golden("eval_s20_symmask.npz")
    p = np.exp(g["log_pi"] - g["log_pi"].max())
    pi = p / p.sum()
    for dtype, tol_loss, tol_grad in (("f64", 1e-12, 1e-10), ("mixed", 1e-12, 2e-5), ("f32", 2e-6, 2e-3)):
        with CherryBank(g["t"], g["C"], dtype=dtype) as bank:
            loss, dQ = bank.loss_grad(g["Q_f64"], pi, normalize=True)
        assert abs(loss[0] - g["loss_f64"]) < tol_loss * abs(g["loss_f64"]), dtype
        assert relerr(dQ[0], g["dQ_f64"]) < tol_grad, (dtype, relerr(dQ[0], g["dQ_f64"]))
    z = load_golden("traj_lgbank.npz")
    E = 100
    u0, p0 = z["upper_diag0_f64"], z["log_pi0_f64"]
    d_ref = relerr(z["Q_last_f32"], z["Q_last_f64"])
    for dtype, tol_loss, tol_q in (("mixed", 1e-6, 1e-4), ("f32", 1e-5, 2e-3)):
        with CherryBank(z["t"], z["C"], dtype=dtype) as bank:
            r = bank.train_pande_reversible(u0, p0, num_epochs=E, lr=0.1)
            assert bank.last_kernel_form() == 4000      # the C-driven tile-kernel loop, not the float64 small-state split
        dl = np.abs(r["loss"] / z["loss_f64"][:E] - 1).max()
        e64, e32 = relerr(r["Q_last"], z["Q_last_f64"]), relerr(r["Q_last"], z["Q_last_f32"])
        print(f"LG bank, {dtype}: loss curve within {dl:.1e} of the f64 reference; Q_last to the f64 reference {e64:.2e}, to the reference's own float32 run {e32:.2e} "
              f"(reference f32 vs f64: {d_ref:.2e})")
        assert dl < tol_loss and e64 < tol_q and e32 < tol_q + 2 * d_ref
    s = load_golden("siterm_aa.npz")
    with pytest.raises(NotImplementedError):   # batches of sites: float64 only
        CherryBank(np.asarray(s["times"]), s["counts"], dtype="f32")
    with pytest.raises(ValueError):
        CherryBank(g["t"], g["C"], dtype="bf16")


def test_resumed_training_equals_one_call(dense):
    """CB_TRAIN_RESUME: 6 epochs, then 9 resumed epochs (twice: 5 + 4) = 15 epochs of one call, bit for bit -- loss
    curve, parameters, Q_last, Q_best; a resume without a finished call, with another mask, or on a small bank is
    refused.  (bench.py times its K epochs as the continuation of its W warm-up epochs through this.)"""
    from cherryml_amd import CherryBank
    z = load_golden("coevo_dense_traj.npz")
    sel, mask = z["sel"], dense["mask"]
    t, C = dense["t"][sel], dense["C"][sel]
    u0, p0 = z["upper_diag0"], z["log_pi0"]
    with CherryBank(t, C) as bank:
        whole = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=15, lr=0.1)
    with CherryBank(t, C) as bank:   # (a fresh handle: the same cold first eigensolve as `whole`)
        with pytest.raises(ValueError):
            bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=3, lr=0.1, resume=True)
        a = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=6, lr=0.1)
        with pytest.raises(ValueError):
            bank.train_pande_reversible(u0, p0, mask=None, num_epochs=3, lr=0.1, resume=True)
        with pytest.raises(ValueError):   # another learning rate, other mask CONTENTS, another normalisation
            bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=3, lr=0.05, resume=True)
        other = mask.copy()
        other[0, 1] = other[1, 0] = 1.0 - other[0, 1]
        with pytest.raises(ValueError):
            bank.train_pande_reversible(u0, p0, mask=other, num_epochs=3, lr=0.1, resume=True)
        with pytest.raises(ValueError):
            bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=3, lr=0.1, resume=True, normalize=False)
        b = bank.train_pande_reversible(u0 * 0, p0 * 0, mask=mask, num_epochs=5, lr=0.1, resume=True)   # (inputs ignored)
        c = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=4, lr=0.1, resume=True)
    assert np.array_equal(np.concatenate([a["loss"], b["loss"], c["loss"]]), whole["loss"])
    for key in ("Q_last", "Q_best", "upper_diag", "log_pi"):
        assert np.array_equal(c[key], whole[key]), key
    assert c["Q_pow2"] == {}
    g = load_golden("eval_s20_symmask.npz")
    with CherryBank(g["t"], g["C"]) as small:
        nup = 20 * 19 // 2
        small.train_pande_reversible(np.zeros(nup), np.zeros(20), num_epochs=2)
        with pytest.raises(NotImplementedError):
            small.train_pande_reversible(np.zeros(nup), np.zeros(20), num_epochs=2, resume=True)


def test_planned_eigensolves_match_the_host_driven_solver_and_survive_short_plans(dense, monkeypatch):
    """Round 4: every warm eigensolve of the C-driven trainer is a device-controlled PLAN (csrc/eigh_planned.hip.h: the sweep
    decisions are taken on the device, the host only reads a record while the epoch's bank kernels are queued).  The same 25
    epochs (a) planned, (b) with the host-driven solver of rounds 1-3 (CB_EIGH_HOST=1), (c) planned with every plan cut down
    to ONE sweep (CB_EIGH_SHORT_PLAN=1: every solve runs out of plan, is continued from its current state, and the epoch's
    kernels are enqueued again), (d) the same with that sweep limited to the second-order polynomial (= 2: a damped rotation): same loss curves to 1e-11, same matrices to 1e-9; the counters say which path ran."""
    from cherryml_amd import CherryBank
    z = load_golden("coevo_dense_traj.npz")
    sel, mask = z["sel"], dense["mask"]
    t, C = dense["t"][sel], dense["C"][sel]
    u0, p0 = z["upper_diag0"], z["log_pi0"]
    E = 25
    runs = {}
    for name, env in (("planned", {}), ("host", {"CB_EIGH_HOST": "1"}), ("short", {"CB_EIGH_SHORT_PLAN": "1"}),
                      ("damped", {"CB_EIGH_SHORT_PLAN": "2"})):
        for k in ("CB_EIGH_HOST", "CB_EIGH_SHORT_PLAN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        with CherryBank(t, C) as bank:
            r = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
            r["counters"] = bank.eigh_counters()
        runs[name] = r
    a, b, c, d = runs["planned"], runs["host"], runs["short"], runs["damped"]
    assert a["counters"]["planned_solves"] == E - 3       # every solve from the fourth epoch on
    assert a["counters"]["stalls"] <= 2
    assert b["counters"]["planned_solves"] == 0
    assert c["counters"]["planned_solves"] == E - 3 and c["counters"]["stalls"] >= E - 3
    # (d) the one sweep of every plan can only evaluate the second-order polynomial: it applies exp(alpha X) with alpha << 1 --
    # a DAMPED, still exactly orthogonal rotation (a slot that under-provides is never wrong, only slower) -- and stalls
    assert d["counters"]["stalls"] >= E - 3
    for other in (b, c, d):
        assert np.all(np.isfinite(other["loss"]))
        assert np.allclose(a["loss"], other["loss"], rtol=1e-11, atol=0)
        for key in ("Q_last", "Q_best"):
            assert relerr(a[key], other[key]) < 1e-9, key


@pytest.mark.parametrize("S", [64, 100, 200])
def test_planned_eigensolves_at_other_sizes(S, monkeypatch):
    """The device-controlled plans at other matrix sizes (LD = 64, 112, 208: 4, 7 and 13 tiles of 16; 8, 14 and 26 column
    blocks, the smallest size that takes the plans at all): a random reversible model's expected counts, 20 epochs planned
    and with the host-driven solver -- same loss curve (1e-11), same matrices (1e-9); and against the float64 oracle
    (torch.matrix_exp + autograd + Adam) over the first 6 epochs."""
    from cherryml_amd import CherryBank
    from oracle import ratelearn_oracle as orc
    rng = np.random.default_rng(S)
    B = 9
    t = np.geomspace(0.02, 2.0, B)
    R = rng.gamma(2.0, 0.5, size=(S, S))
    R = 0.5 * (R + R.T)
    pi = rng.dirichlet(np.full(S, 5.0))
    Q = R * pi[None, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(1))
    Q /= -(pi * np.diag(Q)).sum()
    w, V = np.linalg.eigh(np.sqrt(pi)[:, None] * Q / np.sqrt(pi)[None, :])
    C = np.stack([pi[:, None] * ((V * np.exp(tb * w)) @ V.T) * np.sqrt(pi)[None, :] / np.sqrt(pi)[:, None] for tb in t]) * 1e5
    C = 0.5 * (C + C.transpose(0, 2, 1))
    mask = np.ones((S, S))
    u0 = rng.normal(0.0, 0.3, S * (S - 1) // 2)
    p0 = rng.normal(0.0, 0.2, S)
    runs = {}
    for name, env in (("planned", None), ("host", "1")):
        if env is None:
            monkeypatch.delenv("CB_EIGH_HOST", raising=False)
        else:
            monkeypatch.setenv("CB_EIGH_HOST", env)
        with CherryBank(t, C) as bank:
            runs[name] = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=20, lr=0.1)
            runs[name]["counters"] = bank.eigh_counters()
    a, b = runs["planned"], runs["host"]
    assert a["counters"]["planned_solves"] == 17 and b["counters"]["planned_solves"] == 0
    assert np.all(np.isfinite(a["loss"])) and np.allclose(a["loss"], b["loss"], rtol=1e-11, atol=0)
    assert relerr(a["Q_last"], b["Q_last"]) < 1e-9 and relerr(a["Q_best"], b["Q_best"]) < 1e-9
    o = orc.train(t, C, mask=mask, upper_diag=u0, log_pi=p0, num_epochs=6, lr=0.1)
    assert np.allclose(a["loss"][:6], o["loss"], rtol=1e-10, atol=0)


def test_two_queue_bank_option_gives_the_same_bits(dense, tmp_path):
    """CB_BANK_STREAMS=2 (read once per process, hence the subprocess): the buckets' K1 -> K2 -> K3 chains on two
    queues -- the trajectory is bit-identical to the single-queue one."""
    import os
    import subprocess
    import sys
    z = load_golden("coevo_dense_traj.npz")
    sel, mask = z["sel"], dense["mask"]
    np.savez(tmp_path / "in.npz", t=dense["t"][sel], C=dense["C"][sel], mask=mask, u0=z["upper_diag0"], p0=z["log_pi0"])
    code = ("import sys, numpy as np; from cherryml_amd import CherryBank; z = np.load(sys.argv[1]);\n"
            "b = CherryBank(z['t'], z['C']); r = b.train_pande_reversible(z['u0'], z['p0'], mask=z['mask'], num_epochs=12, lr=0.1);\n"
            "np.savez(sys.argv[2], loss=r['loss'], Q=r['Q_last']); b.close()")
    outs = []
    for n in ("1", "2"):
        env = dict(os.environ, CB_BANK_STREAMS=n, PYTHONPATH=os.getcwd() + os.pathsep + os.environ.get("PYTHONPATH", ""))
        out = tmp_path / f"out{n}.npz"
        subprocess.run([sys.executable, "-c", code, str(tmp_path / "in.npz"), str(out)], check=True, env=env, timeout=600)
        outs.append(np.load(out))
    assert np.array_equal(outs[0]["loss"], outs[1]["loss"]) and np.array_equal(outs[0]["Q"], outs[1]["Q"])
    assert np.allclose(outs[0]["loss"], z["loss_f64"][:12], rtol=1e-9, atol=0)


def _random_bank(S, B, seed, sym=True):
    rng = np.random.default_rng(seed)
    t = np.geomspace(0.02, 2.0, B)
    R = rng.gamma(2.0, 0.5, size=(S, S))
    R = 0.5 * (R + R.T)
    pi = rng.dirichlet(np.full(S, 5.0))
    Q = R * pi[None, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(1))
    Q /= -(pi * np.diag(Q)).sum()
    w, V = np.linalg.eigh(np.sqrt(pi)[:, None] * Q / np.sqrt(pi)[None, :])
    C = np.stack([pi[:, None] * ((V * np.exp(tb * w)) @ V.T) * np.sqrt(pi)[None, :] / np.sqrt(pi)[:, None] for tb in t]) * 1e5
    if sym:
        C = 0.5 * (C + C.transpose(0, 2, 1))
    else:
        C = C * rng.uniform(0.5, 1.5, size=C.shape)     # C_b != C_b^T: K3 then runs all its tiles
    return t, C, Q, pi


@pytest.mark.parametrize("dtype", ["f64", "mixed", "f32"])
def test_fused_bank_launch_gives_the_bits_of_the_three_launches_on_the_bench_bank(dense, dtype, monkeypatch):
    """K1 -> K2 -> K3 as ONE persistent launch of ticket-drawing workgroups (k123_bank, the default) against the three
    separate launches (CB_BANK_UNFUSED=1) on the bank bench.py times: every tile is computed by the same instructions in the
    same order, only WHEN and WHERE it runs differs, so loss, dL/dQ and 12 epochs of training agree bit for bit.  (This is
    the test that would see a tile read before its inputs were visible: the fused launch publishes Gt_b / T_b to the other
    XCDs with write-through stores and per-bucket counters instead of a launch boundary.)"""
    from cherryml_amd import CherryBank
    from cherryml_amd.estimation._jtt_ipw import jtt_ipw_from_arrays
    import cherryml_amd
    import torch
    wl = dense
    S = 400
    init = jtt_ipw_from_arrays(wl["t"], wl["C"], wl["mask"])
    mod = cherryml_amd.RateMatrix(num_states=S, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                                  pi=torch.ones(S, dtype=torch.float64) / S, pi_requires_grad=True, initialization=init)
    u0, p0 = mod.upper_diag.detach().numpy().copy(), mod._pi.detach().numpy().copy()
    Q0, pi0 = mod().detach().numpy(), mod.stationary().detach().numpy()
    out = {}
    for name, env in (("fused", "CB_BANK_FUSED"), ("separate", "CB_BANK_UNFUSED")):
        for k in ("CB_BANK_FUSED", "CB_BANK_UNFUSED"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv(env, "1")
        with CherryBank(wl["t"], wl["C"], dtype=dtype) as bank:
            loss, dQ = bank.loss_grad(Q0, pi0)
            loss2, dQ2 = bank.loss_grad(Q0, pi0)          # the queues are reset by every evaluation (the second one starts
            assert np.allclose(loss, loss2, rtol=1e-6) and relerr(dQ, dQ2) < 1e-4      # its eigensolve warm: last bits differ)
            bank.profile(True)
            r = bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=12, lr=0.1)
            tm = bank.timing_means()
            bank.profile(False)
        out[name] = (loss, dQ, r, tm)
    (la, da, ra, ta), (lb, db, rb, tb) = out["fused"], out["separate"]
    assert np.all(np.isfinite(la)) and np.array_equal(la, lb)
    assert np.array_equal(da, db)
    assert np.array_equal(ra["loss"], rb["loss"])
    assert np.array_equal(ra["Q_last"], rb["Q_last"]) and np.array_equal(ra["Q_best"], rb["Q_best"])
    # the phase report says which form ran: one span for the fused launch, three for the separate ones
    assert ta["k1"] > 0 and ta["k2"] == 0 and ta["k3"] == 0
    assert tb["k1"] > 0 and tb["k2"] > 0 and tb["k3"] > 0


@pytest.mark.parametrize("kg", [1, 2])
@pytest.mark.parametrize("dtype", ["f64", "mixed", "f32"])
@pytest.mark.parametrize("S,B,sym", [(100, 3, True), (100, 9, False), (64, 1, True), (200, 20, True), (400, 7, False),
                                     (400, 17, True), (400, 43, True)])
def test_fused_bank_launch_on_small_and_ragged_banks(S, B, sym, dtype, kg, monkeypatch):
    """Fewer buckets than ticket queues (empty queues, every ticket drawn from a neighbour's queue), one bucket, fewer tickets
    than resident workgroups, LD = 112 / 64 / 208 (2 / 1 / 3 tiles a side), asymmetric counts (K3 runs all its tiles), the
    sizes of one rank's share of an 8-rank job (17 buckets) and of the reference's real bank (43): the fused launch
    (CB_BANK_FUSED=1: by itself large_eval picks the three launches below 64 live buckets) against the three launches bit for
    bit, in every arithmetic (with float32 outputs a tile edge falls inside a 128-byte line, so two workgroups on
    different XCDs write through parts of the same line) and both tile forms (four / eight waves per tile, CB_BANK_KG); the
    float64 result also against the oracle."""
    from cherryml_amd import CherryBank
    from oracle import ratelearn_oracle as orc
    t, C, Q, pi = _random_bank(S, B, 1000 * S + B, sym)
    monkeypatch.setenv("CB_BANK_KG", str(kg))
    out = {}
    for name, env in (("fused", "CB_BANK_FUSED"), ("separate", "CB_BANK_UNFUSED")):
        for k in ("CB_BANK_FUSED", "CB_BANK_UNFUSED"):
            monkeypatch.delenv(k, raising=False)
        monkeypatch.setenv(env, "1")
        with CherryBank(t, C, dtype=dtype) as bank:
            out[name] = bank.loss_grad(Q * 0.9, pi)
    (la, da), (lb, db) = out["fused"], out["separate"]
    assert np.all(np.isfinite(la)) and np.array_equal(la, lb) and np.array_equal(da, db)
    if dtype != "f64" or S * B > 4000:   # (the oracle: float64, sizes it finishes in seconds)
        return
    import torch
    Qt = torch.tensor(Q * 0.9, dtype=torch.float64, requires_grad=True)
    lo = orc.bank_loss(Qt, torch.tensor(t, dtype=torch.float64), torch.tensor(C, dtype=torch.float64))
    lo.backward()
    lo = float(lo.detach())
    assert abs(float(la[0]) - lo) <= 1e-12 * abs(lo)
    assert relerr(da[0], Qt.grad.numpy()) < 1e-10


def _product_model_bank(S1, B, seed):
    """Q (x) I + I (x) Q on S1^2 states: every eigenvalue lam_a + lam_b with a != b is EXACTLY degenerate (a, b) / (b, a)"""
    t, _, Q1, pi1 = _random_bank(S1, B, seed)
    I = np.eye(S1)
    Q = np.kron(Q1, I) + np.kron(I, Q1)
    pi = np.kron(pi1, pi1)
    d = np.sqrt(pi)
    w, V = np.linalg.eigh(d[:, None] * Q / d[None, :])
    C = np.stack([pi[:, None] * ((V * np.exp(tb * w)) @ V.T) * d[None, :] / d[:, None] for tb in t]) * 1e5
    return t, 0.5 * (C + C.transpose(0, 2, 1)), Q, pi


@pytest.mark.parametrize("case", ["bench", "demo", "ragged", "product", "long_branches"])
def test_bucket_sum_before_the_last_product_matches_the_per_bucket_third_product(case, dense, monkeypatch):  # noqa: C901
    """Symmetric counts (round 5): M = sum_b (U^T G_b U) o Phi_b is evaluated as differences / series of the ONE-SIDED sums
    sum_b diag(c_b) U^T G_b U = (sum_b T_b diag(c_b))^T U -- the buckets are summed before the product, K3 disappears
    (csrc/large_bank.hip.h, ky_reduce_loss / kphi_combine).  Against the per-bucket third product (test hook CB_BANK_K3=1):
    the loss is the same number (K1 is untouched) and dL/dQ agrees to 1e-11 (the reference's real bank: 5e-11 -- its G~_b
    hold entries of 1e9 where a counted double substitution meets P ~ 1e-11, and the difference Le - Le^T cancels three digits
    more than on the synthetic banks; the bar against the reference itself is 1e-10, where the per-bucket form measured
    2.9e-11) -- on the bench bank, the reference's real bank,
    a small ragged bank, a product model (exactly degenerate eigenvalue pairs: the series branch at dlam = 0) and a bank
    with a branch length of 500 (delta = 0.2 / t_max = 4e-4: nearly every pair takes the quotient)."""
    from cherryml_amd import CherryBank
    if case == "bench":
        import cherryml_amd
        import torch
        from cherryml_amd.estimation._jtt_ipw import jtt_ipw_from_arrays
        t, C = dense["t"], dense["C"]
        init = jtt_ipw_from_arrays(t, C, dense["mask"])
        mod = cherryml_amd.RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(dense["mask"]),
                                      pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
        Q, pi = mod().detach().numpy(), mod.stationary().detach().numpy()
    elif case == "demo":
        z = load_golden("coevo_demo_full.npz")
        t, C = _demo_bank(z)
        Q, pi = _from_support(z["Q_support_f64"], _mask_of(z)), _pi_of(z["log_pi"])
    elif case == "ragged":
        t, C, Q, pi = _random_bank(100, 9, 77)
    elif case == "product":
        t, C, Q, pi = _product_model_bank(10, 12, 3)
    else:
        t, C, Q, pi = _random_bank(64, 7, 11)
        t = t.copy()
        t[-1] = 500.0
    out = {}
    for name, hook in (("summed", "0"), ("per_bucket", "1")):   # (forced both ways: by itself large_eval sums first from 24 buckets on)
        monkeypatch.setenv("CB_BANK_K3", hook)
        with CherryBank(t, C) as bank:
            out[name] = bank.loss_grad(Q * 0.95, pi)
            form = bank.last_bank_form()
        assert form["bucket_sum_first"] == (hook == "0"), form
    (la, da), (lb, db) = out["summed"], out["per_bucket"]
    assert np.all(np.isfinite(da)) and la[0] == lb[0]
    print(f"bucket sum first vs per-bucket third product ({case}): dL/dQ rel. Frobenius {relerr(da[0], db[0]):.2e}")
    assert relerr(da[0], db[0]) < (5e-11 if case == "demo" else 1e-11), relerr(da[0], db[0])


@pytest.mark.parametrize("dtype", ["f64", "mixed", "f32"])
def test_reserved_tickets_nobody_claims_are_run_by_the_workgroups_that_wait_for_them(dense, dtype, monkeypatch):
    """The fused bank launch reserves the first K1 tickets of every queue for its workgroups (one uncontended claim instead of
    128 draws on one counter); a reserved ticket whose workgroup is not resident -- another launch holds its slot -- must not be
    waited for forever.  Test hook CB_BANK_TEST_NO_CLAIM=1: no workgroup takes its reserved ticket; every one of them (128 per
    queue on the bench bank) is then found and run by a workgroup that waits for its bucket.  Same bits as the three separate
    launches, on the bench bank and on a small one."""
    from cherryml_amd import CherryBank
    for t, C, Q, pi in ((dense["t"], dense["C"], None, None), _random_bank(100, 9, 77)):
        if Q is None:
            import cherryml_amd
            import torch
            from cherryml_amd.estimation._jtt_ipw import jtt_ipw_from_arrays
            init = jtt_ipw_from_arrays(dense["t"], dense["C"], dense["mask"])
            mod = cherryml_amd.RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(dense["mask"]),
                                          pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True, initialization=init)
            Q, pi = mod().detach().numpy(), mod.stationary().detach().numpy()
        out = {}
        for name, env in (("helped", {"CB_BANK_TEST_NO_CLAIM": "1", "CB_BANK_FUSED": "1"}), ("separate", {"CB_BANK_UNFUSED": "1"})):
            for k in ("CB_BANK_TEST_NO_CLAIM", "CB_BANK_UNFUSED", "CB_BANK_FUSED"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            with CherryBank(t, C, dtype=dtype) as bank:
                out[name] = bank.loss_grad(Q, pi)
        (la, da), (lb, db) = out["helped"], out["separate"]
        assert np.all(np.isfinite(la)) and np.array_equal(la, lb) and np.array_equal(da, db)
