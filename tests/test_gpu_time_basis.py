"""The 400-state bank in a TIME BASIS (csrc/tbasis.hip.h): products on a few skeleton buckets, one elementwise kernel over all
buckets in between.  Parity of that form with the reference is asserted where every other form's is (tests/test_gpu_s400_full.py,
test_gpu_demo_e2e.py run through it by default: the bench bank has 129 live buckets); HERE it is compared with the per-bucket
products of the same library (cb_create's CB_PER_BUCKET_PRODUCTS) on the bench bank and on ragged shapes, its maintenance is
exercised (a basis that is outgrown at once: the device notices, the epoch is repeated with per-bucket products, a new basis is
built), and the shapes it does not serve are checked to fall back.  Needs an MI355X."""
import numpy as np
import pytest

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dense():
    import bench
    return bench.make_workload("coevo400", 0, np.random.default_rng(0))


def _reversible(S, rng, scale=1.0):
    pi = rng.dirichlet(np.full(S, 5.0))
    R = rng.gamma(1.0, 1.0, (S, S))
    R = (R + R.T) * 0.5
    Q = R * pi[None, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(1))
    return Q * (scale / -(pi * np.diag(Q)).sum()), pi


def _sym_counts(S, B, rng, density=1.0):
    C = rng.poisson(3.0, (B, S, S)).astype(np.float64) * (rng.random((B, S, S)) < density)
    return C + C.transpose(0, 2, 1)


def test_bench_bank_against_the_per_bucket_products(dense):
    """one evaluation at the JTT-IPW start of the bank bench.py times: the time basis against three products per bucket"""
    from cherryml_amd import CherryBank
    z = load_golden("coevo_dense_eval.npz")
    t, C = dense["t"], dense["C"]
    keep = (dense["mask"] != 0) | np.eye(400, dtype=bool)
    Q = np.zeros((400, 400))
    Q[keep] = z["Q_support_f64"]
    p = np.exp(z["log_pi"] - z["log_pi"].max())
    pi = p / p.sum()
    with CherryBank(t, C) as a, CherryBank(t, C, per_bucket_products=True) as b:
        la, ga = a.loss_grad(Q, pi)
        lb, gb = b.loss_grad(Q, pi)
        fa, fb, info = a.last_bank_form(), b.last_bank_form(), a.time_basis_info()
    assert fa["time_basis"] and not fa["fused"] and not fa["bucket_sum_first"] and not fb["time_basis"]
    assert 8 <= info["forward_skeleton"] <= 24 and 16 <= info["gradient_skeleton"] <= 40 and info["builds"] == 1
    assert info["forward_skeleton"] + info["direct"] < 64 and info["rho_max"] > 0
    el, eg = abs(la[0] - lb[0]) / abs(lb[0]), relerr(ga[0], gb[0])
    print(f"time basis {info}: loss {el:.2e}, dL/dQ {eg:.2e} from the per-bucket products")
    assert el < 1e-13 and eg < 2e-11
    # ... and from the reference itself (the figure test_gpu_s400_full.py asserts for whatever form is the default)
    assert abs(la[0] - float(z["loss_f64"])) < 1e-12 * abs(float(z["loss_f64"])) and relerr(ga[0], z["dQ_f64"]) < 1e-10


@pytest.mark.parametrize("S,B,density,scale", [(48, 70, 1.0, 1.0), (50, 129, 0.3, 2.5), (96, 64, 1.0, 0.4), (33, 100, 0.05, 1.0)])
def test_ragged_banks_against_the_per_bucket_products(S, B, density, scale):
    """other state counts (LD = 48, 64, 96: one tile and several, S not a multiple of 16), bucket counts that leave the last
    step of 16 ragged, sparse counts (most pairs contribute nothing), faster and slower matrices (more or fewer long-branch buckets)"""
    from cherryml_amd import CherryBank
    rng = np.random.default_rng(S * 1000 + B)
    Q, pi = _reversible(S, rng, scale)
    t = 0.03 * 1.1 ** (np.arange(B) - B // 2)
    C = _sym_counts(S, B, rng, density)
    with CherryBank(t, C) as a, CherryBank(t, C, per_bucket_products=True) as b:
        la, ga = a.loss_grad(Q, pi)
        lb, gb = b.loss_grad(Q, pi)
        assert a.last_bank_form()["time_basis"] and not b.last_bank_form()["time_basis"]
        info = a.time_basis_info()
    el, eg = abs(la[0] - lb[0]) / abs(lb[0]), relerr(ga[0], gb[0])
    print(f"S {S} B {B} density {density}: {info}; loss {el:.2e}, dL/dQ {eg:.2e}")
    assert el < 1e-13 and eg < 1e-11


def test_two_hundred_live_buckets_need_more_than_64_kb_of_lds_on_every_handle():
    """B = 200 live buckets: tb_ew's interpolation matrices take more than the 64 KB of dynamic LDS a kernel gets without
    hipFuncSetAttribute.  The limit is raised when a basis is INSTALLED, per handle (round 5 kept one process-wide
    word, so a second handle -- on another GPU -- never raised it there): two handles in a row, and a second basis on the first
    (another spectral range, other ranks), all launch."""
    from cherryml_amd import CherryBank
    from cherryml_amd import _lib
    rng = np.random.default_rng(200)
    S, B = 48, 200
    Q, pi = _reversible(S, rng)
    t = 0.03 * 1.07 ** (np.arange(B) - B // 2)
    C = _sym_counts(S, B, rng)
    with CherryBank(t, C, per_bucket_products=True) as b:
        lb, gb = b.loss_grad(Q, pi)
    outs = []
    for _ in range(2):
        with CherryBank(t, C) as a:
            la, ga = a.loss_grad(Q, pi)
            assert a.last_bank_form()["time_basis"]
            info = a.time_basis_info()
            l2, g2 = a.loss_grad(3.0 * Q, pi)       # out of the first basis' range: a second one, larger ranks
            info2, form2 = a.time_basis_info(), a.last_bank_form()
            print(f"B {B}: {info} -> {info2} {form2}")
            assert form2["time_basis"] and info2["builds"] == 2
        outs.append((la[0], ga[0]))
    ns, ng = info["forward_skeleton"], info["gradient_skeleton"]
    lds = ((B + 15) // 16 * 16) * ((16 if ns <= 16 else 24) + 1 + (32 if ng <= 32 else 48) + 2 + 1) * 8
    print(f"B {B}: {info} -> {info2}; tb_ew LDS >= {lds} bytes")
    assert lds > 65536
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert abs(outs[0][0] - lb[0]) < 1e-13 * abs(lb[0]) and relerr(outs[0][1], gb[0]) < 1e-11
    with CherryBank(t, C, per_bucket_products=True) as b:
        l3, g3 = b.loss_grad(3.0 * Q, pi)
    assert abs(l2[0] - l3[0]) < 1e-13 * abs(l3[0]) and relerr(g2[0], g3[0]) < 1e-10
    assert _lib.load().cb_device_count() >= 1


def test_shapes_the_time_basis_does_not_serve_keep_the_per_bucket_forms(monkeypatch):
    from cherryml_amd import CherryBank
    rng = np.random.default_rng(5)
    S, B = 48, 70
    Q, pi = _reversible(S, rng)
    t = 0.03 * 1.1 ** (np.arange(B) - B // 2)
    C = _sym_counts(S, B, rng)
    with CherryBank(t, C) as ref:
        l0, g0 = ref.loss_grad(Q, pi)
    perm = rng.permutation(B)                     # the same bank with its buckets in another order: not an ascending grid
    with CherryBank(t[perm], C[perm]) as a:
        l1, g1 = a.loss_grad(Q, pi)
        assert not a.last_bank_form()["time_basis"]
    assert abs(l1[0] - l0[0]) < 1e-13 * abs(l0[0]) and relerr(g1[0], g0[0]) < 1e-11
    Ca = C.copy()
    Ca[3, 1, 2] += 1.0                            # asymmetric counts
    with CherryBank(t, Ca) as a:
        a.loss_grad(Q, pi)
        assert not a.last_bank_form()["time_basis"]
    with CherryBank(t[:26], C[:26]) as a:         # a short bank: by itself below the threshold, with the hook on
        a.loss_grad(Q, pi)
        assert not a.last_bank_form()["time_basis"]
        monkeypatch.setenv("CB_BANK_TB", "1")
        l2, g2 = a.loss_grad(Q, pi)
        assert a.last_bank_form()["time_basis"]
        monkeypatch.setenv("CB_BANK_TB", "0")
        l3, g3 = a.loss_grad(Q, pi)
        assert not a.last_bank_form()["time_basis"]
    assert abs(l2[0] - l3[0]) < 1e-13 * abs(l3[0]) and relerr(g2[0], g3[0]) < 1e-11
    with CherryBank(t, C, dtype="f32") as a:      # float32 P_b keeps its per-bucket kernels
        a.loss_grad(Q, pi)
        assert not a.last_bank_form()["time_basis"]


def test_a_basis_that_is_outgrown_is_noticed_on_the_device_and_the_epoch_repeated(dense, monkeypatch):
    """CB_TB_TEST_GROWTH=1.0005 (test hook): every basis is built for the matrix at hand with no room to grow, and the host does
    not replace it in time.  The first optimiser steps raise max |Q_ii|: lge_norms finds 2 sigma outside the range, the bank, the
    reduction and K4 return at once, the trainer repeats the evaluation with per-bucket products on the finished decomposition
    and builds a new basis.  The optimisation must come out as without the hook (to rounding: other bases, other sums)."""
    from cherryml_amd import CherryBank
    z = load_golden("coevo_dense_traj.npz")
    t, C, mask = dense["t"], dense["C"], dense["mask"]
    u0, p0 = z["upper_diag0"], z["log_pi0"]
    E = 14
    with CherryBank(t, C) as bank:
        ref = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
        info0 = bank.time_basis_info()
    assert info0["repeated_epochs"] == 0 and info0["builds"] >= 1
    monkeypatch.setenv("CB_TB_TEST_GROWTH", "1.0005")
    with CherryBank(t, C) as bank:
        out = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
        info = bank.time_basis_info()
    print(f"without the hook {info0}; with it {info}")
    assert info["repeated_epochs"] >= 2 and info["builds"] >= info["repeated_epochs"]
    assert np.all(np.isfinite(out["loss"]))
    assert np.abs(out["loss"] - ref["loss"]).max() < 1e-11 * np.abs(ref["loss"]).max()
    assert relerr(out["Q_last"], ref["Q_last"]) < 1e-8


def test_an_optimisation_builds_its_own_basis_whatever_ran_on_the_handle_before(dense):
    """bitwise: a fresh handle, and a handle that has already run an optimisation from another start (whose basis it must not
    inherit), give the same bits; a resumed call continues with the basis it has (test_resumed_training_equals_one_call)"""
    from cherryml_amd import CherryBank
    z = load_golden("coevo_dense_traj.npz")
    t, C, mask = dense["t"], dense["C"], dense["mask"]
    u0, p0 = z["upper_diag0"], z["log_pi0"]
    with CherryBank(t, C) as bank:
        a = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=8, lr=0.1)
    with CherryBank(t, C) as bank:
        bank.train_pande_reversible(u0 + 1.5, p0, mask=mask, num_epochs=5, lr=0.1)   # faster matrix: another spectral bound
        r1 = bank.time_basis_info()["rho_max"]
        b = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=8, lr=0.1)
        r2 = bank.time_basis_info()["rho_max"]
    assert r1 != r2
    # (the handle's eigensolver is warm-started from the first optimisation's eigenvectors in its first epoch: the same matrix
    # from another start converges to the same decomposition to rounding, not to the bit -- so compare to rounding here ...)
    assert np.abs(a["loss"] - b["loss"]).max() < 1e-12 * np.abs(a["loss"]).max()
    with CherryBank(t, C) as bank:   # ... and to the bit between two fresh handles
        c = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=8, lr=0.1)
    assert np.array_equal(a["loss"], c["loss"]) and np.array_equal(a["Q_last"], c["Q_last"])


def test_the_next_basis_is_built_beside_the_epochs_and_swapped_at_a_fixed_epoch(dense, monkeypatch):
    """CB_TB_TEST_WARN="100 3" (test hook): a helper thread starts on the next basis as soon as one is installed and the trainer
    swaps it in three epochs later -- what a real optimisation does a few times in thousands of epochs when max |Q_ii| drifts.
    The swap epoch depends on the sigma sequence only: two runs give the same bits however long the thread took, and the loss
    curve is the one of the unhooked run to rounding (other bases, other sums)."""
    from cherryml_amd import CherryBank
    z = load_golden("coevo_dense_traj.npz")
    t, C, mask = dense["t"], dense["C"], dense["mask"]
    u0, p0 = z["upper_diag0"], z["log_pi0"]
    E = 16
    with CherryBank(t, C) as bank:
        ref = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
        assert bank.time_basis_info()["builds"] == 1
    monkeypatch.setenv("CB_TB_TEST_WARN", "100 3")
    runs = []
    for _ in range(2):
        with CherryBank(t, C) as bank:
            a = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=10, lr=0.1)
            b = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E - 10, lr=0.1, resume=True)   # a build pending across calls
            info = bank.time_basis_info()
        runs.append((np.concatenate([a["loss"], b["loss"]]), b["Q_last"], info))
    print(runs[0][2])
    assert runs[0][2]["builds"] >= 4 and runs[0][2]["repeated_epochs"] == 0
    assert np.array_equal(runs[0][0], runs[1][0]) and np.array_equal(runs[0][1], runs[1][1])
    assert np.abs(runs[0][0] - ref["loss"]).max() < 1e-11 * np.abs(ref["loss"]).max()
    assert relerr(runs[0][1], ref["Q_last"]) < 1e-8


def test_a_grid_that_needs_too_many_skeleton_buckets_keeps_the_per_bucket_products():
    from cherryml_amd import CherryBank
    rng = np.random.default_rng(11)
    S, B = 48, 100
    Q, pi = _reversible(S, rng, 5.0)
    t = np.geomspace(1e-8, 1e6, B)                 # fourteen decades: more than 40 gradient skeleton buckets
    C = _sym_counts(S, B, rng)
    with CherryBank(t, C) as a, CherryBank(t, C, per_bucket_products=True) as b:
        la, ga = a.loss_grad(Q, pi)
        assert not a.last_bank_form()["time_basis"] and a.time_basis_info()["builds"] == 0
        lb, gb = b.loss_grad(Q, pi)
    assert abs(la[0] - lb[0]) <= 1e-13 * abs(lb[0]) and relerr(ga[0], gb[0]) < 1e-11


def test_mixed_bank_in_the_time_basis(dense):
    """CB_MIXED: forward products, P_b, loss and G_b in float64, Gh_r rounded to float32 once, the two gradient products on the
    float32 MFMA -- against the float64 time basis (loss: the same arithmetic; gradient: float32 rounding of Gh_r and Th_r) and
    against three float32 products per bucket (CB_PER_BUCKET_PRODUCTS), on the bench bank and on a ragged one"""
    from cherryml_amd import CherryBank
    z = load_golden("coevo_dense_eval.npz")
    keep = (dense["mask"] != 0) | np.eye(400, dtype=bool)
    Q = np.zeros((400, 400))
    Q[keep] = z["Q_support_f64"]
    p = np.exp(z["log_pi"] - z["log_pi"].max())
    cases = [(dense["t"], dense["C"], Q, p / p.sum())]
    rng = np.random.default_rng(3)
    Qr, pir = _reversible(50, rng, 2.0)
    cases.append((0.03 * 1.1 ** (np.arange(129) - 64), _sym_counts(50, 129, rng, 0.3), Qr, pir))
    for t, C, Qx, pix in cases:
        with CherryBank(t, C) as a, CherryBank(t, C, dtype="mixed") as m, CherryBank(t, C, dtype="mixed", per_bucket_products=True) as mb:
            la, ga = a.loss_grad(Qx, pix)
            lm, gm = m.loss_grad(Qx, pix)
            lb, gb = mb.loss_grad(Qx, pix)
            assert m.last_bank_form()["time_basis"] and not mb.last_bank_form()["time_basis"]
        e64, eb = relerr(gm[0], ga[0]), relerr(gm[0], gb[0])
        print(f"S {C.shape[1]}: mixed time basis: loss {abs(lm[0] - la[0]) / abs(la[0]):.1e}, dL/dQ {e64:.1e} from float64, {eb:.1e} from the mixed per-bucket products")
        assert abs(lm[0] - la[0]) < 1e-13 * abs(la[0]) and e64 < 2e-6 and eb < 4e-6
