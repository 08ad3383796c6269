"""End-to-end parity of the mirrored Python API on the MI355X against the
reference's outputs (golden trajectories, f64 recipe) and the oracle."""
import os

import numpy as np
import pandas as pd
import pytest
import torch

from conftest import load_golden, relerr

pytestmark = pytest.mark.gpu


def _write_inputs(tmp_path, g, states, with_init=True):
    from cherryml_amd.io import write_count_matrices, write_rate_matrix
    cpath = str(tmp_path / "counts.txt")
    write_count_matrices([(float(t), pd.DataFrame(C, index=states, columns=states))
                          for t, C in zip(g["t"], g["C"])], cpath)
    ipath = None
    if with_init and "init" in g:
        ipath = str(tmp_path / "init.txt")
        write_rate_matrix(g["init"], states, ipath)
    mpath = None
    if "mask" in g and not np.all(g["mask"] == 1):
        mpath = str(tmp_path / "mask.txt")
        pd.DataFrame(g["mask"].astype(int), index=states, columns=states).to_csv(mpath, sep=" ")
    return cpath, ipath, mpath


@pytest.mark.parametrize("device", ["cuda", "cpu"])
@pytest.mark.parametrize("case,states", [("toy3_init", list("ABC")), ("toy3_mask", list("ABC"))])
def test_stage_function_matches_reference_trajectory(case, states, device, tmp_path):
    """quantized_transitions_mle (files in, files out) vs the reference's f64 recipe.  device="cpu" is the
    reference's default and what its own tests pass (tests/estimation_tests/quantized_transitions_mle_test.py):
    accepted, same result -- this package has one execution target (cherryml_amd/_device.py)."""
    import cherryml_amd
    from cherryml_amd.io import read_rate_matrix
    e = load_golden(f"eval_{case}.npz")
    g = load_golden(f"traj_{case}.npz")
    cpath, ipath, mpath = _write_inputs(tmp_path, e, states)
    out = str(tmp_path / "out")
    ret = cherryml_amd.quantized_transitions_mle(
        count_matrices_path=cpath, initialization_path=ipath, mask_path=mpath,
        output_rate_matrix_dir=out, stationary_distribution_path=None,
        rate_matrix_parameterization="pande_reversible", device=device, learning_rate=1e-1,
        num_epochs=int(g["num_epochs"]), do_adam=True)
    assert ret is None  # no cache dir set: straight call, like the reference
    have = set(os.listdir(out))
    assert {str(f) for f in g["files"]} - {"training_plot.png"} <= have
    df = pd.read_csv(os.path.join(out, "df_res.txt"))
    assert np.allclose(df.loss.to_numpy(), g["loss_f64"], rtol=1e-9, atol=0)
    for key in ["Q_best", "Q_last", "Q_1", "Q_2"]:
        got = read_rate_matrix(os.path.join(out, key + ".txt")).to_numpy()
        assert relerr(got, g[key + "_f64"]) < 1e-6, key
    res = read_rate_matrix(os.path.join(out, "result.txt")).to_numpy()
    assert relerr(res, g["Q_best_f64"]) < 1e-6
    # distance to the reference exactly as a user runs it (float32 expm): reported bound
    assert relerr(res, g["result_f32"]) < 1e-3
    # mask pattern is respected (reference test_smoke_toy_matrix_mask)
    assert np.all((res != 0) == (e["mask"] != 0))
    prof = open(os.path.join(out, "profiling.txt")).read().split()
    assert prof[:2] == ["Total", "time:"] and float(prof[2]) > 0


def test_mask_incompatible_init_raises(tmp_path):
    import cherryml_amd
    e = dict(load_golden("eval_toy3_init.npz"))
    e["mask"] = load_golden("eval_toy3_mask.npz")["mask"]
    cpath, ipath, mpath = _write_inputs(tmp_path, e, list("ABC"))
    with pytest.raises(ValueError):
        cherryml_amd.quantized_transitions_mle(
            count_matrices_path=cpath, initialization_path=ipath, mask_path=mpath,
            output_rate_matrix_dir=str(tmp_path / "o"), device="cuda", num_epochs=3)


def test_learner_class_lg_bank_100_epochs():
    """RateMatrixLearner in memory (the SiteRM entry, _site_specific_rate_matrix.py:64-83)
    on a 129-bucket LG-shaped bank: 1e-6 bar on Q vs the reference's f64 recipe."""
    from cherryml_amd import RateMatrixLearner
    g = load_golden("traj_lgbank.npz")
    states = [str(s) for s in load_golden("data_lg.npz")["states"]]
    learner = RateMatrixLearner(
        branches=list(g["t"]), mats=list(g["C"]), states=states, output_dir=None,
        stationnary_distribution=None, device="cuda", mask=None,
        rate_matrix_parameterization="pande_reversible", initialization=g["init"],
        skip_writing_to_output_dir=True)
    with pytest.raises(ValueError):
        learner.get_learnt_rate_matrix()
    learner.train(lr=0.1, num_epochs=int(g["num_epochs"]), do_adam=True, loss_normalization=True,
                  return_best_iter=True)
    assert np.allclose(learner.df_res.loss.to_numpy(), g["loss_f64"], rtol=1e-9, atol=0)
    # every row's `time` is filled (trainer.py:207-217): seconds since the start, from the device's own clock stamps
    tt = learner.df_res.time.to_numpy()
    assert np.all(np.isfinite(tt)) and tt[0] > 0 and np.all(np.diff(tt) > 0) and tt[-1] < 60.0
    assert 1e-6 < np.median(np.diff(tt)) < 5e-3      # an LG epoch is tens of microseconds
    Q = learner.get_learnt_rate_matrix()
    assert list(Q.index) == states
    assert relerr(Q.to_numpy(), g["Q_best_f64"]) < 1e-6
    assert relerr(learner.Q_dict["Q_last"], g["Q_last_f64"]) < 1e-6
    # reported: distance to the as-is float32 reference
    print("dist to f32 reference Q_best:", relerr(Q.to_numpy(), g["Q_best_f32"]))


def test_random_init_symmetric_mask_20_states():
    """No initialisation: seed-0 randn parameters, 20x20 symmetric mask, 50 epochs."""
    from cherryml_amd import RateMatrixLearner
    e = load_golden("eval_s20_symmask.npz")
    g = load_golden("traj_s20_symmask.npz")
    states = [str(s) for s in load_golden("data_lg.npz")["states"]]
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        mpath = os.path.join(d, "m.txt")
        np.savetxt(mpath, e["mask"], fmt="%d")
        learner = RateMatrixLearner(
            branches=list(e["t"]), mats=list(e["C"]), states=states, output_dir=None,
            stationnary_distribution=None, device="cuda", mask=mpath,
            initialization=None, skip_writing_to_output_dir=True)
        learner.train(lr=0.1, num_epochs=int(g["num_epochs"]), do_adam=True,
                      loss_normalization=True)
    assert np.allclose(learner.df_res.loss.to_numpy(), g["loss_f64"], rtol=1e-8, atol=0)
    assert relerr(learner.Q_dict["Q_best"], g["Q_best_f64"]) < 1e-6


@pytest.mark.parametrize("fused", [True, False])
def test_coevolution_400_states_three_epochs(fused):
    """fused=True: the C-driven device loop (train_large.hip.h); False: torch glue + cb_loss_grad.
    Both against the reference's own f64 trajectory."""
    from cherryml_amd import RateMatrix, train_quantization
    from torch.utils.data import TensorDataset
    e = load_golden("eval_s400_mask.npz")
    g = load_golden("traj_s400_mask.npz")
    mod = RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(e["mask"]),
                     pi=torch.ones(400, dtype=torch.float64) / 400, pi_requires_grad=True)
    with torch.no_grad():
        mod.upper_diag.copy_(torch.tensor(g["upper_diag0_f64"]))
        mod._pi.copy_(torch.tensor(g["log_pi0_f64"]))
    mod = mod.to("cuda")
    opt = torch.optim.Adam(mod.parameters(), lr=0.1)
    ds = TensorDataset(torch.tensor(e["t"]), torch.tensor(e["C"]))
    df, Qd = train_quantization(mod, ds, num_epochs=3, optimizer=opt, loss_normalization=True, fused=fused)
    assert np.allclose(df.loss.to_numpy(), g["loss_f64"], rtol=1e-10, atol=0)
    assert relerr(Qd["Q_best"], g["Q_best_f64"]) < 1e-8
    assert relerr(Qd["Q_last"], g["Q_last_f64"]) < 1e-8
    assert set(Qd) >= {"Q_1", "Q_2", "Q_best", "Q_last"}
    tt = df.time.to_numpy()   # fused: device clock stamps per epoch; torch glue: host time per epoch
    assert np.all(np.isfinite(tt)) and tt[0] > 0 and np.all(np.diff(tt) > 0)


def test_fused_large_trainer_matches_torch_glue_over_40_epochs():
    """Odd size (S = 50: padding, partial tiles), empty buckets, a symmetric mask with zeros,
    40 epochs: C-driven loop vs torch glue of the same package, and vs the oracle."""
    from cherryml_amd import CherryBank
    from oracle import ratelearn_oracle as orc
    rng = np.random.default_rng(5)
    S, B, E = 50, 9, 40
    t = np.sort(rng.uniform(0.02, 2.0, size=B))
    C = rng.poisson(4.0, size=(B, S, S)).astype(np.float64)
    C[2] = 0.0
    mask = (rng.random((S, S)) < 0.7).astype(np.float64)
    mask = np.triu(mask, 1)
    mask = mask + mask.T
    C *= (mask + np.eye(S))
    u0 = rng.normal(0.0, 0.3, size=S * (S - 1) // 2)
    p0 = rng.normal(0.0, 0.2, size=S)
    ref = orc.train(t, C, mask, upper_diag=u0, log_pi=p0, num_epochs=E, dtype=torch.float64)
    with CherryBank(t, C) as bank:
        r = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
    assert np.allclose(r["loss"], ref["loss"], rtol=1e-9, atol=0)
    assert relerr(r["Q_best"], ref["Q_best"]) < 1e-6
    assert relerr(r["Q_last"], ref["Q_last"]) < 1e-6
    for k, Qk in r["Q_pow2"].items():
        assert relerr(Qk, ref[f"Q_{k}"]) < 1e-6, k
    assert np.allclose(r["upper_diag"], ref["upper_diag"], rtol=1e-6, atol=1e-8)
    # cb_train_epoch_times: one stamp per epoch, increasing, of the size of an epoch
    assert r["time"].shape == (E,) and np.all(np.diff(r["time"]) > 0) and 0 < r["time"][0] < r["time"][-1] < 30.0


def test_siterm_vectorized_matches_reference():
    from cherryml_amd import quantized_transitions_mle_vectorized_over_sites as qvec
    for name in ["siterm_dna.npz", "siterm_aa.npz"]:
        g = load_golden(name)
        E = g["lpe_init"].shape[0]
        r = qvec(g["counts"], g["times"], num_epochs=E, initialization=g["init"], device="cuda")
        assert np.allclose(r["loss_per_epoch_per_site"], g["lpeps_init"], rtol=1e-8, atol=0)
        assert np.allclose(r["loss_per_epoch"], g["lpe_init"], rtol=1e-8, atol=0)
        assert relerr(r["res"], g["res_init"]) < 1e-6
    # no initialisation: the reference's parameters are float32 there, ours float64
    g = load_golden("siterm_dna.npz")
    r = qvec(g["counts"], g["times"], num_epochs=g["lpe_rand"].shape[0], device="cuda")
    assert np.allclose(r["loss_per_epoch_per_site"], g["lpeps_rand"], rtol=1e-4, atol=0)
    assert relerr(r["res"], g["res_rand"]) < 1e-3


@pytest.mark.parametrize("case", ["toy3_init", "toy3_mask", "s20_symmask", "lgbank"])
def test_fused_trainer_matches_oracle_and_torch_glue_path(case):
    """cb_train_pande_reversible (whole loop in one kernel) vs the oracle's f64 trajectory,
    and vs the torch-glue path of the same package."""
    from cherryml_amd import CherryBank, RateMatrix, train_quantization
    from oracle import ratelearn_oracle as orc
    from torch.utils.data import TensorDataset
    if case == "lgbank":
        g = load_golden("traj_lgbank.npz")
        t, C, mask, E = g["t"], g["C"], np.ones((20, 20)), 60
        u0, p0 = orc.invert_pande_reversible(g["init"], mask)
    else:
        e = load_golden(f"eval_{case}.npz")
        g = load_golden(f"traj_{case}.npz")
        t, C, mask, E = e["t"], e["C"], e["mask"], int(g["num_epochs"])
        u0, p0 = g["upper_diag0_f64"], g["log_pi0_f64"]
    ref = orc.train(t, C, mask, upper_diag=u0, log_pi=p0, num_epochs=E, dtype=torch.float64)
    with CherryBank(t, C) as bank:
        r = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
    assert np.allclose(r["loss"], ref["loss"], rtol=1e-9, atol=0)
    assert relerr(r["Q_best"], ref["Q_best"]) < 1e-6
    assert relerr(r["Q_last"], ref["Q_last"]) < 1e-6
    for k, Qk in r["Q_pow2"].items():
        assert relerr(Qk, ref[f"Q_{k}"]) < 1e-6, k
    assert set(r["Q_pow2"]) == {1 << i for i in range(E.bit_length()) if (1 << i) <= E}
    fin = np.isfinite(u0)
    assert np.array_equal(np.isfinite(r["upper_diag"]), fin)  # masked logits stay -inf
    assert np.allclose(r["upper_diag"][fin], ref["upper_diag"][fin], rtol=1e-6, atol=1e-8)
    # the mirrored train_quantization picks the fused path by itself and agrees with fused=False
    outs = []
    for fused in (None, False):
        S = mask.shape[0]
        mod = RateMatrix(num_states=S, mode="pande_reversible", mask=torch.tensor(mask),
                         pi=torch.ones(S, dtype=torch.float64) / S, pi_requires_grad=True)
        with torch.no_grad():
            mod.upper_diag.copy_(torch.tensor(u0))
            mod._pi.copy_(torch.tensor(p0))
        mod = mod.to("cuda")
        opt = torch.optim.Adam(mod.parameters(), lr=0.1)
        df, Qd = train_quantization(mod, TensorDataset(torch.tensor(t), torch.tensor(C)),
                                    num_epochs=E, optimizer=opt, fused=fused)
        outs.append((df.loss.to_numpy(), Qd))
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-10, atol=0)
    assert set(outs[0][1]) == set(outs[1][1])
    for k in outs[0][1]:
        assert relerr(outs[0][1][k], outs[1][1][k]) < 1e-7, k


def test_fused_sgd_option():
    from cherryml_amd import CherryBank
    from oracle import ratelearn_oracle as orc
    e = load_golden("eval_toy3_init.npz")
    g = load_golden("traj_toy3_init.npz")
    ref = orc.train(e["t"], e["C"], e["mask"], upper_diag=g["upper_diag0_f64"],
                    log_pi=g["log_pi0_f64"], num_epochs=5, lr=0.01, do_adam=False)
    with CherryBank(e["t"], e["C"]) as bank:
        r = bank.train_pande_reversible(g["upper_diag0_f64"], g["log_pi0_f64"], mask=e["mask"],
                                        num_epochs=5, lr=0.01, do_adam=False)
    assert np.allclose(r["loss"], ref["loss"], rtol=1e-11, atol=0)


def test_fused_siterm_matches_torch_glue_path():
    from cherryml_amd import quantized_transitions_mle_vectorized_over_sites as qvec
    g = load_golden("siterm_aa.npz")
    E = g["lpe_init"].shape[0]
    a = qvec(g["counts"], g["times"], num_epochs=E, initialization=g["init"], device="cuda", fused=True)
    b = qvec(g["counts"], g["times"], num_epochs=E, initialization=g["init"], device="cuda", fused=False)
    assert np.allclose(a["loss_per_epoch_per_site"], b["loss_per_epoch_per_site"], rtol=1e-10, atol=0)
    assert relerr(a["res"], b["res"]) < 1e-8
    assert np.allclose(a["loss_per_epoch_per_site"], g["lpeps_init"], rtol=1e-8, atol=0)


def test_reference_20x20_non_symmetric_mask_case(tmp_path):
    """The reference's test_smoke 20x20 random (non-symmetric) mask, no initialisation:
    non-reversible Q -> general HIP path; trajectory vs the reference's f64 recipe."""
    import cherryml_amd
    from cherryml_amd.io import read_rate_matrix
    e = load_golden("eval_s20_mask.npz")
    g = load_golden("traj_s20_mask.npz")
    states = [str(s) for s in load_golden("data_lg.npz")["states"]]
    cpath, ipath, mpath = _write_inputs(tmp_path, e, states, with_init=False)
    out = str(tmp_path / "out")
    cherryml_amd.quantized_transitions_mle(
        count_matrices_path=cpath, initialization_path=None, mask_path=mpath,
        output_rate_matrix_dir=out, device="cuda", num_epochs=int(g["num_epochs"]))
    df = pd.read_csv(os.path.join(out, "df_res.txt"))
    assert np.allclose(df.loss.to_numpy(), g["loss_f64"], rtol=1e-8, atol=0)
    res = read_rate_matrix(os.path.join(out, "result.txt")).to_numpy()
    assert relerr(res, g["Q_best_f64"]) < 1e-6
    assert np.all((res != 0) == (e["mask"] != 0))  # learned zero pattern == mask


@pytest.mark.parametrize("mode", ["default", "pande", "stationary", "stationary_reversible"])
def test_other_parameterisations_run_and_descend(mode):
    from cherryml_amd import RateMatrix, train_quantization
    from torch.utils.data import TensorDataset
    e = load_golden("eval_toy3_init.npz")
    torch.manual_seed(0)
    mod = RateMatrix(num_states=3, mode=mode, mask=torch.ones(3, 3),
                     pi=torch.tensor([0.2, 0.3, 0.5], dtype=torch.float64),
                     pi_requires_grad=True).to("cuda")
    opt = torch.optim.Adam(mod.parameters(), lr=0.05)
    df, Qd = train_quantization(mod, TensorDataset(torch.tensor(e["t"]), torch.tensor(e["C"])),
                                num_epochs=20, optimizer=opt)
    loss = df.loss.to_numpy()
    assert np.all(np.isfinite(loss)) and loss[-1] < loss[0]


def test_lg_end_to_end_pipeline(tmp_path):
    """trees + site rates given -> counting [GPU] -> JTT-IPW -> optimiser [GPU]; every stage
    checked against the oracle chain on the same synthetic families."""
    import cherryml_amd
    from cherryml_amd import caching
    from cherryml_amd.io import read_count_matrices_arrays, read_rate_matrix
    from oracle import counting_oracle as co
    from oracle import ratelearn_oracle as orc
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "counting", "synth")
    fams = ["famA", "famB", "famC", "famD"]
    with pytest.raises(caching.CacheUsageError):
        cherryml_amd.lg_end_to_end_with_cherryml_optimizer(
            msa_dir=f"{d}/msa_dir", families=fams, tree_estimator=None,
            initial_tree_estimator_rate_matrix_path=None, tree_dir=f"{d}/tree_dir",
            site_rates_dir=f"{d}/site_rates_dir")
    caching.set_cache_dir(str(tmp_path / "cache"))
    try:
        res = cherryml_amd.lg_end_to_end_with_cherryml_optimizer(
            msa_dir=f"{d}/msa_dir", families=fams, tree_estimator=None,
            initial_tree_estimator_rate_matrix_path=None, num_epochs=40,
            tree_dir=f"{d}/tree_dir", site_rates_dir=f"{d}/site_rates_dir", edge_or_cherry="edge")
        res2 = cherryml_amd.lg_end_to_end_with_cherryml_optimizer(  # second call: all cached
            msa_dir=f"{d}/msa_dir", families=fams, tree_estimator=None,
            initial_tree_estimator_rate_matrix_path=None, num_epochs=40,
            tree_dir=f"{d}/tree_dir", site_rates_dir=f"{d}/site_rates_dir", edge_or_cherry="edge")
    finally:
        caching.set_cache_dir(None)
    assert res["learned_rate_matrix_path"] == res2["learned_rate_matrix_path"]
    grid = [float(x) for x in res["quantization_points"]]
    assert len(grid) == 129 and res["quantization_points"][0] == "0.00006730"
    aa = list("ARNDCQEGHILKMFPSTWYV")
    C = co.count_transitions(f"{d}/tree_dir", f"{d}/msa_dir", f"{d}/site_rates_dir", fams, aa, grid, "edge")
    q, Cg, st = read_count_matrices_arrays(os.path.join(res["count_matrices_dir_0"], "result.txt"))
    assert st == aa and np.array_equal(Cg, C)
    init = orc.jtt_ipw(np.array(grid), C)
    got_init = read_rate_matrix(os.path.join(res["jtt_ipw_dir_0"], "result.txt")).to_numpy()
    assert np.allclose(got_init, init, rtol=1e-12, atol=1e-15)
    ref = orc.train(np.array(grid), C, None, initialization=got_init, num_epochs=40)
    learned = read_rate_matrix(res["learned_rate_matrix_path"]).to_numpy()
    assert relerr(learned, ref["Q_best"]) < 1e-6
    assert res["time_counting"] > 0 and "time_optimization" in res["profiling_str"]


def test_coevolution_end_to_end_pipeline_small_alphabet(tmp_path):
    """co-evolution pipeline on the reference's tiny_2 data (4-letter alphabet -> 16 pair
    states): maximal matching -> co-counting [GPU] -> masked JTT-IPW -> optimiser [GPU]."""
    import cherryml_amd
    from cherryml_amd import caching
    from cherryml_amd.io import read_rate_matrix
    d = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "counting", "tiny_2")
    aa = ["I", "L", "S", "T"]
    pairs = [a + b for a in aa for b in aa]
    mask = np.array([[1 if (p[0] == q[0] or p[1] == q[1]) else 0 for q in pairs] for p in pairs])
    mpath = str(tmp_path / "mask.txt")
    pd.DataFrame(mask, index=pairs, columns=pairs).to_csv(mpath, sep=" ")
    caching.set_cache_dir(str(tmp_path / "cache"))
    try:
        res = cherryml_amd.coevolution_end_to_end_with_cherryml_optimizer(
            msa_dir=f"{d}/msa_dir", contact_map_dir=f"{d}/contact_map_dir",
            minimum_distance_for_nontrivial_contact=2, coevolution_mask_path=mpath,
            families=["fam1", "fam2", "fam3"], tree_estimator=None,
            initial_tree_estimator_rate_matrix_path=None, num_epochs=5, tree_dir=f"{d}/tree_dir",
            quantization_grid_center=4.0, quantization_grid_step=1.5, quantization_grid_num_steps=3,
            alphabet=aa, optimizer_initialization="random")
    finally:
        caching.set_cache_dir(None)
    Q = read_rate_matrix(res["learned_rate_matrix_path"])
    assert list(Q.index) == pairs
    assert np.all((Q.to_numpy() != 0) == (mask != 0))
    assert np.abs(Q.to_numpy().sum(1)).max() < 1e-12


def _random_reversible_init(rng, N):
    pi = rng.dirichlet(np.full(N, 8.0))
    R = rng.gamma(2.0, 0.3, size=(N, N))
    R = np.triu(R, 1)
    R = R + R.T
    d = np.sqrt(pi)
    Q = R * d[None, :] / d[:, None]
    return Q - np.diag(Q.sum(1))


def test_every_trainer_form_matches_the_oracle():
    """Which kernels a training call launches is decided by (S, L, parameterisation) alone -- the CB_LG_SPLIT / CB_SITE_SPLIT
    switches of rounds 1-2 are gone -- and cb_last_kernel_form says which: the site-parallel split (S <= 24), the LG split
    (one 25 .. 32-state bank), the one-kernel trainer (several sites or SiteRM at 25 .. 32 states), each against the oracle."""
    from cherryml_amd import CherryBank
    from cherryml_amd._siterm._vectorized import _invert
    from oracle import ratelearn_oracle as orc
    g = load_golden("traj_lgbank.npz")
    u0, p0 = orc.invert_pande_reversible(g["init"], np.ones((20, 20)))
    with CherryBank(g["t"], g["C"]) as bank:
        r = bank.train_pande_reversible(u0, p0, num_epochs=30, lr=0.1)
        assert bank.last_kernel_form() == 1510          # sp_bank<5, symmetric, two workgroups per CU>
    assert np.allclose(r["loss"], g["loss_f64"][:30], rtol=1e-9, atol=0)
    rng = np.random.default_rng(28)
    N, B, E = 28, 6, 12
    t = np.sort(rng.uniform(0.02, 2.0, size=B))
    for L in (1, 3):
        C = rng.poisson(3.0, size=(L, B, N, N)).astype(np.float64)
        C = C + C.transpose(0, 1, 3, 2) + 5.0 * np.eye(N)
        inits = [_random_reversible_init(rng, N) for _ in range(L)]
        ups = [orc.invert_pande_reversible(q, np.ones((N, N))) for q in inits]
        with CherryBank(np.tile(t, (L, 1)) if L > 1 else t, C if L > 1 else C[0]) as bank:
            got = bank.train_pande_reversible(np.array([u for u, _ in ups]) if L > 1 else ups[0][0],
                                              np.array([p for _, p in ups]) if L > 1 else ups[0][1], num_epochs=E, lr=0.1)
            assert bank.last_kernel_form() == (2000 if L == 1 else 3000)
        for l in range(L):
            ref = orc.train(t, C[l], np.ones((N, N)), upper_diag=ups[l][0], log_pi=ups[l][1], num_epochs=E, dtype=torch.float64)
            loss = got["loss"] if L == 1 else got["loss"][:, l]
            assert np.allclose(loss, ref["loss"], rtol=1e-9, atol=0), (L, l)
            assert relerr(got["Q_last"] if L == 1 else got["Q_last"][l], ref["Q_last"]) < 1e-6
    # SiteRM parameterisation at 28 states: the one-kernel trainer
    L = 4
    C = rng.poisson(3.0, size=(L, B, N, N)).astype(np.float64)
    C = C + C.transpose(0, 1, 3, 2) + 5.0 * np.eye(N)
    T = np.sort(rng.uniform(0.02, 2.0, size=(L, B)), axis=1)
    init = np.array([_random_reversible_init(rng, N) for _ in range(L)])
    ref = orc.siterm_train(C, T, E, initialization=init)
    th0, Th0 = _invert(init)
    with CherryBank(T, C) as bank:
        got = bank.train_siterm(th0, Th0, E, lr=0.1)
        assert bank.last_kernel_form() == 3000
    assert np.allclose(got["loss_per_epoch_per_site"], ref["loss_per_epoch_per_site"], rtol=1e-8, atol=0)
    assert max(relerr(got["res"][l], ref["res"][l]) for l in range(L)) < 1e-6


def test_fused_trainers_with_empty_buckets():
    """Empty buckets are dropped at cb_create; trajectories still equal the oracle's on the
    full (zero-padded) bank -- single bank (split LG trainer) and SiteRM (ragged per site)."""
    from cherryml_amd import CherryBank
    from oracle import ratelearn_oracle as orc
    g = load_golden("traj_lgbank.npz")
    t, C, mask, E = g["t"], g["C"].copy(), np.ones((20, 20)), 30
    C[::3] = 0.0
    C[100:] = 0.0
    u0, p0 = orc.invert_pande_reversible(g["init"], mask)
    ref = orc.train(t, C, mask, upper_diag=u0, log_pi=p0, num_epochs=E, dtype=torch.float64)
    with CherryBank(t, C) as bank:
        assert bank.live_buckets[0] == int(np.any(C.reshape(len(t), -1) != 0, axis=1).sum())
        r = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
    assert np.allclose(r["loss"], ref["loss"], rtol=1e-9, atol=0)
    assert relerr(r["Q_best"], ref["Q_best"]) < 1e-6
    s = load_golden("siterm_aa.npz")
    counts, times = s["counts"].copy(), s["times"]
    for l in range(counts.shape[0]):
        counts[l, l % counts.shape[1]::2] = 0.0
    E = 12
    ref = orc.siterm_train(counts, times, E, initialization=s["init"])
    from cherryml_amd import quantized_transitions_mle_vectorized_over_sites as qvec
    a = qvec(counts, times, num_epochs=E, initialization=s["init"], device="cuda", fused=True)
    assert np.allclose(a["loss_per_epoch_per_site"], ref["loss_per_epoch_per_site"], rtol=1e-8, atol=0)
    assert relerr(a["res"], ref["res"]) < 1e-6


def test_sharded_bank_from_rank_counts_on_device_tensor():
    """The multi-GPU constructor with this rank's counts already on the GPU (world = 1: no process
    group; the N > 1 collectives are covered by the gloo tests and by `bench.py --force-sharded`)."""
    from cherryml_amd.distributed import ShardedBank
    from oracle import ratelearn_oracle as orc
    g = load_golden("traj_lgbank.npz")
    t, C = g["t"][::4], g["C"][::4]
    sb = ShardedBank.from_rank_counts(t, torch.tensor(C, device="cuda"))
    Q = torch.tensor(g["init"], dtype=torch.float64, device="cuda", requires_grad=True)
    pi = torch.tensor(orc.stationary_distribution(g["init"]), device="cuda")
    loss = sb.loss(Q, pi, normalize=True)[0]
    loss.backward()
    Qr = torch.tensor(g["init"], dtype=torch.float64, requires_grad=True)
    ref = orc.bank_loss(Qr, torch.tensor(t), torch.tensor(C))
    ref.backward()
    assert abs(loss.item() - ref.item()) < 1e-12 * abs(ref.item())
    assert relerr(Q.grad.cpu().numpy(), Qr.grad.numpy()) < 1e-10
    sb.close()


@pytest.mark.parametrize("N,B", [(21, 7), (24, 5), (12, 9), (7, 3), (17, 4), (16, 4), (9, 3), (3, 5)])
def test_siterm_trainer_other_state_counts(N, B):
    """The 4x4-tile path instantiates 1, 2, 4, 5 and 6 tiles per side (S <= 4, 8, 16, 20, 24) and the warm eigensolver's
    4x4x4 products 1 .. 6 tiles per side (S <= 4, 8, 12, 16, 20, 24; 17 and 9 states sit just above a tile boundary, i.e.
    on zero-padded frames): fused SiteRM trainer (site-parallel split) against the oracle for alphabets other than 4 and
    20 (21 = amino acids + gap)."""
    from cherryml_amd import quantized_transitions_mle_vectorized_over_sites as qvec
    from oracle import ratelearn_oracle as orc
    rng = np.random.default_rng(N * 10 + B)
    L, E = 6, 15
    counts = rng.poisson(3.0, size=(L, B, N, N)).astype(np.float64)
    counts = counts + counts.transpose(0, 1, 3, 2) + 5.0 * np.eye(N)
    counts[2, 1] = 0.0                                  # an empty bucket in one site
    times = np.sort(rng.uniform(0.02, 2.0, size=(L, B)), axis=1)
    init = np.zeros((L, N, N))
    for l in range(L):
        pi = rng.dirichlet(np.full(N, 8.0))
        R = rng.gamma(2.0, 0.3, size=(N, N))
        R = np.triu(R, 1)
        R = R + R.T
        d = np.sqrt(pi)
        init[l] = R * d[None, :] / d[:, None]
        init[l] -= np.diag(init[l].sum(1))
    ref = orc.siterm_train(counts, times, E, initialization=init)
    got = qvec(counts, times, num_epochs=E, initialization=init, device="cuda", fused=True)
    assert np.allclose(got["loss_per_epoch_per_site"], ref["loss_per_epoch_per_site"], rtol=1e-8, atol=0)
    for l in range(L):
        assert relerr(got["res"][l], ref["res"][l]) < 1e-6, l


def test_in_library_allreduce_with_a_single_rank_rccl_communicator():
    """cb_allreduce_setup: a real RCCL communicator (one rank, created through ctypes on torch's own
    librccl) + the address of its ncclAllReduce.  With one rank the sums equal the local results; the
    normaliser becomes the global n_total handed over."""
    import ctypes as C
    import glob
    from cherryml_amd import CherryBank
    from oracle import ratelearn_oracle as orc
    libs = glob.glob(os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so*"))
    if not libs:
        pytest.skip("no librccl next to torch")
    rccl = C.CDLL(libs[0])

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    torch.cuda.set_device(0)
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    fn = C.cast(rccl.ncclAllReduce, C.c_void_p).value
    try:
        for name in ("traj_lgbank.npz", "eval_s400_mask.npz"):
            g = load_golden(name)
            t, Cc = g["t"][:6], g["C"][:6]
            Q = g["init"] if "init" in g else g["Q_f64"]
            pi = orc.stationary_distribution(Q)
            with CherryBank(t, Cc) as bank:
                l0, d0 = bank.loss_grad(Q, pi)
                bank.allreduce_setup(comm.value, fn, 2.0 * bank.total_counts)   # pretend the job holds twice the counts
                l1, d1 = bank.loss_grad(Q, pi)
                bank.allreduce_setup(None, None, None)
                l2, d2 = bank.loss_grad(Q, pi)
            assert np.allclose(l1, 0.5 * l0, rtol=1e-14) and relerr(d1[0], 0.5 * d0[0]) < 1e-14
            # (the third solve is warm-started, so equal to rounding, not bitwise)
            assert np.allclose(l2, l0, rtol=1e-13) and relerr(d2[0], d0[0]) < 1e-12
    finally:
        rccl.ncclCommDestroy.argtypes = [C.c_void_p]
        rccl.ncclCommDestroy(comm)


def test_c_driven_sharded_trainer_with_a_single_rank_communicator():
    """ShardedBank.enable_in_library_allreduce + train_pande_reversible: the whole sharded epoch loop
    from C (eigh, own buckets, ncclAllReduce of (loss, dL/dA) on the handle's stream, Adam).  One rank:
    the sums are the local values, so the run must reproduce the unsharded C-driven loop and the oracle;
    asymmetric counts make the direct log-pi term (job-wide count margins) non-zero."""
    from cherryml_amd import CherryBank
    from cherryml_amd.distributed import ShardedBank
    from oracle import ratelearn_oracle as orc
    rng = np.random.default_rng(11)
    S, B, E = 50, 7, 12
    t = np.sort(rng.uniform(0.02, 2.0, size=B))
    C = rng.poisson(4.0, size=(B, S, S)).astype(np.float64)
    C[3] = 0.0
    mask = np.ones((S, S)) - np.eye(S)
    u0 = rng.normal(0.0, 0.3, size=S * (S - 1) // 2)
    p0 = rng.normal(0.0, 0.2, size=S)
    ref = orc.train(t, C, mask, upper_diag=u0, log_pi=p0, num_epochs=E, dtype=torch.float64)
    with CherryBank(t, C) as bank:
        plain = bank.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
    torch.cuda.set_device(0)
    sharded = ShardedBank.from_rank_counts(t, torch.tensor(C, device="cuda:0"))
    try:
        with pytest.raises(RuntimeError):
            sharded.train_pande_reversible(u0, p0, mask=mask, num_epochs=1)
        sharded.enable_in_library_allreduce()
        assert sharded.rccl.world == 1
        r = sharded.train_pande_reversible(u0, p0, mask=mask, num_epochs=E, lr=0.1)
    finally:
        sharded.close()
    assert np.allclose(r["loss"], ref["loss"], rtol=1e-9, atol=0)
    assert np.allclose(r["loss"], plain["loss"], rtol=1e-12, atol=0)
    assert relerr(r["Q_best"], ref["Q_best"]) < 1e-6
    assert relerr(r["Q_best"], plain["Q_best"]) < 1e-9
    # a small bank does not shard: the fused small-state trainers refuse a handle with a communicator
    g = load_golden("traj_lgbank.npz")
    sh = ShardedBank.from_rank_counts(g["t"], torch.tensor(g["C"], device="cuda:0"))
    try:
        sh.enable_in_library_allreduce()
        S2 = g["C"].shape[-1]
        with pytest.raises(Exception):
            sh.train_pande_reversible(np.zeros(S2 * (S2 - 1) // 2), np.zeros(S2), num_epochs=2)
    finally:
        sh.close()


def test_hybrid_and_tournament_eigensolvers_agree_on_a_degenerate_spectrum(monkeypatch):
    """Warm-started solves of the C-driven trainer: the hybrid sweeps (far pairs by one GEMM-based
    rotation, near pairs by banded Jacobi passes) against the full tournament sweeps they replaced
    (CB_NO_HYBRID=1), started from an exact product model Q (x) I + I (x) Q -- its spectrum
    lambda_i + lambda_j has exactly degenerate pairs, the case in which the first-order formula divides
    by zero and the band has to take over."""
    import bench
    from cherryml_amd import CherryBank, RateMatrix
    rng = np.random.default_rng(0)
    wl = bench.make_workload("coevo400", 0, rng)
    lg = bench.lg_matrix()
    pi1 = bench.stationary(lg)
    Q0 = np.kron(lg, np.eye(20)) + np.kron(np.eye(20), lg)
    mod = RateMatrix(num_states=400, mode="pande_reversible", mask=torch.tensor(wl["mask"]),
                     pi=torch.tensor(np.kron(pi1, pi1)), pi_requires_grad=True, initialization=Q0)
    u0, p0 = mod.upper_diag.detach().numpy().copy(), mod._pi.detach().numpy().copy()
    runs = {}
    for name, env in (("hybrid", None), ("tournament", "1")):
        if env is None:
            monkeypatch.delenv("CB_NO_HYBRID", raising=False)
        else:
            monkeypatch.setenv("CB_NO_HYBRID", env)
        with CherryBank(wl["t"][::4], wl["C"][::4]) as bank:
            runs[name] = bank.train_pande_reversible(u0, p0, mask=wl["mask"], num_epochs=8, lr=0.1)
    a, b = runs["hybrid"], runs["tournament"]
    assert np.all(np.isfinite(a["loss"])) and a["loss"][-1] < a["loss"][0]
    assert np.allclose(a["loss"], b["loss"], rtol=1e-11, atol=0)
    assert relerr(a["Q_last"], b["Q_last"]) < 1e-8 and relerr(a["Q_best"], b["Q_best"]) < 1e-8


def test_failing_rank_keeps_its_place_in_every_collective(monkeypatch):
    """VERDICT r1 (multi-GPU): a rank whose epoch fails must not strand its peers in ncclAllReduce.
    The all-reduce handed to cb_allreduce_setup is a counting callback here (one rank: the in-place sum
    is the identity); CB_FAULT_INJECT makes this rank's evaluation of epoch 2 fail.  The call returns
    the error AND has still entered both collectives of every one of the E epochs (with NaN payloads from
    epoch 2 on, which turn the peers' parameters NaN, so that they end with non-finite losses too instead
    of hanging).  (Non-finite parameters by themselves are not an error, as in the reference: the losses
    are NaN on every rank alike.)"""
    import ctypes as C
    from cherryml_amd import CherryBank
    rng = np.random.default_rng(5)
    S, B, E = 48, 5, 6
    t = np.sort(rng.uniform(0.05, 1.5, size=B))
    Cc = rng.poisson(3.0, size=(B, S, S)).astype(np.float64)
    Cc = Cc + Cc.transpose(0, 2, 1)
    calls = []
    proto = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
    cb = proto(lambda s, r, n, dt, op, comm, stream: (calls.append(int(n)), 0)[1])
    fn = C.cast(cb, C.c_void_p).value
    u0 = rng.normal(0.0, 0.3, size=S * (S - 1) // 2)
    p0 = rng.normal(0.0, 0.2, size=S)
    LD = -(-S // 16) * 16
    with CherryBank(t, Cc) as bank:
        bank.allreduce_setup(0xC0FFEE, fn, bank.total_counts)
        assert calls == [S]                                    # the count margins, once
        del calls[:]
        ok = bank.train_pande_reversible(u0, p0, num_epochs=E, lr=0.1)
        assert calls == [1, LD * LD] * E and np.all(np.isfinite(ok["loss"]))
        del calls[:]
        monkeypatch.setenv("CB_FAULT_INJECT", "2")
        with pytest.raises(ValueError, match="injected fault"):
            bank.train_pande_reversible(u0, p0, num_epochs=E, lr=0.1)
        assert calls == [1, LD * LD] * E, calls               # every epoch's two collectives were entered
        # single evaluations: a failing call still enters its collective
        del calls[:]
        monkeypatch.setenv("CB_FAULT_INJECT", "-1")
        Q = np.full((S, S), 1.0 / (S - 1))
        np.fill_diagonal(Q, -1.0)
        with pytest.raises(ValueError, match="injected fault"):
            bank.loss_grad(Q, np.full(S, 1.0 / S))
        assert calls == [1, S * S]
        monkeypatch.delenv("CB_FAULT_INJECT")
        del calls[:]
        bank.loss_grad(Q, np.full(S, 1.0 / S))
        assert calls == [1, S * S]


def test_non_symmetric_mask_above_32_states_trains_through_the_general_path():
    """SURVEY row +g2: a non-symmetric mask makes Q non-reversible; above 32 states that is the batched
    scaling-and-squaring path (general_large.hip.h).  Six epochs of `train_quantization` (torch keeps theta -> Q
    and Adam, HIP does loss + dL/dQ) against the oracle's float64 run of the reference algorithm; and one
    400-state evaluation (the co-evolution size) against torch.matrix_exp + autograd."""
    from cherryml_amd import CherryBank, RateMatrix, train_quantization
    from torch.utils.data import TensorDataset
    from oracle import ratelearn_oracle as orc
    rng = np.random.default_rng(3)
    S, B, E = 48, 5, 6
    t = np.sort(rng.uniform(0.02, 3.0, size=B))
    C = rng.poisson(2.0, size=(B, S, S)).astype(np.float64)
    mask = (rng.random((S, S)) < 0.7).astype(np.float64)
    np.fill_diagonal(mask, 0.0)
    assert not np.array_equal(mask, mask.T)
    u0 = rng.normal(0.0, 0.3, size=S * (S - 1) // 2)
    p0 = rng.normal(0.0, 0.2, size=S)
    ref = orc.train(t, C, mask, upper_diag=u0, log_pi=p0, num_epochs=E)
    mod = RateMatrix(num_states=S, mode="pande_reversible", mask=torch.tensor(mask),
                     pi=torch.ones(S, dtype=torch.float64) / S, pi_requires_grad=True)
    with torch.no_grad():
        mod.upper_diag.copy_(torch.tensor(u0))
        mod._pi.copy_(torch.tensor(p0))
    mod = mod.to("cuda")
    assert not mod.is_reversible()
    opt = torch.optim.Adam([mod._pi, mod.upper_diag], lr=0.1)
    df, Qd = train_quantization(mod, TensorDataset(torch.tensor(t), torch.tensor(C)), num_epochs=E, optimizer=opt)
    assert np.allclose(df.loss.to_numpy(), ref["loss"], rtol=1e-9, atol=0)
    assert relerr(Qd["Q_best"], ref["Q_best"]) < 1e-6 and relerr(Qd["Q_last"], ref["Q_last"]) < 1e-6
    # 400 states, general path, single evaluation
    S = 400
    Q = rng.uniform(0.0, 1.0, size=(S, S)) * (rng.random((S, S)) < 0.3) * (4.0 / S)
    np.fill_diagonal(Q, 0.0)
    Q -= np.diag(Q.sum(1))
    # (branch lengths at which no entry of P is tiny: both this path and torch.matrix_exp are accurate in the absolute
    # sense only, and counts sitting on 1e-10-sized entries of P would compare their rounding, not their algebra)
    t = np.array([0.5, 1.5, 6.0])
    C = rng.poisson(0.3, size=(3, S, S)).astype(np.float64)
    Qt = torch.tensor(Q, requires_grad=True)
    want = orc.bank_loss(Qt, torch.tensor(t), torch.tensor(C))
    want.backward()
    with CherryBank(t, C) as bank:
        loss, dQ = bank.loss_grad_general(Q)
        P = bank.expm_bank(Q, None)[0]
    assert abs(loss[0] - want.item()) < 1e-12 * abs(want.item())
    assert relerr(dQ[0], Qt.grad.numpy()) < 1e-10
    assert np.abs(P - orc.expm_bank(Q, t)).max() < 1e-13


@pytest.mark.parametrize("N", [4, 12, 20, 21, 24])
def test_symmetric_counts_take_the_symmetric_quad_form(N, monkeypatch):
    """Symmetric count matrices (cherry counting and the SiteRM assembly with reverse transitions produce them) make
    Pt, G and W symmetric: `sp_bank<TS, true>` computes the tiles on or above the diagonal only and `sp_finish`
    mirrors M.  Same losses and matrices as the full form (CB_NO_SYM=1 at cb_create) and as the oracle, for both
    parameterisations (SiteRM and the per-site pande_reversible)."""
    from cherryml_amd import CherryBank
    from cherryml_amd._siterm._vectorized import _invert
    from oracle import ratelearn_oracle as orc
    rng = np.random.default_rng(100 + N)
    L, B, E = 5, 7, 6
    times = np.sort(rng.uniform(0.05, 2.0, size=(L, B)), axis=1)
    C = rng.poisson(3.0, size=(L, B, N, N)).astype(np.float64)
    C = C + C.transpose(0, 1, 3, 2)
    C[1, 2] = 0.0                                    # an empty bucket
    Q0 = rng.uniform(0.2, 1.0, size=(L, N, N))
    pi = rng.dirichlet(np.full(N, 5.0), size=L)
    Q0 = 0.5 * (Q0 + Q0.transpose(0, 2, 1)) * pi[:, None, :]     # reversible initial matrices
    for l in range(L):
        np.fill_diagonal(Q0[l], 0.0)
        Q0[l] -= np.diag(Q0[l].sum(1))
    th0, Th0 = _invert(Q0)
    runs = {}
    for name, env in (("sym", None), ("full", "1")):
        if env is None:
            monkeypatch.delenv("CB_NO_SYM", raising=False)
        else:
            monkeypatch.setenv("CB_NO_SYM", env)
        with CherryBank(times, C) as bank:
            runs[name] = bank.train_siterm(th0, Th0, E, lr=0.1)
    a, b = runs["sym"], runs["full"]
    assert np.allclose(a["loss_per_epoch_per_site"], b["loss_per_epoch_per_site"], rtol=1e-12, atol=0)
    assert relerr(a["res"], b["res"]) < 1e-11
    ref = orc.siterm_train(C, times, E, initialization=Q0)
    assert np.allclose(a["loss_per_epoch_per_site"], ref["loss_per_epoch_per_site"], rtol=1e-8, atol=0)
    for l in range(L):
        assert relerr(a["res"][l], ref["res"][l]) < 1e-6, l


def test_jtt_ipw_statistics_kernel_and_stage_on_the_reference_goldens(tmp_path):
    """SURVEY 8f #2 on the device (round 4): `cb_jtt_ipw_stats` -- F = sum_b sym(C_b), R = sum_b sym(C_b) / t_b in one
    streaming pass -- against numpy on ragged sizes (1 .. 400 states: edge tiles, diagonal tiles, bucket chunks), float64
    and integer counts (host and device pointers), symmetrised or not; then the stage function `jtt_ipw` itself on the
    reference's own goldens (tests/estimation_tests/jtt_ipw_test.py:12-74) and its `max_time`; the closed form on top is
    checked against the oracle's tensor form (1e-12)."""
    import cherryml_amd
    from cherryml_amd.estimation import jtt_ipw_from_arrays, jtt_ipw_statistics
    from cherryml_amd.io import read_rate_matrix, write_count_matrices
    from oracle import ratelearn_oracle as orc
    rng = np.random.default_rng(11)
    for S, B in ((1, 1), (3, 2), (16, 16), (17, 33), (20, 129), (37, 5), (400, 9)):
        t = np.sort(rng.uniform(1e-3, 4.0, B))
        Ci = rng.poisson(1.5, size=(B, S, S)).astype(np.uint64)
        Ci[::3] = 0
        unit = 0.25
        Cf = Ci.astype(np.float64) * unit
        for sym in (True, False):
            X = 0.5 * (Cf + Cf.transpose(0, 2, 1)) if sym else Cf
            wantF, wantR = X.sum(0), (X / t[:, None, None]).sum(0)
            F, R = jtt_ipw_statistics(t, Cf, 1.0, sym)                         # float64 counts, host pointers
            assert np.allclose(F, wantF, rtol=1e-13, atol=0) and np.allclose(R, wantR, rtol=1e-12, atol=0)
            F2, R2 = jtt_ipw_statistics(t, Ci, unit, sym)                      # integer histogram, host pointers
            assert np.array_equal(F2, F) and np.allclose(R2, R, rtol=1e-13, atol=0)
            d = torch.from_numpy(Ci.astype(np.int64)).cuda()                   # the resident histogram
            F3, R3 = jtt_ipw_statistics(t, d, unit, sym)
            assert np.array_equal(F3, F2) and np.array_equal(R3, R2)
            F4, R4 = jtt_ipw_statistics(t, torch.from_numpy(Cf).cuda(), 1.0, sym)
            assert np.array_equal(F4, F) and np.array_equal(R4, R)
        if S > 1:
            mask = (rng.random((S, S)) < 0.7).astype(np.float64)
            mask = np.maximum(mask, mask.T)
            np.fill_diagonal(mask, 1.0)
            for m in (None, mask):
                for ipw in (True, False):
                    got = jtt_ipw_from_arrays(t, Cf + 1.0, m, use_ipw=ipw)
                    assert np.allclose(got, orc.jtt_ipw(t, Cf + 1.0, m, use_ipw=ipw), rtol=1e-12, atol=1e-15)
            cut = float(t[B // 2])
            got = jtt_ipw_from_arrays(t, Cf + 1.0, None, max_time=cut)
            assert np.allclose(got, orc.jtt_ipw(t[t <= cut], (Cf + 1.0)[t <= cut]), rtol=1e-12, atol=1e-15)
    g = load_golden("jtt_ipw_toy.npz")
    states = list("ABC")
    cpath = str(tmp_path / "c.txt")
    write_count_matrices([(float(t), pd.DataFrame(C, index=states, columns=states)) for t, C in zip(g["t"], g["C"])], cpath)
    mpath = str(tmp_path / "m.txt")
    pd.DataFrame(g["mask"], index=states, columns=states).to_csv(mpath, sep=" ")
    for key, mp, ipw in [("Q1_JTT_IPW_on_toy_matrix", None, True), ("Q1_JTT_IPW_on_toy_matrix_mask", mpath, True),
                         ("Q1_JTT_on_toy_matrix", None, False), ("Q1_JTT_on_toy_matrix_mask", mpath, False)]:
        out = str(tmp_path / key)
        os.makedirs(out)
        cherryml_amd.jtt_ipw(count_matrices_path=cpath, mask_path=mp, use_ipw=ipw, output_rate_matrix_dir=out)
        got = read_rate_matrix(os.path.join(out, "result.txt")).to_numpy()
        np.testing.assert_almost_equal(got, g[key], decimal=7)
        np.testing.assert_almost_equal(got, orc.jtt_ipw(g["t"], g["C"], g["mask"].astype(float) if mp else None, use_ipw=ipw),
                                       decimal=12)


@pytest.mark.parametrize("case", ["toy3", "s20"])
@pytest.mark.parametrize("mode", ["default", "pande", "stationary", "stationary_reversible"])
def test_other_parameterisations_match_the_reference(mode, case):
    """VERDICT r3 (parity margin a): the four non-default parameterisations (rate.py:98-128, 190-218) were only tested to
    "run and descend".  Goldens from the reference itself in float64 (tests/golden/make_golden_modes.py): one evaluation --
    Q 1e-14, loss 1e-12, every parameter gradient 1e-9 -- and 30 epochs of train_quantization (Adam, lr 0.05): loss curve
    1e-8, Q_best / Q_last 1e-6.  The 20-state case carries the reference's own NON-symmetric random mask, so every mode goes
    through the general (scaling-and-squaring) kernels there; "stationary_reversible" without a mask takes the spectral
    path."""
    from cherryml_amd import CherryBank, RateMatrix, train_quantization
    from cherryml_amd._autograd import bank_loss
    from torch.utils.data import TensorDataset
    z = load_golden("modes.npz")
    k = f"{mode}_{case}"
    t, C, mask = z[f"{case}_t"], z[f"{case}_C"], z[f"{case}_mask"]
    S = C.shape[-1]

    def make():
        m = RateMatrix(num_states=S, mode=mode, mask=torch.tensor(mask), pi=torch.ones(S, dtype=torch.float64) / S,
                       pi_requires_grad=True)
        with torch.no_grad():
            m.upper_diag.copy_(torch.tensor(z[f"{k}_upper"]))
            if hasattr(m, "lower_diag"):
                m.lower_diag.copy_(torch.tensor(z[f"{k}_lower"]))
            m._pi.copy_(torch.tensor(z[f"{k}_log_pi"]))
        return m.to("cuda")

    mod = make()
    with CherryBank(t, C) as bank:
        Q = mod()
        Q.retain_grad()
        pi = mod.stationary() if mod.is_reversible() else None
        loss = bank_loss(Q.unsqueeze(0), None if pi is None else pi.unsqueeze(0), bank, True).sum()
        loss.backward()
    assert relerr(Q.detach().cpu().numpy(), z[f"{k}_Q"]) < 1e-14
    assert abs(float(loss) - float(z[f"{k}_loss"])) <= 1e-12 * abs(float(z[f"{k}_loss"]))
    assert relerr(Q.grad.cpu().numpy(), z[f"{k}_dQ"]) < 1e-9
    assert relerr(mod.upper_diag.grad.cpu().numpy(), z[f"{k}_d_upper"]) < 1e-9
    if hasattr(mod, "lower_diag"):
        assert relerr(mod.lower_diag.grad.cpu().numpy(), z[f"{k}_d_lower"]) < 1e-9
    if mode != "default":
        assert relerr(mod._pi.grad.cpu().numpy(), z[f"{k}_d_log_pi"]) < 1e-9
    mod = make()
    opt = torch.optim.Adam(mod.parameters(), lr=0.05)
    df, Qd = train_quantization(mod, TensorDataset(torch.tensor(t), torch.tensor(C)), num_epochs=30, optimizer=opt)
    assert np.allclose(df.loss.to_numpy(), z[f"{k}_traj_loss"], rtol=1e-8, atol=0)
    assert relerr(Qd["Q_best"], z[f"{k}_Q_best"]) < 1e-6 and relerr(Qd["Q_last"], z[f"{k}_Q_last"]) < 1e-6


def test_non_finite_iterate_gives_nan_loss_and_never_becomes_the_best(tmp_path):
    """the table logarithm of the small-state trainers read only exponent and mantissa bits, so a non-finite P came
    out as a large FINITE loss, which the best-iterate comparison (`loss < best`) could record.  A plain gradient step with an
    absurd learning rate throws the parameters to +-inf after the first epoch: every later loss must be NaN (as
    torch.matrix_exp's would be), and Q_best must stay the first iterate."""
    from cherryml_amd import CherryBank
    g = load_golden("eval_s20_symmask.npz")
    with CherryBank(g["t"], g["C"]) as bank:
        ok = bank.train_pande_reversible(g["upper_diag"], g["log_pi"], mask=g["mask"], num_epochs=1, lr=0.1)
        r = bank.train_pande_reversible(g["upper_diag"], g["log_pi"], mask=g["mask"], num_epochs=6, lr=1e300, do_adam=False)
    assert np.isfinite(r["loss"][0]) and r["loss"][0] == ok["loss"][0]
    assert np.all(np.isnan(r["loss"][1:])), r["loss"]
    assert np.array_equal(r["Q_best"], ok["Q_best"])      # the iterate of epoch 0, not a later non-finite one
