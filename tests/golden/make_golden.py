#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

This script only works in the build container (it needs /root/reference); its
outputs (small .npz files = inputs + the reference's outputs) are committed and
are what travels to the GPU box.  Nothing of the reference's source is copied
into the repo: the reference package is copied to a scratch dir under /tmp,
its Cython helper is built there, and a few packages that the hot path never
touches (ete3, seaborn, wget, parameterized, biotite) are stubbed so that
`import cherryml` succeeds (recipe: SURVEY.md section 8c).

Fixtures written (all arrays float64 unless noted):

  eval_<case>.npz   single evaluation of the epoch body
                    (cherryml/estimation/_ratelearn/trainer.py:156-186):
                    inputs t[B], C[B,S,S], mask[S,S], params upper_diag, log_pi
                    outputs, for the as-is float32 path (`*_f32`) and the
                    float64 recipe (`*_f64`): Q, loss, dL/dQ, dL/dupper_diag,
                    dL/dlog_pi.
  traj_<case>.npz   whole optimisation (`quantized_transitions_mle` as is, and
                    the f64 recipe through `train_quantization`): loss curve,
                    Q_best, Q_last, Q_1, Q_2.
  siterm_<case>.npz `quantized_transitions_mle_vectorized_over_sites`
                    (cherryml/_siterm/_cherryml_vectorized.py:107): counts,
                    times, initialisation -> res, loss_per_epoch_per_site.
  jtt_ipw_toy.npz   the reference tests' own golden files for jtt_ipw
                    (tests/test_input_data/Q1_JTT*), as data.
  data_lg.npz       the LG rate matrix (data/rate_matrices/lg.txt) and the
                    co-evolution mask structure check, as data.

Usage:  python tests/golden/make_golden.py
"""
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _prepare_scratch() -> str:
    scratch = os.path.join(tempfile.gettempdir(), "cherryml_ref_scratch")
    work = os.path.join(scratch, "work")
    stubs = os.path.join(scratch, "stubs")
    if not os.path.exists(os.path.join(work, "cherryml", "_siterm")) or not any(
        f.startswith("fast_site_rates.") and f.endswith(".so")
        for f in os.listdir(os.path.join(work, "cherryml", "_siterm"))
    ):
        shutil.rmtree(scratch, ignore_errors=True)
        os.makedirs(work)
        for item in ["cherryml", "setup.py", "tests", "data"]:
            src = os.path.join(REF, item)
            dst = os.path.join(work, item)
            if os.path.isdir(src):
                shutil.copytree(src, dst)
            else:
                shutil.copy(src, dst)
        subprocess.check_call(
            [sys.executable, "setup.py", "build_ext", "--inplace"],
            cwd=work,
            stdout=subprocess.DEVNULL,
            stderr=subprocess.DEVNULL,
        )
    os.makedirs(stubs, exist_ok=True)
    with open(os.path.join(stubs, "ete3.py"), "w") as f:
        f.write("class Tree:\n    pass\n")
    for name in ["seaborn", "wget"]:
        open(os.path.join(stubs, name + ".py"), "w").close()
    with open(os.path.join(stubs, "parameterized.py"), "w") as f:
        f.write(
            "class parameterized:\n"
            "    @staticmethod\n"
            "    def expand(params):\n"
            "        def deco(fn):\n"
            "            return fn\n"
            "        return deco\n"
        )
    os.makedirs(os.path.join(stubs, "biotite"), exist_ok=True)
    open(os.path.join(stubs, "biotite", "__init__.py"), "w").close()
    sys.path.insert(0, work)
    sys.path.insert(0, stubs)
    os.chdir(work)  # the reference's data paths are cwd-relative
    return work


def _count_arrays(count_matrices):
    t = np.array([q for q, _ in count_matrices], dtype=np.float64)
    C = np.stack([m.to_numpy() for _, m in count_matrices]).astype(np.float64)
    return t, C


def _eval_reference(torch, RateMatrix, t, C, mask, init, dtype, seed=0):
    """One epoch body of trainer.py:156-186, on the reference's RateMatrix."""
    S = C.shape[-1]
    old = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        torch.manual_seed(seed)
        pi = torch.ones(S, dtype=dtype) / S
        module = RateMatrix(
            num_states=S,
            mode="pande_reversible",
            pi=pi,
            pi_requires_grad=True,
            initialization=init,
            mask=torch.tensor(mask, dtype=torch.float),
        )
        if dtype == torch.float64:
            module = module.double()
        params = dict(
            upper_diag=module.upper_diag.detach().numpy().astype(np.float64),
            log_pi=module._pi.detach().numpy().astype(np.float64),
        )
        Q = module()
        Q.retain_grad()
        tt = torch.tensor(t, dtype=dtype)
        CC = torch.tensor(C)  # float64, as ratelearner.py:150
        mats = torch.log(torch.matrix_exp(tt[:, None, None] * Q)) * CC
        loss = -mats.sum() / CC.sum()
        loss.backward()
        out = dict(
            Q=Q.detach().numpy().astype(np.float64),
            loss=float(loss.item()),
            dQ=Q.grad.numpy().astype(np.float64),
            d_upper=module.upper_diag.grad.numpy().astype(np.float64),
            d_log_pi=module._pi.grad.numpy().astype(np.float64),
        )
    finally:
        torch.set_default_dtype(old)
    return params, out


def _traj_reference_f64(torch, RateMatrix, train_quantization, t, C, mask, init,
                        num_epochs, upper_diag=None, log_pi=None):
    """The f64 oracle recipe of SURVEY.md 8c: the reference's own
    `train_quantization` on a float64 module."""
    from torch.utils.data import TensorDataset

    S = C.shape[-1]
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        torch.manual_seed(0)
        module = RateMatrix(
            num_states=S,
            mode="pande_reversible",
            pi=torch.ones(S, dtype=torch.float64) / S,
            pi_requires_grad=True,
            initialization=init,
            mask=torch.tensor(mask, dtype=torch.float),
        ).double()
        if upper_diag is not None:
            module.upper_diag.data.copy_(torch.tensor(upper_diag))
            module._pi.data.copy_(torch.tensor(log_pi))
        params = dict(
            upper_diag=module.upper_diag.detach().numpy().copy(),
            log_pi=module._pi.detach().numpy().copy(),
        )
        ds = TensorDataset(torch.tensor(t), torch.tensor(C))
        opt = torch.optim.Adam(module.parameters(), lr=0.1)
        df, Qd = train_quantization(
            rate_module=module,
            quantized_dataset=ds,
            num_epochs=num_epochs,
            Q_true=None,
            optimizer=opt,
            loss_normalization=True,
            return_best_iter=True,
        )
    finally:
        torch.set_default_dtype(old)
    out = dict(
        loss=df.loss.to_numpy().astype(np.float64),
        Q_best=np.asarray(Qd["Q_best"], dtype=np.float64),
        Q_last=np.asarray(Qd["Q_last"], dtype=np.float64),
        Q_1=np.asarray(Qd["Q_1"], dtype=np.float64),
        Q_2=np.asarray(Qd["Q_2"], dtype=np.float64),
    )
    return params, out


def _traj_reference_asis(quantized_transitions_mle, read_rate_matrix,
                         count_path, init_path, mask_path, num_epochs):
    """`quantized_transitions_mle` exactly as a user calls it (float32 expm)."""
    import pandas as pd

    with tempfile.TemporaryDirectory() as out_dir:
        quantized_transitions_mle(
            count_matrices_path=count_path,
            initialization_path=init_path,
            mask_path=mask_path,
            output_rate_matrix_dir=out_dir,
            stationary_distribution_path=None,
            rate_matrix_parameterization="pande_reversible",
            device="cpu",
            learning_rate=1e-1,
            num_epochs=num_epochs,
            do_adam=True,
        )
        files = sorted(os.listdir(out_dir))
        out = dict(
            loss=pd.read_csv(os.path.join(out_dir, "df_res.txt")).loss.to_numpy(),
            Q_best=read_rate_matrix(os.path.join(out_dir, "Q_best.txt")).to_numpy(),
            Q_last=read_rate_matrix(os.path.join(out_dir, "Q_last.txt")).to_numpy(),
            result=read_rate_matrix(os.path.join(out_dir, "result.txt")).to_numpy(),
            Q_1=read_rate_matrix(os.path.join(out_dir, "Q_1.txt")).to_numpy(),
            Q_2=read_rate_matrix(os.path.join(out_dir, "Q_2.txt")).to_numpy(),
        )
    return out, files


def main():
    work = _prepare_scratch()
    import torch

    import cherryml  # noqa: F401
    from cherryml.estimation import quantized_transitions_mle
    from cherryml.estimation._ratelearn.rate import RateMatrix
    from cherryml.estimation._ratelearn.trainer import train_quantization
    from cherryml.io import (
        read_count_matrices,
        read_mask_matrix,
        read_rate_matrix,
        write_count_matrices,
        write_rate_matrix,
    )
    from cherryml._siterm._cherryml_vectorized import (
        quantized_transitions_mle_vectorized_over_sites,
    )

    torch.set_num_threads(4)
    tid = "tests/test_input_data"
    save = lambda name, **kw: np.savez_compressed(os.path.join(HERE, name), **kw)

    # ------------------------------------------------------------------ data
    lg = read_rate_matrix("data/rate_matrices/lg.txt")
    wag = read_rate_matrix("data/rate_matrices/wag.txt").to_numpy()
    equ = read_rate_matrix("data/rate_matrices/equ.txt").to_numpy()
    comask = read_mask_matrix("data/mask_matrices/aa_coevolution_mask.txt")
    save(
        "data_lg.npz",
        lg=lg.to_numpy(),
        wag=wag,
        equ=equ,
        states=np.array(list(lg.index)),
        coevolution_mask_rowsum=comask.to_numpy().sum(1),
        coevolution_mask_packed=np.packbits(comask.to_numpy().astype(np.uint8)),
        coevolution_states=np.array(list(comask.index)),
    )

    # ------------------------------------------------------ single evaluations
    cases = {
        "toy3_init": (f"{tid}/matrices_toy.txt",
                      f"{tid}/3x3_pande_reversible_initialization.txt", None),
        "toy3_mask": (f"{tid}/matrices_toy.txt",
                      f"{tid}/3x3_pande_reversible_initialization_mask.txt",
                      f"{tid}/3x3_mask.txt"),
        "s20_mask": (f"{tid}/matrices_small/matrices_by_quantized_branch_length.txt",
                     None, f"{tid}/20x20_random_mask.txt"),
        "s400_mask": (f"{tid}/co_matrices_small/matrices_by_quantized_branch_length.txt",
                      None, f"{tid}/synthetic_rate_matrices/mask_Q2.txt"),
    }
    # the reference's 20x20 random mask is NOT symmetric (=> non-reversible Q);
    # add a symmetrised copy so the reversible path has a 20-state masked case.
    symdir = tempfile.mkdtemp()
    m20 = read_mask_matrix(f"{tid}/20x20_random_mask.txt")
    (m20 * m20.T).to_csv(os.path.join(symdir, "mask_sym.txt"), sep=" ")
    cases["s20_symmask"] = (cases["s20_mask"][0], None, os.path.join(symdir, "mask_sym.txt"))
    for name, (cpath, ipath, mpath) in cases.items():
        t, C = _count_arrays(read_count_matrices(cpath))
        S = C.shape[-1]
        mask = (read_mask_matrix(mpath).to_numpy().astype(np.float64)
                if mpath else np.ones((S, S)))
        init = read_rate_matrix(ipath).to_numpy() if ipath else None
        p32, o32 = _eval_reference(torch, RateMatrix, t, C, mask, init, torch.float32)
        # f64 recipe evaluated at the SAME parameters (the f32 ones, widened),
        # so that both outputs belong to one input.
        old = torch.get_default_dtype()
        torch.set_default_dtype(torch.float64)
        module = RateMatrix(
            num_states=S, mode="pande_reversible",
            pi=torch.ones(S, dtype=torch.float64) / S, pi_requires_grad=True,
            initialization=None, mask=torch.tensor(mask, dtype=torch.float),
        ).double()
        module.upper_diag.data.copy_(torch.tensor(p32["upper_diag"]))
        module._pi.data.copy_(torch.tensor(p32["log_pi"]))
        Q = module()
        Q.retain_grad()
        CC = torch.tensor(C)
        loss = -(torch.log(torch.matrix_exp(torch.tensor(t)[:, None, None] * Q)) * CC).sum() / CC.sum()
        loss.backward()
        o64 = dict(Q=Q.detach().numpy(), loss=float(loss.item()), dQ=Q.grad.numpy(),
                   d_upper=module.upper_diag.grad.numpy(),
                   d_log_pi=module._pi.grad.numpy())
        torch.set_default_dtype(old)
        kw = dict(t=t, C=C, mask=mask, upper_diag=p32["upper_diag"], log_pi=p32["log_pi"])
        if init is not None:
            kw["init"] = init
        if name == "s400_mask":  # keep the fixture small: C is sparse
            nz = np.nonzero(C)
            kw["C_shape"] = np.array(C.shape)
            kw["C_idx"] = np.stack(nz).astype(np.int32)
            kw["C_val"] = C[nz]
            del kw["C"]
            kw["mask"] = np.packbits(mask.astype(np.uint8))
            kw["mask_shape"] = np.array(mask.shape)
            for o in (o32, o64):
                o["Q"] = o["Q"].astype(np.float64)
        for k, v in o32.items():
            kw[k + "_f32"] = v
        for k, v in o64.items():
            kw[k + "_f64"] = v
        save(f"eval_{name}.npz", **kw)
        print(f"eval_{name}: loss f32 {o32['loss']:.9f} f64 {o64['loss']:.12f}")

    # ------------------------------------------------------------ trajectories
    for name, epochs in [("toy3_init", 30), ("toy3_mask", 30), ("s20_mask", 50),
                         ("s20_symmask", 50), ("s400_mask", 3)]:
        cpath, ipath, mpath = cases[name]
        t, C = _count_arrays(read_count_matrices(cpath))
        S = C.shape[-1]
        mask = (read_mask_matrix(mpath).to_numpy().astype(np.float64)
                if mpath else np.ones((S, S)))
        init = read_rate_matrix(ipath).to_numpy() if ipath else None
        asis, files = _traj_reference_asis(
            quantized_transitions_mle, read_rate_matrix, cpath, ipath, mpath, epochs)
        ev = np.load(os.path.join(HERE, f"eval_{name}.npz"))
        p64, o64 = _traj_reference_f64(
            torch, RateMatrix, train_quantization, t, C, mask, init, epochs,
            upper_diag=None if init is not None else ev["upper_diag"],
            log_pi=None if init is not None else ev["log_pi"])
        kw = dict(num_epochs=epochs, files=np.array(files),
                  upper_diag0_f64=p64["upper_diag"], log_pi0_f64=p64["log_pi"])
        for k, v in asis.items():
            kw[k + "_f32"] = v
        for k, v in o64.items():
            kw[k + "_f64"] = v
        save(f"traj_{name}.npz", **kw)
        print(f"traj_{name}: f32 loss {asis['loss'][0]:.7f}->{asis['loss'][-1]:.7f}; "
              f"f64 {o64['loss'][0]:.10f}->{o64['loss'][-1]:.10f}; files={files}")

    # LG-shaped bank (config 2 generator of SURVEY.md 8d, scaled down):
    # C_b = w_b diag(pi) expm(t_b Q_LG), B = 129 grid, JTT-IPW-free init = 0.8*LG
    from cherryml.markov_chain import compute_stationary_distribution, matrix_exponential
    Qlg = lg.to_numpy()
    grid = np.array([float("%.8f" % (0.03 * 1.1 ** i)) for i in range(-64, 65)])
    rng = np.random.default_rng(0)
    lengths = rng.exponential(0.4, size=200000)
    lengths = lengths[(lengths >= grid[0]) & (lengths <= grid[-1])]
    idx = np.abs(np.log(lengths[:, None] / grid[None, :])).argmin(1)
    w = np.bincount(idx, minlength=129).astype(np.float64)
    pi_lg = compute_stationary_distribution(Qlg)
    P = matrix_exponential(exponents=grid, Q=Qlg, fact=None, reversible=False, device="cpu")
    Cb = w[:, None, None] * pi_lg[None, :, None] * P
    Cb = 0.5 * (Cb + Cb.transpose(0, 2, 1))
    init = 0.8 * Qlg
    with tempfile.TemporaryDirectory() as d:
        import pandas as pd
        states = list(lg.index)
        cpath = os.path.join(d, "counts.txt")
        ipath = os.path.join(d, "init.txt")
        write_count_matrices(
            [(float(grid[b]), pd.DataFrame(Cb[b], index=states, columns=states))
             for b in range(129)], cpath)
        write_rate_matrix(init, states, ipath)
        t, C = _count_arrays(read_count_matrices(cpath))
        asis, files = _traj_reference_asis(
            quantized_transitions_mle, read_rate_matrix, cpath, ipath, None, 100)
    p64, o64 = _traj_reference_f64(
        torch, RateMatrix, train_quantization, t, C, np.ones((20, 20)), init, 100)
    kw = dict(num_epochs=100, t=t, C=C, init=init,
              upper_diag0_f64=p64["upper_diag"], log_pi0_f64=p64["log_pi"])
    for k, v in asis.items():
        kw[k + "_f32"] = v
    for k, v in o64.items():
        kw[k + "_f64"] = v
    save("traj_lgbank.npz", **kw)
    print(f"traj_lgbank: f64 loss {o64['loss'][0]:.10f}->{o64['loss'][-1]:.10f}")

    # ------------------------------------------------------------------ SiteRM
    from cherryml._siterm._site_specific_rate_matrix import (
        get_synthetic_counts_DNA,
        get_synthetic_counts_amino_acids,
    )
    Qs, counts, times = get_synthetic_counts_DNA(L=8, B=11)
    times = np.array(times)
    r_init = quantized_transitions_mle_vectorized_over_sites(
        counts, times, num_epochs=20, initialization=0.7 * Qs)
    r_rand = quantized_transitions_mle_vectorized_over_sites(
        counts, times, num_epochs=20, initialization=None)
    save("siterm_dna.npz", Qs_true=Qs, counts=counts, times=times, init=0.7 * Qs,
         res_init=r_init["res"], lpe_init=r_init["loss_per_epoch"],
         lpeps_init=r_init["loss_per_epoch_per_site"],
         res_rand=r_rand["res"].astype(np.float64), lpe_rand=r_rand["loss_per_epoch"],
         lpeps_rand=r_rand["loss_per_epoch_per_site"])
    Qs, counts, times = get_synthetic_counts_amino_acids(L=4, B=9)
    r_init = quantized_transitions_mle_vectorized_over_sites(
        counts, times, num_epochs=10, initialization=0.9 * Qs)
    save("siterm_aa.npz", Qs_true=Qs, counts=counts, times=times, init=0.9 * Qs,
         res_init=r_init["res"], lpe_init=r_init["loss_per_epoch"],
         lpeps_init=r_init["loss_per_epoch_per_site"])
    print("siterm: dna init loss", r_init["loss_per_epoch"][:2])

    # ----------------------------------------------------------------- jtt_ipw
    t, C = _count_arrays(read_count_matrices(f"{tid}/matrices_toy.txt"))
    kw = dict(t=t, C=C, mask=read_mask_matrix(f"{tid}/3x3_mask.txt").to_numpy())
    for d in ["Q1_JTT-IPW_on_toy_matrix", "Q1_JTT-IPW_on_toy_matrix_mask",
              "Q1_JTT_on_toy_matrix", "Q1_JTT_on_toy_matrix_mask"]:
        kw[d.replace("-", "_")] = np.loadtxt(f"{tid}/{d}/learned_matrix.txt")  # as jtt_ipw_test.py:36 reads them
    save("jtt_ipw_toy.npz", **kw)
    print("done; scratch at", work)


if __name__ == "__main__":
    main()
