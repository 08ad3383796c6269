#!/usr/bin/env python3
"""End-to-end golden on the reference's own demo_data (BASELINE.json configs 1 and 3, scaled to three
real families): produced by RUNNING THE REFERENCE pipelines

    lg_end_to_end_with_cherryml_optimizer          (estimation_end_to_end/_cherry.py:209)
    coevolution_end_to_end_with_cherryml_optimizer (estimation_end_to_end/_cherry.py:449)

with the trees / site rates of demo_data (no tree estimation), the Python counting implementation,
the CPU optimiser.  Written to tests/golden/demo_e2e.npz:

  inputs   the three families' demo_data files as text (msa, tree, site rates, contact map) --
           data files the reference ships -- and the co-evolution mask structure
  LG       lg_counts [129,20,20], lg_init (JTT-IPW), lg_learned_f32 (the pipeline's result.txt,
           float32 expm as is), and the float64 recipe on the same counts: lg_loss_f64, lg_Q_best_f64
  co-evo   co_counts (sparse), co_init (masked JTT-IPW), co_learned_f32 (3 epochs), co_loss_f64,
           co_Q_best_f64 (3 epochs)

Usage: python tests/golden/make_golden_demo_e2e.py   (build container only; ~10 minutes)"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import REF, _count_arrays, _prepare_scratch, _traj_reference_f64  # noqa: E402

FAMILIES = ["1a92_1_A", "1a4p_1_A", "1a64_1_A"]
LG_EPOCHS, CO_EPOCHS = 60, 3


def main():
    _prepare_scratch()
    import torch
    from cherryml import caching
    from cherryml.estimation._ratelearn.rate import RateMatrix
    from cherryml.estimation._ratelearn.trainer import train_quantization
    from cherryml.estimation_end_to_end import (coevolution_end_to_end_with_cherryml_optimizer,
                                                lg_end_to_end_with_cherryml_optimizer)
    from cherryml.io import read_count_matrices, read_mask_matrix, read_rate_matrix
    demo = os.path.join(REF, "demo_data")
    out = {"families": np.array(FAMILIES)}
    for kind, sub in (("msa", "msas"), ("tree", "trees"), ("site_rates", "site_rates"), ("contact_map", "contact_maps")):
        out[f"text_{kind}"] = np.array([open(os.path.join(demo, sub, f + ".txt")).read() for f in FAMILIES])
    torch.set_num_threads(8)
    with tempfile.TemporaryDirectory() as cache:
        caching.set_cache_dir(cache)
        common = dict(tree_estimator=None, initial_tree_estimator_rate_matrix_path=None,
                      use_cpp_counting_implementation=False, optimizer_device="cpu", tree_dir=os.path.join(demo, "trees"),
                      num_processes_tree_estimation=1, num_processes_counting=1, num_processes_optimization=1)
        # ---------------------------------------------------------------- LG (config 1)
        r = lg_end_to_end_with_cherryml_optimizer(msa_dir=os.path.join(demo, "msas"), families=FAMILIES,
                                                  num_epochs=LG_EPOCHS, site_rates_dir=os.path.join(demo, "site_rates"),
                                                  **common)
        t, C = _count_arrays(read_count_matrices(os.path.join(r["count_matrices_dir_0"], "result.txt")))
        init = read_rate_matrix(os.path.join(r["jtt_ipw_dir_0"], "result.txt")).to_numpy()
        learned = read_rate_matrix(r["learned_rate_matrix_path"]).to_numpy()
        _, o64 = _traj_reference_f64(torch, RateMatrix, train_quantization, t, C, np.ones((20, 20)), init, LG_EPOCHS)
        out.update(lg_t=t, lg_counts=C, lg_init=init, lg_learned_f32=learned, lg_loss_f64=o64["loss"],
                   lg_Q_best_f64=o64["Q_best"], lg_epochs=np.int64(LG_EPOCHS),
                   quantization_points=np.array(r["quantization_points"]))
        print("LG: sum C", C.sum(), "loss", o64["loss"][0], "->", o64["loss"][-1])
        # ---------------------------------------------------------------- co-evolution (config 3)
        mask_path = os.path.join("data", "mask_matrices", "aa_coevolution_mask.txt")
        r = coevolution_end_to_end_with_cherryml_optimizer(
            msa_dir=os.path.join(demo, "msas"), contact_map_dir=os.path.join(demo, "contact_maps"),
            minimum_distance_for_nontrivial_contact=7, coevolution_mask_path=mask_path, families=FAMILIES,
            num_epochs=CO_EPOCHS, **common)
        t, C = _count_arrays(read_count_matrices(os.path.join(r["count_matrices_dir_0"], "result.txt")))
        init = read_rate_matrix(os.path.join(r["jtt_ipw_dir_0"], "result.txt")).to_numpy()
        learned = read_rate_matrix(r["learned_rate_matrix_path"]).to_numpy()
        mask = read_mask_matrix(mask_path).to_numpy().astype(np.float64)
        _, o64 = _traj_reference_f64(torch, RateMatrix, train_quantization, t, C, mask, init, CO_EPOCHS)
        nz = np.argwhere(C != 0).astype(np.int32)
        out.update(co_t=t, co_counts_nz=nz, co_counts_val=C[C != 0], co_init=init, co_learned_f32=learned,
                   co_loss_f64=o64["loss"], co_Q_best_f64=o64["Q_best"], co_epochs=np.int64(CO_EPOCHS),
                   co_mask_packed=np.packbits(mask.astype(bool)))
        print("co-evolution: sum C", C.sum(), "nonzeros", len(nz), "non-empty buckets",
              int((C.reshape(len(t), -1).sum(1) > 0).sum()), "loss", o64["loss"])
        caching.set_cache_dir(None) if hasattr(caching, "set_cache_dir") else None
    np.savez_compressed(os.path.join(HERE, "demo_e2e.npz"), **out)
    print("wrote", os.path.join(HERE, "demo_e2e.npz"))


if __name__ == "__main__":
    main()
