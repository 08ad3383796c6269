"""`cherryml_public_api` (reference cherryml/_cherryml_public_api.py:36-251, what `python -m cherryml`
calls): learn an LG-type 20x20 matrix or the 400x400 co-evolution matrix from MSAs.  Same arguments; the
stages behind it are this package's: FastCherries (`tree_estimator_name="FastCherries"`; seeded pairing on
the host, branch lengths / site rates on the GPU) or trees handed over in `tree_dir` (+ `site_rates_dir`),
GPU counting, JTT-IPW, and the optimiser of the hot path.  FastTree / PhyML are external programs the
reference shells out to; they are not built here (give `tree_dir`, or use FastCherries).  The optimiser
runs on the MI355X whatever `optimizer_device` says ("cpu", the reference's default, or "cuda": cherryml_amd/_device.py)."""
import os
import tempfile
from functools import partial
from typing import List, Optional

from . import caching
from .estimation_end_to_end import (coevolution_end_to_end_with_cherryml_optimizer,
                                    lg_end_to_end_with_cherryml_optimizer)
from .io import read_rate_matrix, write_rate_matrix
from .phylogeny_estimation import fast_cherries


def _families_of(msa_dir: str) -> List[str]:
    """utils.get_families (utils.py:80-95): sorted stems of the *.txt files."""
    return sorted(f[:-4] for f in os.listdir(msa_dir) if f.endswith(".txt"))


def cherryml_public_api(
    output_path: str,
    model_name: str,
    msa_dir: str,
    contact_map_dir: Optional[str] = None,
    tree_dir: Optional[str] = None,
    site_rates_dir: Optional[str] = None,
    cache_dir: Optional[str] = None,
    num_processes_tree_estimation: int = 32,
    num_processes_counting: int = 8,
    num_processes_optimization: int = 2,
    num_rate_categories: int = 20,
    initial_tree_estimator_rate_matrix_path: Optional[str] = None,
    num_iterations: int = 1,
    quantization_grid_center: float = 0.03,
    quantization_grid_step: float = 1.1,
    quantization_grid_num_steps: int = 64,
    use_cpp_counting_implementation: bool = True,
    optimizer_device: str = "cpu",
    learning_rate: float = 1e-1,
    num_epochs: int = 500,
    minimum_distance_for_nontrivial_contact: int = 7,
    do_adam: bool = True,
    cherryml_type: str = "cherry++",
    cpp_counting_command_line_prefix: str = "",
    cpp_counting_command_line_suffix: str = "",
    optimizer_initialization: str = "jtt-ipw",
    sites_subset_dir: Optional[str] = None,
    coevolution_mask_path: Optional[str] = None,
    use_maximal_matching: bool = True,
    families: Optional[List[str]] = None,
    tree_estimator_name: str = "FastCherries",
) -> str:
    """Writes the learned rate matrix to `output_path` and returns the profiling string of the pipeline.
    Differences from the reference: `tree_estimator_name` defaults to "FastCherries" (the reference's
    default "FastTree" is an external program); `initial_tree_estimator_rate_matrix_path` (the reference
    defaults to its bundled LG file) must be given when trees are estimated; `optimizer_device` defaults
    to "cuda"."""
    if model_name not in ["LG", "co-evolution"]:
        raise ValueError('model_name should be either "LG" or "co-evolution".')
    keep = None
    if cache_dir is None:
        keep = tempfile.TemporaryDirectory()
        cache_dir = keep.name
    caching.set_cache_dir(cache_dir)
    if families is None:
        families = _families_of(msa_dir)
    if tree_estimator_name == "FastCherries":
        tree_estimator = partial(fast_cherries, max_iters=50, num_rate_categories=num_rate_categories, verbose=False)
    elif tree_estimator_name in ("FastTree", "PhyML"):
        tree_estimator = None    # only usable with tree_dir (the pipelines raise NotImplementedError otherwise)
    else:
        raise ValueError(f"Unknown tree_estimator_name: {tree_estimator_name}")
    if tree_dir is None and initial_tree_estimator_rate_matrix_path is None:
        raise ValueError("initial_tree_estimator_rate_matrix_path is required when trees are to be estimated "
                         "(the reference defaults to its bundled LG matrix, data/rate_matrices/lg.txt)")
    common = dict(
        msa_dir=msa_dir, families=families, tree_estimator=tree_estimator,
        initial_tree_estimator_rate_matrix_path=initial_tree_estimator_rate_matrix_path,
        quantization_grid_center=quantization_grid_center, quantization_grid_step=quantization_grid_step,
        quantization_grid_num_steps=quantization_grid_num_steps,
        use_cpp_counting_implementation=use_cpp_counting_implementation, optimizer_device=optimizer_device,
        learning_rate=learning_rate, num_epochs=num_epochs, do_adam=do_adam, edge_or_cherry=cherryml_type,
        cpp_counting_command_line_prefix=cpp_counting_command_line_prefix,
        cpp_counting_command_line_suffix=cpp_counting_command_line_suffix,
        num_processes_tree_estimation=num_processes_tree_estimation,
        num_processes_counting=num_processes_counting, num_processes_optimization=num_processes_optimization,
        optimizer_initialization=optimizer_initialization, tree_dir=tree_dir)
    try:
        if model_name == "LG":
            outputs = lg_end_to_end_with_cherryml_optimizer(
                num_iterations=num_iterations, sites_subset_dir=sites_subset_dir, site_rates_dir=site_rates_dir,
                **common)
        else:
            if num_iterations > 1:
                raise ValueError("Iteration is not used for learning a coevolution model. "
                                 f"You provided: num_iterations={num_iterations}. Set this argument to 1 and retry.")
            outputs = coevolution_end_to_end_with_cherryml_optimizer(
                contact_map_dir=contact_map_dir,
                minimum_distance_for_nontrivial_contact=minimum_distance_for_nontrivial_contact,
                coevolution_mask_path=coevolution_mask_path, use_maximal_matching=use_maximal_matching, **common)
        learned = read_rate_matrix(outputs["learned_rate_matrix_path"])
        write_rate_matrix(learned.to_numpy(), list(learned.columns), output_path)
        return outputs.get("profiling_str", "")
    finally:
        if keep is not None:
            caching.set_cache_dir(None)
            keep.cleanup()
