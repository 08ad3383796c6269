"""The two pipelines that call the hot path (reference:
cherryml/estimation_end_to_end/_cherry.py:209-447 LG, :449-584 co-evolution), with the same
keyword signatures and result dictionaries:

    trees (given, or from a caller-supplied estimator) -> count_(co_)transitions [GPU]
      -> jtt_ipw [host, closed form] -> quantized_transitions_mle [GPU]

Tree estimation itself (FastTree / PhyML / FastCherries wrappers) is outside the scope of
this package (out of scope): pass `tree_dir` (+ `site_rates_dir`), or any
callable with the reference's tree-estimator interface as `tree_estimator`.
Like the reference, the pipelines need a cache directory (`caching.set_cache_dir`): the stage
functions hand their output directories to each other through it.
"""
import os
from typing import Callable, Dict, List, Optional

import numpy as np

from .. import caching
from ..counting import _host, count_co_transitions, count_transitions
from ..estimation import jtt_ipw, quantized_transitions_mle

AMINO_ACIDS = list("ARNDCQEGHILKMFPSTWYV")  # cherryml/utils.py:7-28
CHERRYML_TYPE = "cherry++"


def _grid(center: float, step: float, num_steps: int) -> List[str]:
    """_cherry.py:267-272: points rounded through '%.8f'."""
    return ["%.8f" % (center * step ** i) for i in range(-num_steps, num_steps + 1)]


def _runtime(profiling_path: str) -> float:
    """third whitespace token of 'Total time: X seconds ...' (_cherry.py:148-155)"""
    with open(profiling_path) as f:
        return float(f.read().split()[2])


def _need_cache():
    if caching.get_cache_dir() is None:
        raise caching.CacheUsageError(
            "the end-to-end pipelines pass directories between stages through the cache: "
            "call cherryml_amd.caching.set_cache_dir(...) first (as with the reference)")


def _equ_matrix(states: List[str]) -> np.ndarray:
    n = len(states)
    Q = np.full((n, n), 1.0 / (n - 1))
    np.fill_diagonal(Q, -1.0)
    return Q


def lg_end_to_end_with_cherryml_optimizer(
    msa_dir: str,
    families: List[str],
    tree_estimator: Optional[Callable],
    initial_tree_estimator_rate_matrix_path: Optional[str],
    num_iterations: Optional[int] = 1,
    quantization_grid_center: float = 0.03,
    quantization_grid_step: float = 1.1,
    quantization_grid_num_steps: int = 64,
    use_cpp_counting_implementation: bool = True,
    optimizer_device: str = "cpu",
    learning_rate: float = 1e-1,
    num_epochs: int = 2000,
    do_adam: bool = True,
    edge_or_cherry: str = CHERRYML_TYPE,
    cpp_counting_command_line_prefix: str = "",
    cpp_counting_command_line_suffix: str = "",
    num_processes_tree_estimation: int = 8,
    num_processes_counting: int = 8,
    num_processes_optimization: int = 2,
    optimizer_initialization: str = "jtt-ipw",
    sites_subset_dir: Optional[str] = None,
    tree_dir: Optional[str] = None,
    site_rates_dir: Optional[str] = None,
    alphabet: List[str] = AMINO_ACIDS,
) -> Dict:
    _need_cache()
    if sites_subset_dir is not None:
        raise NotImplementedError("sites_subset_dir is not supported by this build")
    if (tree_dir is None) != (site_rates_dir is None):
        raise ValueError("tree_dir and site_rates_dir must be either both provided or none "
                         f"provided. You provided: tree_dir={tree_dir} ; site_rates_dir={site_rates_dir}")
    res: Dict = {}
    quantization_points = _grid(quantization_grid_center, quantization_grid_step,
                                quantization_grid_num_steps)
    res["quantization_points"] = quantization_points
    t_count = t_jtt = t_opt = 0.0
    current = initial_tree_estimator_rate_matrix_path
    for iteration in range(num_iterations):
        if iteration == 0 and tree_dir is not None:
            dirs = {"output_tree_dir": tree_dir, "output_site_rates_dir": site_rates_dir}
        elif tree_estimator is None:
            raise NotImplementedError(
                "tree estimation is out of scope here: provide tree_dir and site_rates_dir, or a "
                "tree_estimator callable (reference interface) for further iterations")
        else:
            dirs = tree_estimator(msa_dir=msa_dir, families=families, rate_matrix_path=current,
                                  num_processes=num_processes_tree_estimation)
        res[f"tree_estimator_output_dirs_{iteration}"] = dirs
        count_dir = count_transitions(
            tree_dir=dirs["output_tree_dir"], msa_dir=msa_dir,
            site_rates_dir=dirs["output_site_rates_dir"], families=families,
            amino_acids=alphabet[:], quantization_points=quantization_points,
            edge_or_cherry=edge_or_cherry, num_processes=num_processes_counting,
            use_cpp_implementation=use_cpp_counting_implementation,
            cpp_command_line_prefix=cpp_counting_command_line_prefix,
            cpp_command_line_suffix=cpp_counting_command_line_suffix)["output_count_matrices_dir"]
        res[f"count_matrices_dir_{iteration}"] = count_dir
        t_count += _runtime(os.path.join(count_dir, "profiling.txt"))
        jtt_dir = jtt_ipw(count_matrices_path=os.path.join(count_dir, "result.txt"), mask_path=None,
                          use_ipw=True, normalize=False)["output_rate_matrix_dir"]
        res[f"jtt_ipw_dir_{iteration}"] = jtt_dir
        t_jtt += _runtime(os.path.join(jtt_dir, "profiling.txt"))
        if optimizer_initialization == "jtt-ipw":
            init_path = os.path.join(jtt_dir, "result.txt")
        elif optimizer_initialization == "equ":
            from ..io import write_rate_matrix
            init_path = os.path.join(jtt_dir, "equ.txt")
            if not os.path.exists(init_path):
                write_rate_matrix(_equ_matrix(alphabet), alphabet, init_path)
        elif optimizer_initialization == "random":
            init_path = None
        else:
            raise ValueError(f"Unknown optimizer_initialization = {optimizer_initialization}")
        rate_dir = quantized_transitions_mle(
            count_matrices_path=os.path.join(count_dir, "result.txt"),
            initialization_path=init_path, mask_path=None, stationary_distribution_path=None,
            rate_matrix_parameterization="pande_reversible", device=optimizer_device,
            learning_rate=learning_rate, num_epochs=num_epochs, do_adam=do_adam,
            OMP_NUM_THREADS=num_processes_optimization,
            OPENBLAS_NUM_THREADS=num_processes_optimization)["output_rate_matrix_dir"]
        t_opt += _runtime(os.path.join(rate_dir, "profiling.txt"))
        res[f"rate_matrix_dir_{iteration}"] = rate_dir
        current = os.path.join(rate_dir, "result.txt")
    res["learned_rate_matrix_path"] = current
    res["time_tree_estimation"] = 0.0
    res["time_counting"], res["time_jtt_ipw"], res["time_optimization"] = t_count, t_jtt, t_opt
    res["total_cpu_time"] = t_count + t_jtt + t_opt
    res["profiling_str"] = (
        "CherryML runtimes:\n"
        f"time_tree_estimation (without parallelization): {res['time_tree_estimation']}\n"
        f"time_counting: {t_count}\ntime_jtt_ipw: {t_jtt}\ntime_optimization: {t_opt}\n"
        f"total_cpu_time: {res['total_cpu_time']}\n")
    return res


@caching.cached_computation(output_dirs=["o_contact_map_dir"], exclude_args=["num_processes"],
                            write_extra_log_files=True)
def create_maximal_matching_contact_map(
    i_contact_map_dir: str,
    families: List[str],
    minimum_distance_for_nontrivial_contact: int,
    num_processes: int,
    o_contact_map_dir: Optional[str] = None,
) -> None:
    """Replace each contact map by a maximal matching of its non-trivial contacts
    (reference: cherryml/evaluation/_maximal_matching.py:37-93).  The reference calls
    networkx.maximal_matching, which is greedy over the graph's edge iteration order; with
    nodes 0..n-1 inserted first and edges inserted in row-major (i < j) order that is
    row-major order again, restated here directly."""
    os.makedirs(o_contact_map_dir, exist_ok=True)
    mdnc = int(minimum_distance_for_nontrivial_contact)
    for fam in families:
        cm = _host.read_contact_map(os.path.join(i_contact_map_dir, fam + ".txt"))
        n = cm.shape[0]
        ii, jj = np.nonzero(cm == 1)
        keep = (ii < jj) & (jj - ii >= mdnc)
        matched = np.zeros(n, dtype=bool)
        out = np.zeros((n, n), dtype=np.int64)
        for u, v in zip(ii[keep], jj[keep]):
            if not matched[u] and not matched[v]:
                matched[u] = matched[v] = True
                out[u, v] = out[v, u] = 1
        with open(os.path.join(o_contact_map_dir, fam + ".txt"), "w") as f:
            f.write(f"{n} sites\n")
            np.savetxt(f, out, delimiter="", fmt="%i")


def coevolution_end_to_end_with_cherryml_optimizer(
    msa_dir: str,
    contact_map_dir: str,
    minimum_distance_for_nontrivial_contact: int,
    coevolution_mask_path: Optional[str],
    families: List[str],
    tree_estimator: Optional[Callable],
    initial_tree_estimator_rate_matrix_path: Optional[str],
    quantization_grid_center: float = 0.03,
    quantization_grid_step: float = 1.1,
    quantization_grid_num_steps: int = 64,
    use_cpp_counting_implementation: bool = True,
    optimizer_device: str = "cpu",
    learning_rate: float = 1e-1,
    num_epochs: int = 500,
    do_adam: bool = True,
    edge_or_cherry: str = CHERRYML_TYPE,
    cpp_counting_command_line_prefix: str = "",
    cpp_counting_command_line_suffix: str = "",
    num_processes_tree_estimation: int = 8,
    num_processes_counting: int = 8,
    num_processes_optimization: int = 8,
    optimizer_initialization: str = "jtt-ipw",
    use_maximal_matching: bool = True,
    tree_dir: Optional[str] = None,
    alphabet: List[str] = AMINO_ACIDS,
) -> Dict:
    _need_cache()
    res: Dict = {}
    quantization_points = _grid(quantization_grid_center, quantization_grid_step,
                                quantization_grid_num_steps)
    res["quantization_points"] = quantization_points
    if tree_dir is not None:
        dirs = {"output_tree_dir": tree_dir}
    elif tree_estimator is None:
        raise NotImplementedError("tree estimation is out of scope here: provide tree_dir or a "
                                  "tree_estimator callable (reference interface)")
    else:
        dirs = tree_estimator(msa_dir=msa_dir, families=families,
                              rate_matrix_path=initial_tree_estimator_rate_matrix_path,
                              num_processes=num_processes_tree_estimation)
    res["tree_estimator_output_dirs_0"] = dirs
    mdnc = minimum_distance_for_nontrivial_contact
    if use_maximal_matching:
        contact_map_dir = create_maximal_matching_contact_map(
            i_contact_map_dir=contact_map_dir, families=families,
            minimum_distance_for_nontrivial_contact=mdnc,
            num_processes=num_processes_counting)["o_contact_map_dir"]
    count_dir = count_co_transitions(
        tree_dir=dirs["output_tree_dir"], msa_dir=msa_dir, contact_map_dir=contact_map_dir,
        families=families, amino_acids=alphabet[:], quantization_points=quantization_points,
        edge_or_cherry=edge_or_cherry, minimum_distance_for_nontrivial_contact=mdnc,
        num_processes=num_processes_counting,
        use_cpp_implementation=use_cpp_counting_implementation,
        cpp_command_line_prefix=cpp_counting_command_line_prefix,
        cpp_command_line_suffix=cpp_counting_command_line_suffix)["output_count_matrices_dir"]
    res["count_matrices_dir_0"] = count_dir
    jtt_dir = jtt_ipw(count_matrices_path=os.path.join(count_dir, "result.txt"),
                      mask_path=coevolution_mask_path, use_ipw=True,
                      normalize=False)["output_rate_matrix_dir"]
    res["jtt_ipw_dir_0"] = jtt_dir
    if optimizer_initialization == "jtt-ipw":
        init_path = os.path.join(jtt_dir, "result.txt")
    elif optimizer_initialization == "random":
        init_path = None
    else:
        raise ValueError(f"Unknown optimizer_initialization = {optimizer_initialization}")
    rate_dir = quantized_transitions_mle(
        count_matrices_path=os.path.join(count_dir, "result.txt"), initialization_path=init_path,
        mask_path=coevolution_mask_path, stationary_distribution_path=None,
        rate_matrix_parameterization="pande_reversible", device=optimizer_device,
        learning_rate=learning_rate, num_epochs=num_epochs, do_adam=do_adam,
        OMP_NUM_THREADS=num_processes_optimization,
        OPENBLAS_NUM_THREADS=num_processes_optimization)["output_rate_matrix_dir"]
    res["rate_matrix_dir_0"] = rate_dir
    res["learned_rate_matrix_path"] = os.path.join(rate_dir, "result.txt")
    # (the reference's co-evolution pipeline reports no timings, _cherry.py:449-584; these keys are extra)
    t_count = _runtime(os.path.join(count_dir, "profiling.txt"))
    t_jtt = _runtime(os.path.join(jtt_dir, "profiling.txt"))
    t_opt = _runtime(os.path.join(rate_dir, "profiling.txt"))
    res["time_counting"], res["time_jtt_ipw"], res["time_optimization"] = t_count, t_jtt, t_opt
    res["profiling_str"] = (
        "CherryML runtimes:\n"
        f"time_counting: {t_count}\ntime_jtt_ipw: {t_jtt}\ntime_optimization: {t_opt}\n"
        f"total_cpu_time: {t_count + t_jtt + t_opt}\n")
    return res
