from ._vectorized import quantized_transitions_mle_vectorized_over_sites  # noqa: F401
from ._assembly import (  # noqa: F401
    estimate_site_specific_rate_matrices_given_tree_and_site_rates,
    estimate_site_specific_rate_matrices_given_trees_and_site_rates,
    get_cherry_transitions,
    get_count_prior_probability_matrices,
    get_edge_transitions,
    get_raw_count_matrices,
)
from ._site_rates import compute_optimal_site_rates  # noqa: F401
from ._learn import (  # noqa: F401
    get_standard_site_rate_grid,
    get_standard_site_rate_prior,
    learn_site_rate_matrices,
    learn_site_specific_rate_matrices,
)
