"""FastCherries: branch-length / site-rate estimation on the GPU (log-transition bank, the
two bisection passes, their coordinate ascent) and, around it, the reference's seeded divide-and-conquer
cherry pairing (host) and tree / site-rate files (`fast_cherries`, `fast_cherries_family`)."""
from ._ble import (  # noqa: F401
    BleBank,
    branch_lengths,
    compute_log_transition_matrices,
    estimate_branch_lengths_and_site_rates,
    estimate_branch_lengths_and_site_rates_batch,
    rate_priors,
    site_rates,
)
from ._fast_cherries import (  # noqa: F401,E402
    cherries_to_tree,
    divide_and_pair,
    fast_cherries,
    fast_cherries_families,
    fast_cherries_family,
    get_weights_for_initial_site_rates,
    rate_categories_ble,
)
