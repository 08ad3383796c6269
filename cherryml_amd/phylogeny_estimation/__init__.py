"""FastCherries' branch-length / site-rate estimation on the GPU.  Only the
likelihood kernels of FastCherries are here (log-transition bank, the two bisection passes, their
coordinate ascent); its cherry-pairing heuristics and tree writing are not part of this build."""
from ._ble import (  # noqa: F401
    branch_lengths,
    compute_log_transition_matrices,
    estimate_branch_lengths_and_site_rates,
    rate_priors,
    site_rates,
)
