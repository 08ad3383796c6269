"""Held-out log-likelihood evaluation (reference: cherryml/evaluation/__init__.py)."""
from ._likelihood import compute_log_likelihoods, dp_likelihood_computation, tree_likelihood  # noqa: F401
