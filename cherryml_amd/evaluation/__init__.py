"""Held-out log-likelihood evaluation (reference: cherryml/evaluation/__init__.py)."""
from ._likelihood import (LikelihoodModel, compute_log_likelihoods, dp_likelihood_computation, dp_likelihood_computation_batch,  # noqa: F401
                          tree_likelihood, tree_likelihood_batch)
