"""cherryml_amd: MI355X-native composite-likelihood core with CherryML's API
surface for that path (see DESIGN.md / INTEGRATION.md)."""
from . import caching, counting, io  # noqa: F401
from .counting import count_co_transitions, count_transitions  # noqa: F401
from .estimation_end_to_end import (coevolution_end_to_end_with_cherryml_optimizer,  # noqa: F401
                                    lg_end_to_end_with_cherryml_optimizer)
from ._siterm import learn_site_specific_rate_matrices, quantized_transitions_mle_vectorized_over_sites  # noqa: F401
from .bank import CherryBank  # noqa: F401
from .evaluation import compute_log_likelihoods  # noqa: F401,E402
from .phylogeny_estimation import fast_cherries  # noqa: F401,E402
from ._cherryml_public_api import cherryml_public_api  # noqa: F401,E402
from .estimation import (RateMatrix, RateMatrixLearner, jtt_ipw, quantized_transitions_mle,  # noqa: F401
                         train_quantization)

__all__ = [
    "CherryBank", "RateMatrix", "RateMatrixLearner", "train_quantization",
    "quantized_transitions_mle", "quantized_transitions_mle_vectorized_over_sites", "jtt_ipw",
    "learn_site_specific_rate_matrices", "cherryml_public_api", "compute_log_likelihoods", "fast_cherries",
    "io", "caching", "counting", "count_transitions", "count_co_transitions",
    "lg_end_to_end_with_cherryml_optimizer", "coevolution_end_to_end_with_cherryml_optimizer",
]
