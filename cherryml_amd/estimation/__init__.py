from ._jtt_ipw import jtt_ipw, jtt_ipw_from_arrays  # noqa: F401
from ._quantized_transitions_mle import quantized_transitions_mle  # noqa: F401
from ._ratelearn import RateMatrix, RateMatrixLearner, train_quantization  # noqa: F401
