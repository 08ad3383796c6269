from ._jtt_ipw import jtt_ipw, jtt_ipw_from_arrays, jtt_ipw_from_statistics, jtt_ipw_statistics  # noqa: F401
from ._quantized_transitions_mle import quantized_transitions_mle  # noqa: F401
from ._ratelearn import RateMatrix, RateMatrixLearner, train_quantization  # noqa: F401
