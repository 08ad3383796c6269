from ._epoch_loop import train_quantization  # noqa: F401
from ._learner import RateMatrixLearner  # noqa: F401
from ._rate_matrix import RateMatrix  # noqa: F401
