from ._vectorized import quantized_transitions_mle_vectorized_over_sites  # noqa: F401
