"""What the reference's `device` arguments mean here.

In the reference `device` ("cpu", its default everywhere, or "cuda") names where torch keeps the
tensors of the optimiser (`_quantized_transitions_mle.py:46,61`, `_cherry.py:221`,
`_cherryml_vectorized.py:110`).  This package has ONE execution target for that path -- the HIP
kernels of libcherrybank on an MI355X -- and inputs / outputs are host arrays and files either way.
So both spellings are accepted with the reference's signatures and defaults, and both run on the GPU:
"cpu" is NOT a CPU fallback (there is none: without a GPU or without the built library every entry
point raises `CherryBankError`), it only says the caller kept the reference's default.  A
`UserWarning` says so once per process."""
import warnings

_warned = False


def resolve_device(device: str, what: str) -> str:
    """Validate a reference-style `device` argument; returns "cuda" (the HIP device's torch name)."""
    global _warned
    if device not in ("cpu", "cuda"):
        raise ValueError(f'{what}: device must be "cpu" or "cuda" (got {device!r})')
    from . import _lib
    if _lib.load().cb_device_count() <= 0:   # (a missing library raises CherryBankError inside load())
        raise _lib.CherryBankError(f"{what}: no HIP device visible; cherryml_amd computes this path on the MI355X only "
                                   "and has no CPU fallback")
    if device == "cpu" and not _warned:
        _warned = True
        warnings.warn(f'{what}: device="cpu" (the reference\'s default) is accepted for call compatibility, but '
                      "cherryml_amd computes this path on the MI355X only; there is no CPU implementation.",
                      UserWarning, stacklevel=3)
    return "cuda"
