"""ctypes binding of include/cherrybank.h.  There is no fallback: if the HIP
library is missing or a call fails, the caller gets an exception."""
import ctypes as C
import os

from ._build import LIB

CB_PTR_DEVICE = 1
CB_NORMALIZE = 2
CB_TRAIN_RESUME = 16
CB_NO_SYNC = 4
CB_EXPM_ONLY = 8
CB_PER_BUCKET_PRODUCTS = 32
CB_F64, CB_F32, CB_MIXED = 0, 1, 2

CB_EINVAL, CB_EHIP, CB_ENOMEM, CB_ENUMERIC, CB_EUNSUPPORTED = -1, -2, -3, -4, -5

_dp = C.POINTER(C.c_double)
_vp = C.c_void_p

# name -> (restype, argtypes); must list every symbol of include/cherrybank.h
SIGNATURES = {
    "cb_version": (C.c_int, []),
    "cb_last_error": (C.c_char_p, []),
    "cb_device_count": (C.c_int, []),
    "cb_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int, C.POINTER(_vp)]),
    "cb_destroy": (None, [_vp]),
    "cb_set_stream": (C.c_int, [_vp, _vp, C.c_int]),
    "cb_total_counts": (C.c_int, [_vp, _vp]),
    "cb_live_buckets": (C.c_int, [_vp, _vp]),
    "cb_allreduce_setup": (C.c_int, [_vp, _vp, _vp, _vp]),
    "cb_format_matrix_rows": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p, _vp, _vp, _vp, C.c_size_t, _vp]),
    "cb_fc_divide_and_pair": (C.c_int, [_vp, C.c_int, C.c_int, C.c_uint, C.c_int, _vp]),
    "cb_parse_count_matrices": (C.c_int, [C.c_char_p, C.c_size_t, C.c_int, C.c_int, _vp, _vp, _vp, _vp, C.c_int]),
    "cb_loss_grad": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp, _vp]),
    "cb_loss_grad_general": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "cb_expm_bank": (C.c_int, [_vp, _vp, _vp, C.c_int, _vp]),
    "cb_eigh": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp]),
    "cb_profile": (C.c_int, [_vp, C.c_int]),
    "cb_last_timings": (C.c_int, [_vp, _vp, C.c_int]),
    "cb_timing_sums": (C.c_int, [_vp, _vp, C.c_int, C.POINTER(C.c_int)]),
    "cb_last_sweeps": (C.c_int, [_vp]),
    "cb_eigh_counters": (C.c_int, [_vp, _vp, C.c_int]),
    "cb_last_kernel_form": (C.c_int, [_vp]),
    "cb_last_bank_form": (C.c_int, [_vp]),
    "cb_time_basis_info": (C.c_int, [_vp, _vp, _vp]),
    "cb_time_basis": (C.c_int, [C.c_int, _vp, C.c_double, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cb_train_pande_reversible": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_double, C.c_int,
                                            C.c_int, _vp, _vp, _vp, _vp, C.c_int]),
    "cb_train_siterm": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_double, C.c_int, _vp, _vp]),
    "cb_train_epoch_times": (C.c_int, [_vp, _vp, C.c_int]),
    "cb_count_transitions": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int64, _vp, C.c_int64,
                                       _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "cb_jtt_ipw_stats": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_double, C.c_int, C.c_int, _vp, _vp]),
    "cb_count_co_transitions": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int64, _vp, C.c_int64,
                                          _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "cb_ble_log_bank": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "cb_ble_branch_lengths": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp]),
    "cb_ble_site_rates": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, _vp, _vp]),
    "cb_ble": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp,
                         C.c_int, _vp, _vp, _vp, _vp]),
    "cb_ble_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                               C.c_int, _vp, _vp, _vp, _vp]),
    "cb_ble_bank_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]),
    "cb_ble_bank_run": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp, _vp]),
    "cb_ble_bank_destroy": (C.c_int, [_vp]),
    "cb_site_rate_gather": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "cb_tree_likelihood": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, C.c_int, _vp,
                                     C.c_int, _vp, _vp, _vp, _vp, _vp]),
    "cb_tree_likelihood_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int, _vp, _vp, _vp, _vp, _vp,
                                           _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cb_tl_model_create": (C.c_int, [C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, _vp]),
    "cb_tl_model_run": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "cb_tl_model_destroy": (C.c_int, [_vp]),
    "cb_siterm_assemble_batch": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp, C.c_int64, _vp, _vp, _vp, _vp,
                                           C.c_double, C.c_int, C.c_int, _vp, _vp]),
    "cb_siterm_assemble": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _vp, _vp, C.c_int64, _vp, C.c_int64,
                                     _vp, _vp, C.c_double, C.c_int, C.c_int, _vp, _vp]),
}

_lib = None


class CherryBankError(RuntimeError):
    pass


def load():
    """Load libcherrybank.so (built by cherryml_amd._build.build / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB):
        raise CherryBankError(
            f"{LIB} is missing: the HIP extension has not been built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (needs hipcc). "
            "cherryml_amd has no CPU fallback.")
    try:   # a library left behind by a profiling script that built with experiment macros must not pass unnoticed
        with open(LIB + ".flags") as f:
            built = f.read().strip()
    except OSError:
        built = ""
    if built and built != " ".join(os.environ.get("CB_EXTRA_HIPCC_FLAGS", "").split()):
        import warnings
        warnings.warn(f"{LIB} was built with CB_EXTRA_HIPCC_FLAGS={built!r} (an experiment build); rebuild with "
                      "`python -m cherryml_amd._build` for the shipped kernels", RuntimeWarning)
    lib = C.CDLL(LIB)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc == 0:
        return
    msg = load().cb_last_error().decode("utf-8", "replace")
    text = f"{what} failed ({rc}): {msg}"
    if rc in (CB_EINVAL, CB_ENUMERIC):
        raise ValueError(text)
    if rc == CB_EUNSUPPORTED:
        raise NotImplementedError(text)
    if rc == CB_ENOMEM:
        raise MemoryError(text)
    raise CherryBankError(text)
