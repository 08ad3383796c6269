"""Filesystem memoisation with the reference's calling convention for stage
functions (cherryml/caching/_cached_computation.py:150-369): keyword-only
calls, output directories chosen from a hash of the arguments when a cache
directory is set, `result.success` tokens, a dict of output dirs returned."""
from ._cached import (  # noqa: F401
    CacheUsageError,
    cached_computation,
    get_cache_dir,
    set_cache_dir,
)
