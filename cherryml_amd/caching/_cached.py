import functools
import hashlib
import inspect
import os
from typing import List, Optional

_CACHE_DIR: Optional[str] = None
_HASH_LEN = 64


class CacheUsageError(Exception):
    pass


def set_cache_dir(cache_dir: Optional[str]) -> None:
    global _CACHE_DIR
    _CACHE_DIR = cache_dir


def get_cache_dir() -> Optional[str]:
    return _CACHE_DIR


def _digest(func_name: str, items) -> str:
    text = func_name + "".join(f"[{k}={v!r}]" for k, v in items)
    return hashlib.sha512(text.encode("utf-8")).hexdigest()[:_HASH_LEN]


def _dist_state():
    """(rank, world) of the default torch.distributed group, (0, 1) without one."""
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return dist.get_rank(), dist.get_world_size()
    except ImportError:  # pragma: no cover
        pass
    return 0, 1


def _agree(value):
    """rank 0's `value` on every rank (one process: the value itself)."""
    rank, world = _dist_state()
    if world == 1:
        return value
    import torch.distributed as dist
    box = [value if rank == 0 else None]
    dist.broadcast_object_list(box, src=0)
    return box[0]


def _raise_together(error: Optional[str], func_name: str):
    """Every rank learns whether ANY rank failed and raises then (a rank that failed alone would
    leave its peers waiting in the next collective)."""
    rank, world = _dist_state()
    if world == 1:
        return
    import torch.distributed as dist
    errs = [None] * world
    dist.all_gather_object(errs, error)
    bad = [(r, e) for r, e in enumerate(errs) if e is not None]
    if bad:
        raise RuntimeError(f"{func_name} failed on rank {bad[0][0]}: {bad[0][1]}")


def cached_computation(output_dirs: List[str], exclude_args: Optional[List[str]] = None,
                       write_extra_log_files: bool = False, collective: bool = False):
    """Decorator for a stage function whose outputs are directories.

    Under torch.distributed (world > 1) the ranks share the cache directory: rank 0 decides
    whether the stage has to run (its view of the success tokens is broadcast), a `collective`
    stage runs on every rank (it shards its work and reduces inside, and writes its files on rank
    0 only), any other stage runs on rank 0 alone; failures are exchanged so that all ranks raise
    together; the success tokens are written by rank 0 between two barriers, so no rank reads an
    output another rank is still writing.

    * positional arguments are refused (CacheUsageError), as in the reference;
    * with no cache directory set, the call goes straight through and returns
      the function's own result (None for stage functions);
    * otherwise every output-dir argument left as None is set to
      <cache>/<function>/<hash of the other arguments>/<name>, the function
      runs unless "<dir>/result.success" already exists for all of them, and
      a dict {name: dir} is returned.
    """
    exclude = set(exclude_args or [])

    def deco(func):
        sig = inspect.signature(func)

        @functools.wraps(func)
        def wrapper(*args, **kwargs):
            if args:
                raise CacheUsageError(
                    f"Please call {func.__name__} with keyword arguments only; "
                    f"positional arguments are not allowed: {args}")
            bound = sig.bind_partial(**kwargs)
            bound.apply_defaults()
            full = dict(bound.arguments)
            for name in output_dirs:
                full.setdefault(name, None)
            if _CACHE_DIR is None:
                return func(**full)
            key_items = sorted((k, v) for k, v in full.items()
                               if k not in exclude and k not in output_dirs)
            h = _digest(func.__name__, key_items)
            chosen = {}
            for name in output_dirs:
                if full.get(name) is None:
                    full[name] = os.path.join(_CACHE_DIR, func.__name__, h, name)
                chosen[name] = full[name]
            tokens = [os.path.join(d, "result.success") for d in chosen.values()]
            rank, world = _dist_state()
            if _agree(all(os.path.exists(tk) for tk in tokens)):
                return chosen
            def prepare():
                for d in chosen.values():
                    os.makedirs(d, exist_ok=True)
                    tk = os.path.join(d, "result.success")
                    if os.path.exists(tk):
                        os.remove(tk)

            def write_tokens():
                for d in chosen.values():
                    if write_extra_log_files:
                        with open(os.path.join(d, "_function_binding.log"), "w") as f:
                            f.write(func.__name__ + "\n" +
                                    "\n".join(f"{k}={v!r}" for k, v in key_items) + "\n")
                    with open(os.path.join(d, "result.success"), "w") as f:
                        f.write("SUCCESS\n")

            if world == 1:
                prepare()
                func(**full)
                write_tokens()
                return chosen
            # world > 1: every step that can fail on one rank alone (rank 0's filesystem work, the stage body) ends in
            # an exchange of the error, so that no rank is left waiting in a collective its peer never enters
            def guarded(step, runs_here):
                error = None
                if runs_here:
                    try:
                        step()
                    except Exception as exc:
                        error = f"{type(exc).__name__}: {exc}"
                _raise_together(error, func.__name__)

            guarded(prepare, rank == 0)          # the directories exist / stale tokens are gone before any rank starts
            guarded(lambda: func(**full), collective or rank == 0)
            guarded(write_tokens, rank == 0)     # tokens are on disk before any rank moves on to read the outputs
            return chosen

        return wrapper

    return deco
