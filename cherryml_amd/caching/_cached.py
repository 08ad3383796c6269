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


def cached_computation(output_dirs: List[str], exclude_args: Optional[List[str]] = None,
                       write_extra_log_files: bool = False):
    """Decorator for a stage function whose outputs are directories.

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
            if all(os.path.exists(tk) for tk in tokens):
                return chosen
            for d in chosen.values():
                os.makedirs(d, exist_ok=True)
                tk = os.path.join(d, "result.success")
                if os.path.exists(tk):
                    os.remove(tk)
            func(**full)
            for d in chosen.values():
                if write_extra_log_files:
                    with open(os.path.join(d, "_function_binding.log"), "w") as f:
                        f.write(func.__name__ + "\n" +
                                "\n".join(f"{k}={v!r}" for k, v in key_items) + "\n")
                with open(os.path.join(d, "result.success"), "w") as f:
                    f.write("SUCCESS\n")
            return chosen

        return wrapper

    return deco
