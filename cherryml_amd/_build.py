"""Build libcherrybank.so (HIP, gfx950) in-tree with hipcc."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(PKG_DIR, "csrc", "cherrybank.hip")
LIB = os.path.join(PKG_DIR, "libcherrybank.so")


def _sources():
    d = os.path.join(PKG_DIR, "csrc")
    inc = os.path.join(os.path.dirname(PKG_DIR), "include")
    out = [os.path.join(d, f) for f in sorted(os.listdir(d))]
    out += [os.path.join(inc, f) for f in sorted(os.listdir(inc))]
    return out


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    mt = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > mt for s in _sources())


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile for gfx950.  hipcc cross-compiles without a GPU."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-Wall", "-Wextra",
           SRC, "-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
