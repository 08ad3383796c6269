"""Build libcherrybank.so (HIP, gfx950) in-tree with hipcc."""
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
# one translation unit per subsystem (kernels live in the *.hip.h headers each of them includes)
SRCS = [os.path.join(CSRC, f) for f in ("cherrybank.hip", "cb_bank_fused.hip", "cb_tbasis.hip", "cb_counting.hip", "cb_ble.hip",
                                        "cb_likelihood.hip", "cb_host_io.hip")]
# per-file flags (cb_bank_fused.hip says why)
FILE_FLAGS = {"cb_bank_fused.hip": ["-mllvm", "-disable-machine-licm"]}
LIB = os.path.join(PKG_DIR, "libcherrybank.so")


def _sources():
    d = os.path.join(PKG_DIR, "csrc")
    inc = os.path.join(os.path.dirname(PKG_DIR), "include")
    out = [os.path.join(d, f) for f in sorted(os.listdir(d))]
    out += [os.path.join(inc, f) for f in sorted(os.listdir(inc))]
    return out


FLAGS_FILE = LIB + ".flags"   # the CB_EXTRA_HIPCC_FLAGS the library was built with (git-ignored like the library)


def _extra_flags() -> str:
    return " ".join(os.environ.get("CB_EXTRA_HIPCC_FLAGS", "").split())


def built_flags() -> str:
    try:
        with open(FLAGS_FILE) as f:
            return f.read().strip()
    except OSError:
        return ""


def needs_build() -> bool:
    """Stale when a source is newer than the library -- or when the library was built with OTHER experiment flags than
    the ones asked for now (a profiling script that rebuilt with -DCB_CO_PLAIN must not leave that library behind for the
    next bench or test run: ADVICE r3)."""
    if not os.path.exists(LIB):
        return True
    if built_flags() != _extra_flags():
        return True
    mt = os.path.getmtime(LIB)
    return any(os.path.getmtime(s) > mt for s in _sources())


def build(force: bool = False, verbose: bool = False) -> str:
    """Compile for gfx950.  hipcc cross-compiles without a GPU."""
    if not force and not needs_build():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wextra"]
    flags += os.environ.get("CB_EXTRA_HIPCC_FLAGS", "").split()   # (experiments: e.g. -DCB_SETPRIO)
    objdir = os.path.join(PKG_DIR, "build")
    os.makedirs(objdir, exist_ok=True)
    # compile the translation units side by side, then link
    procs = []
    for src in SRCS:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        cmd = [hipcc] + flags + FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((obj, cmd, subprocess.Popen(cmd)))
    for obj, cmd, proc in procs:
        if proc.wait() != 0:
            raise subprocess.CalledProcessError(proc.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + [obj for obj, _, _ in procs] + ["-o", LIB + ".tmp"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    with open(FLAGS_FILE, "w") as f:
        f.write(_extra_flags() + "\n")
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
