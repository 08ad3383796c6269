"""Register / spill / LDS metadata of the kernels in the BUILT library's objects (cherryml_amd/build/*.o): the gfx950 code object
is pulled out of each host object (llvm-objdump --offloading) and its notes are read (llvm-readelf --notes).  No GPU needed.

    python profiles/tools/kernel_meta.py [name filter]        # table
    from kernel_meta import kernel_meta; kernel_meta()        # {demangled name: {vgpr, vgpr_spill, sgpr_spill, lds, scratch}}
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_meta(objdir=None):
    objdir = objdir or os.path.join(ROOT, "cherryml_amd", "build")
    out = {}
    tmp = tempfile.mkdtemp()
    try:
        for f in sorted(os.listdir(objdir)):
            if not f.endswith(".o"):
                continue
            shutil.copy(os.path.join(objdir, f), os.path.join(tmp, f))
            subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", f], cwd=tmp, check=True, capture_output=True)
            for co in os.listdir(tmp):
                if not (co.startswith(f + ".") and "amdgcn" in co):
                    continue
                notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], cwd=tmp, check=True,
                                       capture_output=True, text=True).stdout
                cur = {}
                for line in notes.splitlines():
                    m = re.match(r"\s*-?\s*\.(\w+):\s*(\S+)", line)
                    if not m:
                        continue
                    k, v = m.group(1), m.group(2)
                    if k == "args" or (line.lstrip().startswith("- .") and k in ("agpr_count",) and cur.get("name")):
                        pass
                    if line.lstrip().startswith("- ") and k != "name" and "name" in cur and "vgpr_count" in cur:
                        out[cur["name"]] = cur
                        cur = {}
                    cur[k] = v
                if "name" in cur and "vgpr_count" in cur:
                    out[cur["name"]] = cur
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    names = list(out)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    res = {}
    for n, d in zip(names, dem):
        c = out[n]
        res[d] = dict(vgpr=int(c.get("vgpr_count", -1)), vgpr_spill=int(c.get("vgpr_spill_count", 0)),
                      sgpr_spill=int(c.get("sgpr_spill_count", 0)), lds=int(c.get("group_segment_fixed_size", 0)),
                      scratch=int(c.get("private_segment_fixed_size", 0)), symbol=n)
    return res


if __name__ == "__main__":
    filt = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, c in sorted(kernel_meta().items()):
        if filt in name:
            print(f"{name[:100]:100s} vgpr {c['vgpr']:4d} spill {c['vgpr_spill']:3d} sgpr_spill {c['sgpr_spill']:3d} lds {c['lds']:6d} scratch {c['scratch']}")
