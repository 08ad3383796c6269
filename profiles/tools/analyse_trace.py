"""Per-epoch composition of the eigensolver (and the rest of the epoch) from a rocprofv3 --kernel-trace csv."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows)
KEYS = ("lgj_round", "sg_gemm", "lgx_build", "lgx_combine", "lgx_poly8", "lgx_transpose", "lgj_sigma", "lgj_norms", "lgj_finish",
        "lgj_check", "lgj_init", "lgj_sort", "lg_tables", "lg_cast", "k1_pt", "k2_t", "k3_w", "k3_reduce", "lg_finish_loss", "lt_pi",
        "lt_build", "lt_gd", "lt_step_pi", "lt_step_up")
def short(n):
    for k in KEYS:
        if k in n:
            return k
    return n[:24]
idx = [i for i, e in enumerate(ev) if e[2].startswith("lt_pi")]
print("epochs", len(idx))
for e in [int(x) for x in sys.argv[2:]] or [10, 50, 100, 150]:
    if e + 1 >= len(idx):
        continue
    seg = ev[idx[e]:idx[e + 1]]
    names = [short(s[2]) for s in seg]
    i0, i1 = names.index("lgj_sigma"), names.index("lgj_finish")
    es = seg[i0:i1 + 1]
    cnt, dur = collections.Counter(), collections.defaultdict(int)
    for x in seg:
        k = short(x[2]); cnt[k] += 1; dur[k] += x[1] - x[0]
    print(f"epoch {e}: span {(seg[-1][1] - seg[0][0]) / 1e3:.0f} us; eigh span {(es[-1][1] - es[0][0]) / 1e3:.0f} us in {len(es)} kernels")
    print("   ", {k: (cnt[k], round(dur[k] / 1e3, 1)) for k in cnt})
