"""Kernel-by-kernel timeline of chosen epochs of the 400-state trainer from a rocprofv3 --kernel-trace csv:
start offset, duration, gap to the previous kernel's end.  python epoch_timeline.py trace.csv [epoch ...]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "")) for r in rows)
idx = [i for i, e in enumerate(ev) if e[2].startswith("lt_pi")] or [i for i, e in enumerate(ev) if e[2].startswith("lt_build")]
print("epochs", len(idx))
for e in ([] if sys.argv[2:3] == ["spans"] else [int(x) for x in sys.argv[2:]] or [15]):
    if e + 1 >= len(idx):
        continue
    seg = ev[idx[e]:idx[e + 1]]
    t0, prev = seg[0][0], seg[0][0]
    print(f"== epoch {e}: span {(seg[-1][1] - t0) / 1e3:.1f} us, kernels {sum(s[1] - s[0] for s in seg) / 1e3:.1f} us in {len(seg)} launches")
    for s in seg:
        print(f"  {(s[0] - t0) / 1e3:8.1f}  dur {(s[1] - s[0]) / 1e3:7.2f}  gap {(s[0] - prev) / 1e3:6.2f}  {s[2][:40]}")
        prev = s[1]
if len(sys.argv) > 2 and sys.argv[2] == "spans":   # python epoch_timeline.py trace.csv spans: one line per epoch
    for e in range(len(idx) - 1):
        seg = ev[idx[e]:idx[e + 1]]
        gaps = sum(max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1))
        big = max((seg[i + 1][0] - seg[i][1], seg[i][2][:20], seg[i + 1][2][:20]) for i in range(len(seg) - 1))
        print(f"epoch {e:3d}: span {(ev[idx[e + 1]][0] - seg[0][0]) / 1e3:8.1f} us  launches {len(seg):3d}  gaps {gaps / 1e3:6.1f}  largest gap {big[0] / 1e3:6.1f} us ({big[1]} -> {big[2]})")
