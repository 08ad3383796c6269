"""API calls longer than <ms> in the rocprofv3 --hip-trace / --hsa-trace csvs of an output directory (which call of the run
blocked the host?).   python profiles/tools/find_slow_api.py <dir> [ms]"""
import csv, glob, os, sys
lim = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
for f in sorted(glob.glob(os.path.join(sys.argv[1], "**", "*_api_trace.csv"), recursive=True)):
    rows = list(csv.DictReader(open(f)))
    if not rows:
        continue
    t0 = min(int(r["Start_Timestamp"]) for r in rows)
    for r in rows:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d > lim:
            print(f"{os.path.basename(f)[-22:]:22s} {(int(r['Start_Timestamp']) - t0) / 1e6:10.2f} ms  {d:8.2f} ms  {r['Function']}")
