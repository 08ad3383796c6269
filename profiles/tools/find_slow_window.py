"""In a rocprofv3 --hip-trace output directory: the 100 ms window (after the first 3 s) in which the host spent the most time
inside HIP API calls, and what it called there.   python profiles/tools/find_slow_window.py <dir>"""
import csv, glob, os, sys, collections
f = sorted(glob.glob(os.path.join(sys.argv[1], "**", "*hip_api_trace.csv"), recursive=True))[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Function"]) for r in csv.DictReader(open(f))]
t0 = min(r[0] for r in rows)
rows = [(a - t0, b - t0, fn) for a, b, fn in rows if a - t0 > 3e9]
rows.sort()
# gaps between consecutive API calls (host busy outside HIP) and long calls
big = sorted(((rows[i + 1][0] - rows[i][1]) / 1e6, rows[i][1] / 1e6, rows[i][2], rows[i + 1][2]) for i in range(len(rows) - 1))[-6:]
print("largest gaps BETWEEN consecutive HIP calls (ms, at ms, after, before):")
for g in big:
    print("  %8.2f ms at %9.2f  after %s  before %s" % g)
longest = sorted(((b - a) / 1e6, a / 1e6, fn) for a, b, fn in rows)[-6:]
print("longest HIP calls:")
for d, at, fn in longest:
    print("  %8.2f ms at %9.2f  %s" % (d, at, fn))
