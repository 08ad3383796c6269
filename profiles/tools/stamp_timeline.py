"""Timeline of the bank's tiles from the in-kernel stamps of a -DCB_CLOCK_STAMP build (profiles/tools/clock_probe.py writes
gpurun_out/clock_stamps_*.npy): per stage first / last start, K-loop and epilogue medians, and every `step` us the number of
tiles inside their K loop / inside their epilogue.   python profiles/tools/stamp_timeline.py <stamps.npy> [step_us]"""
import sys
import numpy as np

b = np.load(sys.argv[1]).astype(np.int64)
step = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
S = []
for kid in range(3):
    s = b[kid]
    S.append(s[(s[:, 2] > 0) & (s[:, 4] > 0)])
t0 = min(s[:, 2].min() for s in S if len(s))
for kid, s in enumerate(S):
    if not len(s):   # (K3 is gone when the buckets are summed before the last product)
        print("K%d: no tiles" % (kid + 1))
        continue
    st, ke, en = [(s[:, c] - t0) * 0.01 for c in (2, 3, 4)]
    print("K%d: %d tiles, first start %.1f, last start %.1f, last end %.1f us; K loop median %.1f us (clock %.2f GHz), epilogue "
          "median %.1f us" % (kid + 1, len(s), st.min(), st.max(), en.max(), np.median(ke - st),
                              np.median(s[:, 0] / np.maximum(s[:, 1], 1) * 0.1), np.median(en - ke)))
tend = max(((s[:, 4] - t0) * 0.01).max() for s in S if len(s))
print("   t us | tiles in flight | (in K loop, in epilogue) per stage")
for t in np.arange(5, tend + step, step):
    row = []
    for s in S:
        st, ke, en = [(s[:, c] - t0) * 0.01 for c in (2, 3, 4)]
        row.append((int(np.sum((st <= t) & (ke > t))), int(np.sum((ke <= t) & (en > t)))))
    print(f"  {t:6.0f} | {sum(a + c for a, c in row):5d} | K1 {row[0]}  K2 {row[1]}  K3 {row[2]}")
# slots (CU, wave slot of wave 0): how busy, and the gap between consecutive tiles of one slot (fused launch: ticket draw etc.)
allt = np.concatenate(S)
hw = allt[:, 5]
key = (hw >> 32) * 100000 + ((hw >> 13) & 7) * 10000 + ((hw >> 12) & 1) * 1000 + ((hw >> 8) & 0xf) * 10 + (hw & 0xf) % 4
busy = float(((allt[:, 4] - allt[:, 2]) * 0.01).sum())
nslots = len(np.unique(key))
gaps, ends = [], []
for k in np.unique(key):
    t = allt[key == k]
    t = t[np.argsort(t[:, 2])]
    gaps.extend(((t[1:, 2] - t[:-1, 4]) * 0.01).tolist())
    ends.append((t[-1, 4] - t0) * 0.01)
gaps, ends = np.array(gaps), np.array(ends)
print(f"{nslots} slots; busy slot-time / slots = {busy / nslots:.1f} us of {tend:.1f} us ({busy / nslots / tend:.2f}); "
      f"sum of K-loop matrix work at 100 % = {sum(len(s) for s in S) * 625 * 64 * 4 / 1024 / 2400.0:.1f} us (625 MFMAs of 64 cycles per wave and tile, 1024 SIMDs at 2.4 GHz)")
if len(gaps):
    print(f"gap between consecutive tiles of a slot: median {np.median(gaps):.2f} us, mean {gaps.mean():.2f}, p90 {np.percentile(gaps, 90):.2f}"
          f"; slots finish at p10 {np.percentile(ends, 10):.0f} / median {np.median(ends):.0f} / max {ends.max():.0f} us")
