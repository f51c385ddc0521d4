"""Summary of a tests/tools/timeline.py listing: per stream busy time / end, how long k streams are busy, per-stream gaps."""
import sys
import numpy as np
rows = []
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) >= 7 and p[0].replace('.', '').isdigit():
        rows.append((float(p[0]), float(p[1]), float(p[2]), float(p[3]), int(p[5]), p[6], ' '.join(p[7:])))
    elif l.startswith("step"):
        print(l.strip())
ns = max(r[4] for r in rows) + 1
T = max(r[1] for r in rows)
ts = np.arange(0, T, 1.0)
cnt = np.zeros_like(ts)
for s in range(ns):
    rs = sorted(r for r in rows if r[4] == s)
    b = np.zeros_like(ts)
    for r in rs:
        b[(ts >= r[0]) & (ts < r[1])] = 1
    cnt += b
    print("stream %d: %3d launches, busy %5.0f us (alone %5.0f), first %5.0f last %5.0f" % (s, len(rs), b.sum(), sum(r[3] for r in rs), rs[0][0], rs[-1][1]))
for k in range(ns + 1):
    print("%d streams busy: %5.0f us" % (k, (cnt == k).sum()))
