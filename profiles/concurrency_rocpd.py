"""GPU occupancy over time from a rocprofv3 (rocpd SQLite) kernel trace of a multi-stream run: for the busiest window of the
trace (the timed GOF passes), the share of wall time with 0, 1, 2, ... kernels in flight, the mean number in flight, and the
idle time -- tells a GPU-bound run (never idle, many kernels queued) from a host-/latency-bound one (idle gaps).
usage: python profiles/concurrency_rocpd.py <results.db> [skip_leading_fraction]"""
import sqlite3
import sys
from collections import Counter

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select start,end from kernels order by start").fetchall()
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t0, t1 = rows[0][0], max(r[1] for r in rows)
lo = t0 + skip * (t1 - t0)
ev = []
for s, e in rows:
    if e <= lo:
        continue
    ev.append((max(s, lo), 1))
    ev.append((e, -1))
ev.sort()
level, last, hist, gaps = 0, lo, Counter(), []
for t, d in ev:
    hist[level] += t - last
    if level == 0 and t > last:
        gaps.append((t - last, last))
    last, level = t, level + d
span = float(last - lo)
print("# window %.1f ms, %d dispatches, %.0f dispatches/s" % (span / 1e6, len(ev) // 2, (len(ev) // 2) / (span / 1e9)))
print("# kernels in flight: share of wall time")
acc = 0.0
for k in sorted(hist):
    acc += k * hist[k]
    if hist[k] / span >= 0.002:
        print("%3d  %6.2f %%" % (k, 100.0 * hist[k] / span))
print("# mean kernels in flight %.2f, idle %.2f %%" % (acc / span, 100.0 * hist[0] / span))
print("# idle gaps by length: count, total ms")
for a, b in ((0, 10e3), (10e3, 100e3), (100e3, 1e6), (1e6, 5e6), (5e6, 20e6), (20e6, 1e12)):
    sel = [g for g, _ in gaps if a <= g < b]
    print("%8.0f us .. %8.0f us  n=%6d  %8.2f ms" % (a / 1e3, min(b, 1e9) / 1e3, len(sel), sum(sel) / 1e6))
print("# the ten longest idle gaps: length ms, at ms into the window")
for g, t in sorted(gaps, reverse=True)[:10]:
    print("  %8.2f ms at %9.2f ms" % (g / 1e6, (t - lo) / 1e6))
