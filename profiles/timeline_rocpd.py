"""Per-dispatch timeline of a rocprofv3 (rocpd SQLite) kernel trace: start offset, duration and the idle gap since the
previous dispatch ended, plus a per-kernel total of (duration, gap-before).  Shows launch-bound loops at a glance.
usage: python profiles/timeline_rocpd.py <results.db> [first_row [rows]]"""
import re
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else 80
t0, prev = rows[0][1], rows[0][1]
agg = defaultdict(lambda: [0, 0.0, 0.0])
out = []
for i, (n, s, e) in enumerate(rows):
    name = re.sub(r"\(.*", "", n.replace("tmc2::(anonymous namespace)::", "").replace("void ", ""))[:40]
    gap = max(0, s - prev) / 1e3
    a = agg[name]
    a[0] += 1
    a[1] += (e - s) / 1e3
    a[2] += gap
    out.append("%6d %-40s start %10.1f us  dur %8.2f us  gap %8.2f us" % (i, name, (s - t0) / 1e3, (e - s) / 1e3, gap))
    prev = max(prev, e)
print("# span %.2f ms, busy %.2f ms, %d dispatches" % ((prev - t0) / 1e6, sum(a[1] for a in agg.values()) / 1e3, len(rows)))
print("%-40s %7s %12s %14s" % ("kernel", "calls", "busy_us", "gap_before_us"))
for k, a in sorted(agg.items(), key=lambda kv: -(kv[1][1] + kv[1][2])):
    print("%-40s %7d %12.1f %14.1f" % (k, a[0], a[1], a[2]))
print()
print("\n".join(out[first:first + count]))
