"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite) kernel trace into the per-kernel stats table committed under profiles/.
usage: python profiles/summarise_rocpd.py <results.db> "<command that was profiled>" """
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,count(*),sum(duration),avg(duration),min(duration),max(duration) from kernels "
                  "group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
print("# rocprofv3 --kernel-trace --stats (rocpd SQLite output summarised by profiles/summarise_rocpd.py)")
print("# command:", sys.argv[2] if len(sys.argv) > 2 else "?")
print("# total kernel time %.1f ms over %d dispatches" % (tot / 1e6, sum(r[1] for r in rows)))
print("%-46s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
for n, c, t, a, mn, mx in rows:
    s = n.replace("tmc2::(anonymous namespace)::", "").replace("void ", "")
    s = re.sub(r"\(.*", "", s)
    print("%-46s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (s[:46], c, t / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100 * t / tot))
