"""Summarise a rocprofv3 --memory-copy-trace csv: copies by direction and size bucket (count, bytes, time)."""
import csv, glob, sys
from collections import defaultdict
agg = defaultdict(lambda: [0, 0, 0.0])
for path in glob.glob(sys.argv[1] + "/**/*memory_copy_trace.csv", recursive=True):
    with open(path) as f:
        rd = csv.DictReader(f)
        print("# columns:", rd.fieldnames)
        for row in rd:
            d = row.get("Direction", "?")
            size = int(float(row.get("Bytes", row.get("Size", 0)) or 0))
            t = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
            b = 0
            while (1 << b) < max(size, 1):
                b += 1
            a = agg[(d, b)]
            a[0] += 1; a[1] += size; a[2] += t
for (d, b), (c, s, t) in sorted(agg.items()):
    print("%-24s <=2^%-2d B  n=%6d  bytes=%12d  us=%10.1f" % (d, b, c, s, t))
