#!/bin/bash
# one frame in flight: the copies of a frame, by duration, with the kernel that ran before each
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; mkdir -p $OUT
SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
rm -rf $OUT/prof_solo; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/copy_prof.log 2>&1
DB=$(find $OUT/prof_solo -name "*_results.db" | head -1)
python - "$DB" > $OUT/${1:-r03}_copies.txt <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print("tables:", [t for t in tabs if "copy" in t.lower() or "memory" in t.lower()])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
cp = [(re.sub(r"\(.*", "", n)[:40], s, e) for n, s, e in rows if "copyBuffer" in n or "fillBuffer" in n]
print("copy/fill kernels:", len(cp), "total us %.1f" % (sum(e - s for _, s, e in cp) / 1e3))
import collections
b = collections.Counter()
t = collections.Counter()
for n, s, e in cp:
    d = (e - s) / 1e3
    k = "<5us" if d < 5 else "<20us" if d < 20 else "<100us" if d < 100 else ">=100us"
    b[(n, k)] += 1
    t[(n, k)] += d
for k in sorted(b):
    print(k, b[k], "%.1f us" % t[k])
for t_ in tabs:
    if "memory_cop" in t_.lower():
        try:
            r = db.execute("select * from %s limit 3" % t_).fetchall()
            print(t_, r[:3])
        except Exception as ex:
            print(t_, ex)
PY
rm -rf $OUT/prof_solo
