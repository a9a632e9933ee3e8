#!/bin/bash
# round 5, call 10: what the FIRST pass of a process over a GOF spends its time in (HIP API trace of one pass, reservation on)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
REPO=$(pwd); ONE="python $REPO/bench.py --steps 1 --warmup 0 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
rm -rf $O/prof_first; timeout 900 rocprofv3 --hip-trace --stats -d $O/prof_first -- $ONE > $O/r05c10_first.log 2>&1
ls $O/prof_first/*/ | head
DB=$(find $O/prof_first -name "*_results.db" | head -1)
python - "$DB" > $O/r05c10_hip_api_first_pass.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
print("# tables/views:", [t for t in tabs if 'region' in t or 'api' in t.lower()][:12])
try:
    rows = db.execute("select name, count(*), sum(end-start)/1e6, max(end-start)/1e6 from regions group by name order by 3 desc limit 25").fetchall()
except Exception as e:
    rows = []
    print("# regions view failed:", e)
for r in rows:
    print("%-40s calls %7d total %10.1f ms max %8.2f ms" % (r[0][:40], r[1], r[2], r[3]))
PY
cat $O/r05c10_hip_api_first_pass.txt | head -40
rm -rf $O/prof_first
