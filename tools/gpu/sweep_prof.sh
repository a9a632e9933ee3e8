#!/bin/bash
# one frame in flight, kernel trace; per-sweep durations of the S5 kernels
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; mkdir -p $OUT
SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
rm -rf $OUT/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/sweep_prof.log 2>&1
DB=$(find $OUT/prof_solo -name "*_results.db" | head -1)
python - "$DB" > $OUT/${1:-r03}_sweep_kernels.txt <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
want = ("closureKernel", "sweepDirtyKernel", "sweepClosureKernel", "sweepProcessKernel", "closurePrepareKernel", "closureLevelsKernel", "sweepKernel")
per = {}
for n, s, e in rows:
    m = re.search(r"(\w+Kernel)", n)
    k = m.group(1) if m else n
    if k in want:
        per.setdefault(k, []).append((e - s) / 1e3)
for k, v in per.items():
    last = v[-50:]
    print(k, "calls", len(v), "avg %.1f" % (sum(last) / len(last)))
    print("   ", " ".join("%.0f" % x for x in last))
PY
python $REPO/profiles/summarise_rocpd.py "$DB" "$SOLO" | head -40 >> $OUT/${1:-r03}_sweep_kernels.txt
rm -rf $OUT/prof_solo
