#!/bin/bash
# one frame in flight: the blit-kernel copies (D2H / D2D) of a frame by duration, with the kernel that ran before each
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; mkdir -p $OUT
SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
rm -rf $OUT/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/copy_prof.log 2>&1
DB=$(find $OUT/prof_solo -name "*_results.db" | head -1)
python - "$DB" > $OUT/${1:-r03}_copies.txt <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
nm = lambda n: re.sub(r"\(.*", "", n.replace("tmc2::(anonymous namespace)::", "").replace("void ", ""))[:36]
for i, (n, s, e) in enumerate(rows):
    if "copyBuffer" in n and (e - s) > 30000:
        print("%8.1f us  after %-36s  before %s" % ((e - s) / 1e3, nm(rows[i - 1][0]), nm(rows[i + 1][0]) if i + 1 < len(rows) else "-"))
PY
rm -rf $OUT/prof_solo
