#!/bin/bash
# one frame in flight: where is the GPU idle inside a frame?  (gaps > 40 us between consecutive kernels, with their neighbours)
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; mkdir -p $OUT
SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
rm -rf $OUT/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/gaps.log 2>&1
DB=$(find $OUT/prof_solo -name "*_results.db" | head -1)
python - "$DB" > $OUT/${1:-r05}_gaps_one_frame.txt 2>&1 <<'PY'
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
nm = lambda n: re.sub(r"\(.*", "", n.replace("tmc2::(anonymous namespace)::", "").replace("void ", ""))[:34]
# the timed frame: from the second initKernel-before-knn16-self ... simply take the window between the 2nd and 3rd occurrence of the first kernel of a frame
first = [i for i, r in enumerate(rows) if "lvInitKernel" in r[0] or re.search(r"\binitKernel", r[0])]
# frames start with a tree build; a frame has 2 trees (source, recon): windows = every second initKernel
starts = first[0::2]
lo, hi = starts[1], starts[2] if len(starts) > 2 else len(rows) - 1
t0, t1 = rows[lo][1], rows[hi][1]
busy = sum(e - s for n, s, e in rows[lo:hi])
print("# frame window %.2f ms, kernel time %.2f ms, %d dispatches" % ((t1 - t0) / 1e6, busy / 1e6, hi - lo))
gaps = []
for i in range(lo + 1, hi):
    g = rows[i][1] - max(r[2] for r in rows[max(lo, i - 8):i])
    if g > 40000:
        gaps.append((g, i))
print("# gaps > 40 us: %d, total %.2f ms" % (len(gaps), sum(g for g, _ in gaps) / 1e6))
for g, i in gaps:
    print("%8.1f us  after %-34s before %s" % (g / 1e3, nm(rows[i - 1][0]), nm(rows[i][0])))
PY
cat $OUT/${1:-r05}_gaps_one_frame.txt
rm -rf $OUT/prof_solo
