#!/bin/bash
# round 5, call 13: where the copies of one frame sit (every copy of the trace by the kernels around it, counted)
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out
REPO=$(pwd); SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
rm -rf $O/prof_solo; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_solo -- $SOLO > $O/kd_prof.log 2>&1
DB=$(find $O/prof_solo -name "*_results.db" | head -1)
python - "$DB" > $O/r05c13_copies.txt 2>&1 <<'PY'
import sqlite3, sys, re, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
nm = lambda n: re.sub(r"\(.*", "", n.replace("tmc2::(anonymous namespace)::", "").replace("void ", ""))[:34]
cols = [r[1] for r in db.execute("pragma table_info(memory_copies)")]
si, ei, ni, zi = cols.index("start"), cols.index("end"), cols.index("name"), cols.index("size")
mc = db.execute("select * from memory_copies order by start").fetchall()
ev = [(r[1], r[2], nm(r[0]), None) for r in rows if "copyBuffer" not in r[0]]
import bisect
starts = [e[0] for e in ev]
cnt = collections.Counter(); size = collections.defaultdict(int)
for m in mc:
    k = bisect.bisect_right(starts, m[si])
    prev = ev[k - 1][2] if k > 0 else "-"
    nxt = ev[k][2] if k < len(ev) else "-"
    key = (m[ni].replace("MEMORY_COPY_", ""), prev, nxt)
    cnt[key] += 1; size[key] += m[zi]
print("# %d memory copies, %d copyBuffer kernels, %d lvInitKernel (2 per path pass; the metric builds 2 more per pass)" % (len(mc), sum("copyBuffer" in r[0] for r in rows), sum("lvInitKernel" in r[0] for r in rows)))
for key, c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%4d x %-18s %9d B each  after %-34s before %s" % (c, key[0], size[key] // c, key[1], key[2]))
# the blit kernels (device-to-host copies into page-locked memory and device-to-device copies run as __amd_rocclr_copyBuffer)
cnt = collections.Counter(); dur = collections.defaultdict(float)
for i, r in enumerate(rows):
    if "copyBuffer" not in r[0]:
        continue
    j = i - 1
    while j >= 0 and "copyBuffer" in rows[j][0]:
        j -= 1
    k = i + 1
    while k < len(rows) and "copyBuffer" in rows[k][0]:
        k += 1
    key = (nm(rows[j][0]) if j >= 0 else "-", nm(rows[k][0]) if k < len(rows) else "-")
    cnt[key] += 1; dur[key] += (r[2] - r[1]) / 1e3
print("# copyBuffer kernels by the kernels around them:")
for key, c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%4d x  avg %7.2f us  after %-34s before %s" % (c, dur[key] / c, key[0], key[1]))
PY
head -130 $O/r05c13_copies.txt | cut -c1-200
rm -rf $O/prof_solo
