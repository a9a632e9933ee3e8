#!/bin/bash
# round 6, call 16: answers through the context's page-locked lines (scan totals with carried words, the tree's counters and depth, the
# error flags), page-locked staging for the patch rounds' and the placement tables: the whole GPU tier, then a frame's copies, blit
# kernels, API calls and gaps, and the bench
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $O/r06c16_gpu_tier.log 2>&1; tail -4 $O/r06c16_gpu_tier.log
bash tools/gpu/gaps.sh r06c16 > /dev/null 2>&1; head -30 $O/r06c16_gaps_one_frame.txt
SOLO="python $REPO/bench.py --steps 1 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
cd /tmp
rm -rf $O/prof_solo; timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats -d $O/prof_solo -- $SOLO > $O/r06c16_copy_prof.log 2>&1
DB=$(find $O/prof_solo -name "*_results.db" | head -1)
python - "$DB" > $O/r06c16_copies.txt 2>&1 <<'PY'
import sqlite3, sys, re, collections, bisect
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name,start,end from kernels order by start").fetchall()
nm = lambda n: re.sub(r"\(.*", "", n.replace("tmc2::(anonymous namespace)::", "").replace("void ", ""))[:34]
cols = [r[1] for r in db.execute("pragma table_info(memory_copies)")]
si, ei, ni, zi = cols.index("start"), cols.index("end"), cols.index("name"), cols.index("size")
mc = db.execute("select * from memory_copies order by start").fetchall()
ev = [(r[1], r[2], nm(r[0]), None) for r in rows if "copyBuffer" not in r[0]]
starts = [e[0] for e in ev]
cnt = collections.Counter(); size = collections.defaultdict(int)
for m in mc:
    k = bisect.bisect_right(starts, m[si])
    prev = ev[k - 1][2] if k > 0 else "-"
    nxt = ev[k][2] if k < len(ev) else "-"
    key = (m[ni].replace("MEMORY_COPY_", ""), prev, nxt)
    cnt[key] += 1; size[key] += m[zi]
print("# 2 passes of one frame (warm-up + timed): %d memory copies, %d copyBuffer kernels, %d kernels in all" % (len(mc), sum("copyBuffer" in r[0] for r in rows), len(rows)))
for key, c in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print("%4d x %-18s %9d B each  after %-34s before %s" % (c, key[0], size[key] // c, key[1], key[2]))
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
head -60 $O/r06c16_copies.txt | cut -c1-170
rm -rf $O/prof_solo
# HIP API calls of the same two passes
rm -rf $O/prof_api; timeout 600 rocprofv3 --hip-trace --stats -d $O/prof_api -- $SOLO > $O/r06c16_api.log 2>&1
DB=$(find $O/prof_api -name "*_results.db" | head -1)
python - "$DB" > $O/r06c16_hip_api_one_frame.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, count(*), sum(end-start)/1e6, max(end-start)/1e6 from regions group by name order by 2 desc limit 25").fetchall()
print("# HIP API calls of the process: set-up + 2 passes of one frame (bench.py --frames 1 --workers 1 --steps 1 --warmup 1)")
for r in rows:
    print("%-40s calls %7d total %10.1f ms max %8.2f ms" % (r[0][:40], r[1], r[2], r[3]))
PY
head -14 $O/r06c16_hip_api_one_frame.txt
rm -rf $O/prof_api
cd $REPO
B="python $REPO/bench.py --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0 --gen-procs 8"
timeout 600 $B --steps 10 --warmup 3 > $O/r06c16_bench.json 2> $O/r06c16_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r06c16_bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "verified", d["verified"], "proxy", d["per_rank_proxy"]["ms"], "first", d["first_gof_ms"])
PY
