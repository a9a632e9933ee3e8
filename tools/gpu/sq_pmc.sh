#!/bin/bash
# SQ counters of the k-NN kernels, one frame in flight: where do the wave cycles go?
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp; mkdir -p $OUT
SOLO="python $REPO/bench.py --config ${2:-loot} --decoder 0 --steps 1 --warmup 0 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
rm -rf $OUT/pmc_sq; timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/pmc_sq -- $SOLO > $OUT/sq_pmc.log 2>&1
python - $OUT/pmc_sq "${1:-neighbourhoodKernel|reverseRowsKernel|sweepKernel|closureKernel}" <<'PY'
import csv, glob, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if not re.search(sys.argv[2], k): continue
        k = re.sub(r"\(.*", "", k.replace("tmc2::(anonymous namespace)::", "").replace("void ", ""))[:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES": cnt[k] += 1
for k, c in acc.items():
    n = max(cnt[k], 1)
    w = c["SQ_WAVE_CYCLES"]
    print("%-40s launches %d  waves %.0f  wave_cycles %.3g | parked %.0f%%  issue-stall %.0f%%  active %.0f%% | per wave: VALU %.0f  VMEM %.0f  LDS %.0f  quad-cycles %.0f"
          % (k, n, c["SQ_WAVES"] / n, w / n, 100 * c["SQ_WAIT_ANY"] / w, 100 * c["SQ_WAIT_INST_ANY"] / w, 100 * c["SQ_ACTIVE_INST_ANY"] / w,
             c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c["SQ_INSTS_VMEM"] / c["SQ_WAVES"], c["SQ_INSTS_LDS"] / c["SQ_WAVES"], w / c["SQ_WAVES"]))
PY
rm -rf $OUT/pmc_sq
