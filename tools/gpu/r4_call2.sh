#!/bin/bash
# round 4, second GPU call: the whole GPU tier (no early stop), the gcc-ASan bench, one-frame kernel traces of the voxels-of-2 and vox11 configs
mkdir -p gpurun_out; export TMPDIR=/tmp
REPO=$(pwd); O=$REPO/gpurun_out
timeout -k 10 1500 python -m pytest tests -m gpu -q --durations=12 > $O/r04_gpu_tests.log 2>&1; echo "rc=$?" >> $O/r04_gpu_tests.log
tail -n 22 $O/r04_gpu_tests.log
timeout -k 10 900 bash tools/asan_host_gcc.sh run python bench.py --steps 3 --warmup 1 --cpu-baseline 0 --pin 0 > $O/r04_asan_bench.json 2> $O/r04_asan_bench.err; echo "asan rc=$?" | tee -a $O/r04_asan_bench.err
tail -n 3 $O/r04_asan_bench.err
cd /tmp
db() { find "$1" -name "*_results.db" | head -1; }
for c in loot basketball; do
  SOLO="python $REPO/bench.py --config $c --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0 --decoder 0"
  rm -rf $O/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_solo -- $SOLO > $O/r04_prof_$c.log 2>&1
  python $REPO/profiles/summarise_rocpd.py "$(db $O/prof_solo)" "$SOLO  (one frame in flight)" > $O/r04_kernel_stats_one_frame_$c.txt
  rm -rf $O/prof_solo
  head -n 30 $O/r04_kernel_stats_one_frame_$c.txt
done
