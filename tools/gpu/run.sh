cd /root/repo
REPO=$(pwd); OUT=$REPO/gpurun_out; export TMPDIR=/tmp
SOLO="python $REPO/bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 --ingest 0"
cd /tmp
rm -rf $OUT/prof_solo; timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_solo -- $SOLO > $OUT/r02_prof_solo.log 2>&1
python $REPO/profiles/occupancy_rocpd.py "$(find $OUT/prof_solo -name '*_results.db' | head -1)" 3 > $OUT/r02_occupancy_one_frame.txt
rm -rf $OUT/prof_solo
