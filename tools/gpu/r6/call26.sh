#!/bin/bash
# round 6, call 26 (diagnostic, results of the run are wrong by construction): pairSelectKernel without its global atomics
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
db() { find "$1" -name "*_results.db" | head -1; }
ENC="python $REPO/tools/gpu/r6/first_pass.py --config longdress --frames 1 --workers 1 --sets 1 --passes 3 --gen-procs 1 --capacity-h 2304"
for e in 0 1; do
cd /tmp; rm -rf $O/prof_enc; TMC2_ORIENT_EXPERIMENT=$e timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_enc -- $ENC > $O/r06c26_enc_$e.log 2>&1; cd $REPO
python profiles/summarise_rocpd.py "$(db $O/prof_enc)" "x" > $O/r06c26_kernel_stats_$e.txt
echo "== experiment $e"; grep -i "pairSelect\|scatterCompact" $O/r06c26_kernel_stats_$e.txt
rm -rf $O/prof_enc
done
