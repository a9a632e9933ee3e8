#!/bin/bash
# round 6, call 08: the ENCODER's kernels alone (one frame, no metric / decoder legs): kernel stats and wave-slot weighting; then the whole GPU tier
export TMPDIR=/tmp; mkdir -p gpurun_out; O=$(pwd)/gpurun_out; REPO=$(pwd)
db() { find "$1" -name "*_results.db" | head -1; }
for cfg in longdress loot basketball; do
ENC="python $REPO/tools/gpu/r6/first_pass.py --config $cfg --frames 1 --workers 1 --sets 1 --passes 4 --gen-procs 1 --capacity-h 2304"
cd /tmp; rm -rf $O/prof_enc; timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_enc -- $ENC > $O/r06c08_enc_$cfg.log 2>&1; cd $REPO
python profiles/summarise_rocpd.py "$(db $O/prof_enc)" "$ENC  (one frame in flight, the path S1-S22 only, 4 passes)" > $O/r06_kernel_stats_encoder_one_frame_$cfg.txt
python profiles/occupancy_rocpd.py "$(db $O/prof_enc)" 4 > $O/r06_occupancy_encoder_one_frame_$cfg.txt
head -3 $O/r06_occupancy_encoder_one_frame_$cfg.txt | tail -2
rm -rf $O/prof_enc
done
timeout -k 10 1500 python -m pytest tests -x -q -m gpu > $O/r06c08_gpu_tier.log 2>&1; tail -4 $O/r06c08_gpu_tier.log
