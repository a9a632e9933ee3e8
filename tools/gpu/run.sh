cd /root/repo
export TMPDIR=/tmp
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/p_solo -- python /root/repo/bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 > /root/repo/gpurun_out/p_solo.log 2>&1
cd /root/repo
python profiles/summarise_rocpd.py $(find gpurun_out/p_solo -name '*.db' | head -1) "solo" > gpurun_out/r02_solo12_kernels.txt 2>&1
rm -rf gpurun_out/p_solo
