cd /root/repo
export TMPDIR=/tmp
(time timeout 900 python -m pytest tests/test_gpu_segmenter.py -m gpu -q -x --timeout 600  2>&1 | tail -5) > gpurun_out/r02_pytest6.log 2>&1
TMC2_REFINE_TIMING=1 timeout 400 python bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --kdtree device --cpu-baseline 0 --tail 0 > gpurun_out/r02_solo5.json 2> gpurun_out/r02_solo5.err
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02_prof_solo5 -- python /root/repo/bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --kdtree device --gen-procs 1 --cpu-baseline 0 --tail 0 > /root/repo/gpurun_out/r02_prof_solo5.log 2>&1
cd /root/repo
DB=$(find gpurun_out/r02_prof_solo5 -name '*.db' | head -1)
python profiles/summarise_rocpd.py $DB "bench.py --steps 2 --warmup 1 --frames 1 --workers 1 --kdtree device" > gpurun_out/r02_solo5_kernels.txt 2>&1
rm -rf gpurun_out/r02_prof_solo5
