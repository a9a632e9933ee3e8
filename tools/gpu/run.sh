cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 500 python -X faulthandler -m pytest tests/test_gpu_images.py tests/test_gpu_segmenter.py -m gpu -q --timeout 150 2>&1 | tail -30) > gpurun_out/r02_pytest13.log 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-baseline 0 --tail 0 > gpurun_out/r02_bench8.json 2> gpurun_out/r02_bench8.err
cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /root/repo/gpurun_out/r02_prof_solo9 -- python /root/repo/bench.py --steps 3 --warmup 1 --frames 1 --workers 1 --gen-procs 1 --cpu-baseline 0 --tail 0 > /root/repo/gpurun_out/r02_prof_solo9.log 2>&1
cd /root/repo
DB=$(find gpurun_out/r02_prof_solo9 -name '*.db' | head -1)
python profiles/summarise_rocpd.py $DB "bench.py --steps 3 --warmup 1 --frames 1 --workers 1 (one frame in flight, device k-d trees)" > gpurun_out/r02_solo9_kernels.txt 2>&1
rm -rf gpurun_out/r02_prof_solo9
