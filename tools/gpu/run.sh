cd /root/repo
export TMPDIR=/tmp
(time timeout -s ABRT 300 python -X faulthandler -m pytest tests/test_gpu_images.py -m gpu -q --timeout 150 -k "single_rendezvous" 2>&1 | tail -8) > gpurun_out/r02_pytest15.log 2>&1
for r in one phases one phases; do
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-baseline 0 --tail 0 --rendezvous $r >> gpurun_out/r02_bench11_$r.json 2>> gpurun_out/r02_bench11_$r.err
done
