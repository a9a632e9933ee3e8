cd /root/repo
export TMPDIR=/tmp
timeout 900 python -X faulthandler bench.py > gpurun_out/bench_r02_b.json 2> gpurun_out/bench_r02_b.err
echo "rc=$?" >> gpurun_out/bench_r02_b.err
