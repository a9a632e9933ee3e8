timeout 300 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 600 python bench.py --steps 2 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench_r01_b.json
