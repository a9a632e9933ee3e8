python -m pytest tests -m gpu -x -q 2>&1 | tail -5
python bench.py --steps 2 --warmup 1 --cpu-baseline 0 2>&1 | tail -1
python bench.py --steps 2 --warmup 1 --cpu-baseline 0 --workers 1 --frames 4 2>&1 | tail -1
