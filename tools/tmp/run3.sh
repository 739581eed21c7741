set -e
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
timeout -k 10 200 python tools/tmp/t_align.py 2>&1 | tail -1
timeout -k 10 100 python bench.py --workload batch --batch 64 --steps 5 --warmup 2 2>&1 | tail -1 | cut -c80-140
timeout -k 10 100 python bench.py --workload large --steps 5 --warmup 2 2>&1 | tail -1 | cut -c80-140
