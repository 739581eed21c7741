mkdir -p gpurun_out
set -e
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or batch or compaction or dump" > gpurun_out/k1_check_tests.log 2>&1
timeout -k 10 200 python tools/time_k1_1m.py > gpurun_out/k1_check_time.log 2>&1
timeout -k 10 200 python tools/fuzz_grid.py 40 > gpurun_out/k1_check_fuzz.log 2>&1
