mkdir -p gpurun_out
set -e
python bench.py --workload pyramid > gpurun_out/pyr_n1.json 2> gpurun_out/pyr_n1.err
timeout -k 10 500 python bench.py --gpus 2 --rehearse-one-gpu --workload pyramid --steps 2 --warmup 1 > gpurun_out/pyr_n2.json 2> gpurun_out/pyr_n2.err
