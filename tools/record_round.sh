#!/bin/bash
# Everything profiles/rNN_* is made from, in one GPU-box call:  gpurun --timeout 1200 -- bash tools/record_round.sh r03
# (kernel stats + PMC passes of the default command, VALU counters, the batch profile, one bench line per workload)
# afterwards, here: python tools/summarize_batch_profile.py r03; cp gpurun_out/r03/{summary.json,kernel_stats.csv} and gpurun_out/r03_*.json to profiles/
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
cd $R
bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile_round.log 2>&1
bash tools/pmc_valu.sh $tag > gpurun_out/${tag}_pmc_valu.log 2>&1
bash tools/profile_batch.sh > gpurun_out/${tag}_profile_batch.log 2>&1
cd $R
python3 bench.py > gpurun_out/${tag}_bench_default.json 2> gpurun_out/${tag}_bench_default.err
python3 bench.py --workload large > gpurun_out/${tag}_bench_large.json 2>/dev/null
python3 bench.py --workload large --near-guess > gpurun_out/${tag}_bench_large_near.json 2>/dev/null
python3 bench.py --workload large --extent 200 > gpurun_out/${tag}_bench_large200.json 2>/dev/null
python3 bench.py --workload batch --batch 64 > gpurun_out/${tag}_bench_batch64.json 2>/dev/null
python3 bench.py --workload batch --batch 256 > gpurun_out/${tag}_bench_batch256.json 2>/dev/null
python3 bench.py --workload mapbuild > gpurun_out/${tag}_bench_mapbuild_n1.json 2>/dev/null
python3 bench.py --workload pyramid > gpurun_out/${tag}_bench_pyramid.json 2>/dev/null
python3 tools/time_pair.py 2>/dev/null | tail -1 > gpurun_out/${tag}_pair_sequence.json
python3 tools/node_loop_profile.py 40 60000 node gpurun_out/${tag}_node_loop.json > gpurun_out/${tag}_node_loop.log 2>&1
python3 tools/node_loop_profile.py 40 60000 rosbag gpurun_out/${tag}_node_loop_rosbag.json > gpurun_out/${tag}_node_loop_rosbag.log 2>&1
bash tools/pmc_valu_split.sh $tag > gpurun_out/${tag}_pmc_valu_split.log 2>&1
python3 tools/time_k1_forms.py > gpurun_out/${tag}_k1_forms.jsonl 2>/dev/null
python3 tools/time_prefilter.py 2>/dev/null | tail -1 > gpurun_out/${tag}_prefilter.json
ls -la gpurun_out/${tag}_*.json
