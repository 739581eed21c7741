#!/bin/bash
# after a change to the latency kernels (GPU box): the whole GPU suite, the default bench line, align by scan size, the pair sequence
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/${1:-check}
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/gputest.log 2>&1; echo "rc=$?" >> $O/gputest.log; tail -n 3 $O/gputest.log
python bench.py > $O/bench.json 2> $O/bench.err
python3 -c "
import json; d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['protocol_us_per_evaluation'], d['body_us_per_evaluation'])"
python3 tools/time_align_sizes.py | tee $O/sizes.json
python3 tools/time_pair.py | tail -n 1 | tee $O/pair.json
