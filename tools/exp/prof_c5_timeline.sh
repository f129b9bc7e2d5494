#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o c5t -- python3 bench.py --no-cpu --no-others --no-training-step --config C5 --steps 40 --warmup 10 > $O/c5t_bench.log 2>&1 || exit 1
DB=$(find $O -name "*c5t_results.db" | head -1)
python3 tools/exp/timeline_queues.py $DB d_input_partials_kernel 20 > $O/c5_queues_timeline.txt || exit 1
rm -f $DB
