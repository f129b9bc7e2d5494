#!/bin/bash
# idle GPU between the dispatches of a C2 run (kernel trace): where the launch boundaries are
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o c2g -- python3 bench.py --no-cpu --no-others --no-training-step --steps 1024 --warmup 256 > $O/c2g_bench.log 2>&1 || exit 1
DB=$(find $O -name "*c2g_results.db" | head -1)
python3 profiles/gaps_rocpd.py $DB > $O/c2_gaps.txt || exit 1
rm -f $DB
