#!/bin/bash
# C2 step timeline with the optimiser pass inside the reduction launch (and with the two launches beside it)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof
mkdir -p $O
for v in 1 0; do
  export SAMX_APPLY_IN_REDUCTION=$v
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o c2fa$v -- python3 bench.py --no-cpu --no-others --no-training-step --steps 1000 --warmup 100 > $O/c2fa${v}_bench.log 2>&1 || exit 1
  DB=$(find $O -name "*c2fa${v}_results.db" | head -1)
  python3 profiles/timeline_rocpd.py $DB step_kernel 500 > $O/c2_tail_timeline_$v.txt || exit 1
  python3 profiles/summarize_rocpd.py $DB > $O/c2_tail_stats_$v.txt || exit 1
  rm -f $DB
done
