#!/bin/bash
# kernel timeline of one step of C4 with the argparse defaults (return normalisation, PopArt, n-step 10)
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o c4d -- python3 bench.py --no-cpu --no-others --no-training-step --config C4 --argparse-defaults --steps 300 --warmup 50 > $O/c4d_bench.log 2>&1 || exit 1
DB=$(find $O -name "*c4d_results.db" | head -1)
python3 profiles/timeline_rocpd.py $DB targets_kernel 100 > $O/c4_defaults_step_timeline.txt || exit 1
python3 profiles/summarize_rocpd.py $DB > $O/c4_defaults_kernel_stats.txt || exit 1
rm -f $DB
