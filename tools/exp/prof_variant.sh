#!/bin/bash
# kernel-trace stats of one bench variant:  prof_variant.sh <tag> <bench args...>
set -o pipefail
export TMPDIR=/tmp
O=gpurun_out/prof
mkdir -p $O
tag=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o $tag -- python3 bench.py --no-cpu --no-others --no-training-step "$@" > $O/${tag}_bench.log 2>&1 || exit 1
DB=$(find $O -name "*${tag}_results.db" | head -1)
python3 profiles/summarize_rocpd.py $DB > $O/${tag}_kernel_stats.txt || exit 1
rm -f $DB
