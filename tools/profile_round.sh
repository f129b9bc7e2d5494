#!/bin/bash
# Round profile set (run on the GPU box from the repo root): kernel-trace stats + one-step timeline of
# the default bench command, the two PMC passes for its dominant kernel, and kernel-trace stats of C4/C5.
set -o pipefail
export TMPDIR=/tmp
R=${1:-r01}
O=gpurun_out/prof
mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o c2 -- python3 bench.py --no-cpu > $O/c2_bench.log 2>&1 || exit 1
DB=$(ls $O/*c2_results.db $O/*/*c2_results.db 2>/dev/null | head -1)
python3 profiles/summarize_rocpd.py $DB > $O/${R}_c2_bench_kernel_stats.txt || exit 1
python3 profiles/timeline_rocpd.py $DB step_kernel 1000 > $O/${R}_c2_step_timeline.txt || exit 1
grep "^{\"metric\"" $O/c2_bench.log | tail -1 > $O/${R}_bench_c2_under_rocprof.json
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o f --output-format csv -- python3 bench.py --no-cpu --steps 200 --warmup 50 > $O/pmc_f.log 2>&1 || exit 1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o w --output-format csv -- python3 bench.py --no-cpu --steps 200 --warmup 50 > $O/pmc_w.log 2>&1 || exit 1
python3 profiles/collect_pmc.py C2 step_kernel $O/pmc_fetch $O/pmc_write > $O/pmc_collect.log 2>&1 || exit 1
cp profiles/pmc_C2.json $O/pmc_C2.json
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o c4 -- python3 bench.py --no-cpu --config C4 --steps 300 --warmup 50 > $O/c4_bench.log 2>&1 || exit 1
DB4=$(ls $O/*c4_results.db $O/*/*c4_results.db 2>/dev/null | head -1)
python3 profiles/summarize_rocpd.py $DB4 > $O/${R}_c4_bench_kernel_stats.txt || exit 1
grep "^{\"metric\"" $O/c4_bench.log | tail -1 > $O/${R}_bench_c4_under_rocprof.json
for C in C3 C5; do
  c=$(echo $C | tr A-Z a-z)
  N=$([ $C = C5 ] && echo 60 || echo 500)
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o $c -- python3 bench.py --no-cpu --config $C --steps $N --warmup 20 --no-training-step > $O/${c}_bench.log 2>&1 || exit 1
  DBc=$(find $O -name "*${c}_results.db" | head -1)
  python3 profiles/summarize_rocpd.py $DBc > $O/${R}_${c}_bench_kernel_stats.txt || exit 1
  if [ $C = C3 ]; then python3 profiles/timeline_rocpd.py $DBc step_kernel 300 > $O/${R}_c3_step_timeline.txt || exit 1; fi
  rm -f $DBc
done
rm -f $O/*.db $O/*/*.db
echo done
