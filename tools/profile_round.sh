#!/bin/bash
# Round profile set (run on the GPU box from the repo root): kernel-trace stats + one-step timeline of
# the default bench command (C2) and of C2 with the launcher's options, the two PMC passes for the dominant
# kernel of C2 and of C4, kernel-trace stats + timeline of C4, kernel-trace stats of C3/C5.
set -o pipefail
export TMPDIR=/tmp
R=${1:-r03}
O=gpurun_out/prof
mkdir -p $O
prof() {  # prof <tag> <bench args...>: kernel trace -> $O/<tag>_results.db
  local tag=$1; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O -o $tag -- python3 bench.py --no-cpu --no-others "$@" > $O/${tag}_bench.log 2>&1 || return 1
  find $O -name "*${tag}_results.db" | head -1
}
DB=$(prof c2) || exit 1
python3 profiles/summarize_rocpd.py $DB > $O/${R}_c2_bench_kernel_stats.txt || exit 1
python3 profiles/timeline_rocpd.py $DB step_kernel 1000 > $O/${R}_c2_step_timeline.txt || exit 1
grep "^{\"metric\"" $O/c2_bench.log | tail -1 > $O/${R}_bench_c2_under_rocprof.json
rm -f $DB
DBL=$(prof c2l --launcher) || exit 1
python3 profiles/summarize_rocpd.py $DBL > $O/${R}_c2_launcher_kernel_stats.txt || exit 1
python3 profiles/timeline_rocpd.py $DBL targets_kernel_ln 1000 > $O/${R}_c2_launcher_step_timeline.txt || exit 1
grep "^{\"metric\"" $O/c2l_bench.log | tail -1 > $O/${R}_bench_c2_launcher_under_rocprof.json
rm -f $DBL
for P in "C2 step_kernel" "C4 roles_kernel" "C5 gemm_kernel"; do
  set -- $P
  c=$(echo $1 | tr A-Z a-z)
  timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$c -o f --output-format csv -- python3 bench.py --no-cpu --no-others --config $1 --steps $([ $1 = C5 ] && echo 30 || echo 200) --warmup $([ $1 = C5 ] && echo 6 || echo 50) --no-training-step > $O/pmc_f_$c.log 2>&1 || exit 1
  timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$c -o w --output-format csv -- python3 bench.py --no-cpu --no-others --config $1 --steps $([ $1 = C5 ] && echo 30 || echo 200) --warmup $([ $1 = C5 ] && echo 6 || echo 50) --no-training-step > $O/pmc_w_$c.log 2>&1 || exit 1
  python3 profiles/collect_pmc.py $1 $2 $O/pmc_fetch_$c $O/pmc_write_$c $([ $1 = C5 ] && echo apply_kernel) > $O/pmc_collect_$c.log 2>&1 || exit 1
  cp profiles/pmc_$1.json $O/pmc_$1.json
done
DB4=$(prof c4 --config C4 --steps 300 --warmup 50) || exit 1
python3 profiles/summarize_rocpd.py $DB4 > $O/${R}_c4_bench_kernel_stats.txt || exit 1
python3 profiles/timeline_rocpd.py $DB4 roles_kernel 100 > $O/${R}_c4_step_timeline.txt || exit 1
grep "^{\"metric\"" $O/c4_bench.log | tail -1 > $O/${R}_bench_c4_under_rocprof.json
rm -f $DB4
DB4L=$(prof c4l --config C4 --launcher --steps 300 --warmup 50 --no-training-step) || exit 1
python3 profiles/summarize_rocpd.py $DB4L > $O/${R}_c4_launcher_kernel_stats.txt || exit 1
python3 profiles/timeline_rocpd.py $DB4L targets_kernel 100 > $O/${R}_c4_launcher_step_timeline.txt || exit 1
rm -f $DB4L
for C in C3 C5; do
  c=$(echo $C | tr A-Z a-z)
  N=$([ $C = C5 ] && echo 60 || echo 500)
  DBc=$(prof $c --config $C --steps $N --warmup 20 --no-training-step) || exit 1
  python3 profiles/summarize_rocpd.py $DBc > $O/${R}_${c}_bench_kernel_stats.txt || exit 1
  if [ $C = C3 ]; then python3 profiles/timeline_rocpd.py $DBc step_kernel 300 > $O/${R}_c3_step_timeline.txt || exit 1; fi
  grep "^{\"metric\"" $O/${c}_bench.log | tail -1 > $O/${R}_bench_${c}_under_rocprof.json
  rm -f $DBc
done
rm -f $O/*.db $O/*/*.db
echo done
