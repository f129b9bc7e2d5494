#!/bin/bash
# The bench lines kept under profiles/ (run on the GPU box from the repo root): the default run, the driver's own
# invocation, and the option variants of C2 / C4 / C5.  Output: gpurun_out/bench/<R>_bench_*.json
R=${1:-r03}
O=gpurun_out/bench
mkdir -p $O
run() {  # run <file tag> <bench args...>
  local tag=$1; shift
  timeout -k 10 900 python bench.py "$@" > $O/${R}_bench_${tag}.json 2> $O/${tag}.err || { echo "bench $tag failed"; tail -3 $O/${tag}.err; return 1; }
  cut -c1-170 $O/${R}_bench_${tag}.json
}
run default || exit 1
run driver_invocation --steps 20 --warmup 5 || exit 1
for C in C2 C4; do
  for V in launcher layernorm argparse-defaults; do
    run ${C}_${V} --config $C --$V --no-cpu --no-others --no-training-step || exit 1
  done
done
run C5_launcher --config C5 --launcher --no-cpu --no-others --no-training-step --steps 100 --warmup 20 || exit 1
echo done
