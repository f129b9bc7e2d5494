#!/bin/bash
set -o pipefail
run() {  # run <rows> <bench args...>
  local k=$1; shift
  SAMX_STEPS_PER_LAUNCH=$k timeout -k 10 200 python bench.py "$@" --no-cpu --no-others --no-training-step --steps 4096 --warmup 512 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$* rows=$k', round(d['value']), d['config']['steps_per_graph_launch'], d['config']['finite_params'])" || exit 1
}
for rep in 1 2; do
  for k in 64 128; do run $k --config C2; done
  for k in 16 32; do run $k --config C2 --launcher; done
  for k in 16 32; do run $k --config C2 --argparse-defaults; done
  for k in 16 32; do run $k --config C2 --layernorm; done
done
