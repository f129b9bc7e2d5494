#!/bin/bash
# rows per graph launch with the two-launch row of the whole-network path (reduction + optimiser pass in one):
# C2 at 32 / 48 / 64 rows, C3 at 16 / 20 / 24, alternating, one box
set -o pipefail
run() {  # run <config> <rows>
  SAMX_STEPS_PER_LAUNCH=$2 timeout -k 10 200 python bench.py --config $1 --no-cpu --no-others --no-training-step --steps 4032 --warmup 384 \
    | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 rows=$2', round(d['value']), d['config']['steps_per_graph_launch'], d['config']['finite_params'])" || exit 1
}
for rep in 1 2; do
  for k in 32 48 64; do run C2 $k; done
  for k in 16 24 32; do run C3 $k; done
done
