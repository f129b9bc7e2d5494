run() { echo "== $*"; env "$@" timeout -k 10 100 python bench.py --config C5 --steps 200 --warmup 30 --no-cpu --no-others --no-training-step 2>&1 | grep -o '"value": [0-9.]*' ; }
run A=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=3
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_HIP_FORCE_GRAPH_QUEUES=8
run SAMX_SERIAL_ROLES=1
run A=1
