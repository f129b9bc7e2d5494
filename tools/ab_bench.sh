#!/bin/bash
# Alternating A/B of two builds of the library on ONE box: sam_tf_amd/libsamx_a.so against libsamx_b.so (copied over
# libsamx.so in turn; the original is restored).  usage: tools/ab_bench.sh <rounds> <bench args...>
N=${1:-3}; shift
cp sam_tf_amd/libsamx.so /tmp/libsamx_orig.so
for i in $(seq $N); do
  for v in a b; do
    cp sam_tf_amd/libsamx_$v.so sam_tf_amd/libsamx.so
    echo -n "$v "
    timeout -k 10 300 python bench.py --no-cpu --no-training-step "$@" 2>/dev/null | grep -o '"value": [0-9.]*' || { cp /tmp/libsamx_orig.so sam_tf_amd/libsamx.so; exit 1; }
  done
done
cp /tmp/libsamx_orig.so sam_tf_amd/libsamx.so
