#!/bin/bash
# round 3: step-level A/B of the one-pass attention backward (same box, two repetitions each)
set -e
OUT=gpurun_out/r3i
mkdir -p $OUT
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
for rep in 1 2; do
  for m in 0 1 2; do
    MEMEHIP_ATTN_ONEPASS=$m timeout -k 10 200 python bench.py $B > $OUT/bench_m${m}_$rep.log 2>&1
    echo "onepass=$m rep $rep: $(tail -1 $OUT/bench_m${m}_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
