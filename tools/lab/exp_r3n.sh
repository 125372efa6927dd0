#!/bin/bash
# round 3: the side-stream Adam slices with a capped grid (a thinner, longer stream of HBM traffic beside the backward's GEMMs)
OUT=gpurun_out/r3n
mkdir -p $OUT
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
for rep in 1 2; do
  for c in 0 4096 8192 16384; do
    MEMEHIP_ADAM_BLOCKS=$c timeout -k 10 200 python bench.py $B > $OUT/bench_c${c}_$rep.log 2>&1
    echo "adam blocks cap=$c rep $rep: $(tail -1 $OUT/bench_c${c}_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
