#!/bin/bash
# same-box, step-level A/B: side stream (weight gradients + Adam slices) at default vs high stream priority
set -e
OUT=gpurun_out/r2v
mkdir -p $OUT
B="--no-extras --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout -k 10 300 python bench.py $B $EXTRA > $OUT/$name.log 2>&1; echo "$name: $(tail -1 $OUT/$name.log | grep -o '"ms_per_step": [0-9.]*')"; }
for rep in 1 2; do
  EXTRA="" run c3_default_$rep MEMEHIP_SIDE_PRIORITY=0
  EXTRA="" run c3_sidehigh_$rep MEMEHIP_SIDE_PRIORITY=-1
done
for rep in 1 2; do
  EXTRA="--config 5 --steps 5 --warmup 2" run c5_default_$rep MEMEHIP_SIDE_PRIORITY=0
  EXTRA="--config 5 --steps 5 --warmup 2" run c5_sidehigh_$rep MEMEHIP_SIDE_PRIORITY=-1
done
