#!/bin/bash
# round 3: upper bound of what fusing Adam into the weight-gradient epilogue could return -- the step with the matrix slices' Adam launches removed (timing only)
OUT=gpurun_out/r3m
mkdir -p $OUT
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
for rep in 1 2; do
  for m in 0 1; do
    MEMEHIP_PROBE_SKIP_SLICE_ADAM=$m timeout -k 10 200 python bench.py $B > $OUT/bench_m${m}_$rep.log 2>&1
    echo "skip_slice_adam=$m rep $rep: $(tail -1 $OUT/bench_m${m}_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
