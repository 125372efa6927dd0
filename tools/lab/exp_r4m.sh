#!/bin/bash
# round 4, m: TIMING PROBE -- the layer slices' Adam launches beside the FORWARD (of the next step) instead of beside the backward
mkdir -p gpurun_out/r4m
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
show() { tail -1 $1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])'; }
for rep in 1 2 3; do
  timeout -k 10 200 python bench.py $B > gpurun_out/r4m/head_$rep.log 2>&1; echo "HEAD            $rep: $(show gpurun_out/r4m/head_$rep.log)"
  MEMEHIP_PROBE_ADAM_IN_FWD=1 timeout -k 10 200 python bench.py $B > gpurun_out/r4m/fwd_$rep.log 2>&1; echo "Adam beside fwd $rep: $(show gpurun_out/r4m/fwd_$rep.log)"
  MEMEHIP_PROBE_SKIP_SLICE_ADAM=1 timeout -k 10 200 python bench.py $B > gpurun_out/r4m/skip_$rep.log 2>&1; echo "no slice Adam   $rep: $(show gpurun_out/r4m/skip_$rep.log)"
done
