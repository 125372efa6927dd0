#!/bin/bash
# round 3: both towers' weight gradients in ONE launch with per-problem XCD shares vs one launch per tower (same box, two repetitions)
set -e
OUT=gpurun_out/r3k
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py tests/test_round2_gpu.py -m gpu -x -q -k "gemm or config3" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
for rep in 1 2; do
  for m in 0 1; do
    MEMEHIP_WGRAD_ONE_LAUNCH=$m timeout -k 10 200 python bench.py $B > $OUT/bench_m${m}_$rep.log 2>&1
    echo "one_launch=$m rep $rep: $(tail -1 $OUT/bench_m${m}_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
  done
done
MEMEHIP_WGRAD_ONE_LAUNCH=1 MEMEHIP_GEMM_XCD_BALANCE=0 timeout -k 10 200 python bench.py $B > $OUT/bench_nobal.log 2>&1
echo "one launch, no balance: $(tail -1 $OUT/bench_nobal.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"])')"
