#!/bin/bash
# round 4, p: Adam fused into the weight-gradient epilogue: tests, then the step against the separate slices and the round's start, same box
mkdir -p gpurun_out/r4p
timeout -k 10 600 python -m pytest tests/test_fused_adam_gpu.py tests/test_scaler_gpu.py tests/test_gemm_exact_gpu.py -m gpu -x -q > gpurun_out/r4p/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r4p/tests.log)"; grep -n "^E  \|Error" gpurun_out/r4p/tests.log | head -10
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
show() { tail -1 $1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"]["final_loss"])'; }
for rep in 1 2 3; do
  (cd .ab_baseline && timeout -k 10 200 python bench.py $B > ../gpurun_out/r4p/base_$rep.log 2>&1); echo "baseline      $rep: $(show gpurun_out/r4p/base_$rep.log)"
  MEMEHIP_FUSE_ADAM=0 timeout -k 10 200 python bench.py $B > gpurun_out/r4p/sep_$rep.log 2>&1; echo "HEAD separate $rep: $(show gpurun_out/r4p/sep_$rep.log)"
  MEMEHIP_FUSE_ADAM=1 timeout -k 10 200 python bench.py $B > gpurun_out/r4p/fused_$rep.log 2>&1; echo "HEAD fused    $rep: $(show gpurun_out/r4p/fused_$rep.log)"
done
