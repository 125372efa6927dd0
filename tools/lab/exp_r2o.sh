#!/bin/bash
set -e
OUT=gpurun_out/r2o
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_model_fp16_gpu.py tests/test_round2_gpu.py tests/test_config5_gpu.py tests/test_dropout_gpu.py tests/test_packed_gpu.py -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
B="--no-extras --no-cpu-baseline"
MEMEHIP_GEMM_DERIV_AUX=0 timeout -k 10 200 python bench.py $B > $OUT/b0.log 2>&1; echo "deriv_aux=0: $(tail -1 $OUT/b0.log | cut -c100-200)"
MEMEHIP_GEMM_DERIV_AUX=1 timeout -k 10 200 python bench.py $B > $OUT/b1.log 2>&1; echo "deriv_aux=1: $(tail -1 $OUT/b1.log | cut -c100-200)"
python - <<'PY'
import json
for k in ("b0","b1"):
    d=json.loads(open(f"gpurun_out/r2o/{k}.log").read().strip().split("\n")[-1])
    print(k, d["roofline"]["all_gemm_kernels"], d["roofline"]["frac"])
PY
