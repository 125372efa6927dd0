#!/bin/bash
set -e
OUT=gpurun_out/r2p
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py tests/test_packed_gpu.py tests/test_model_gpu.py tests/test_config5_gpu.py tests/test_round2_gpu.py -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
B="--no-extras --no-cpu-baseline"
timeout -k 10 200 python bench.py $B > $OUT/c3.log 2>&1; echo "config3: $(tail -1 $OUT/c3.log | cut -c100-200)"
timeout -k 10 300 python bench.py --config 5 --steps 5 --warmup 2 $B > $OUT/c5.log 2>&1; echo "config5: $(tail -1 $OUT/c5.log | cut -c100-230)"
