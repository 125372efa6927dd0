#!/bin/bash
set -e
OUT=gpurun_out/r2e
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_resnet_gpu.py tests/test_abi.py -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > $OUT/c2.log 2>&1 || { tail -30 $OUT/c2.log; exit 1; }
echo "config2: $(tail -1 $OUT/c2.log | cut -c1-200)"
