#!/bin/bash
set -e
OUT=gpurun_out/r2h
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_heads_gpu.py tests/test_resnet_gpu.py -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
