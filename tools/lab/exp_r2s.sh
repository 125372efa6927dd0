#!/bin/bash
set -e
OUT=gpurun_out/r2s
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_resnet_gpu.py -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
MEMEHIP_RESNET_WGRAD_SIDE=0 timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > $OUT/c2_0.log 2>&1; echo "side=0: $(tail -1 $OUT/c2_0.log | cut -c100-230)"
MEMEHIP_RESNET_WGRAD_SIDE=1 timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > $OUT/c2_1.log 2>&1; echo "side=1: $(tail -1 $OUT/c2_1.log | cut -c100-230)"
