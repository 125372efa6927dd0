#!/bin/bash
# round 3: one-pass attention backward for the ViT heads -- A/B in one process, then the attention parity tests with it on
set -e
OUT=gpurun_out/r3h
mkdir -p $OUT
timeout -k 10 300 python tools/attn_onepass_ab.py > $OUT/ab.log 2>&1 || { tail -20 $OUT/ab.log; exit 1; }
cat $OUT/ab.log | grep "B="
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q -k "attn or attention or step or train or parity or oracle" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
