#!/bin/bash
set -e
OUT=gpurun_out/r2m
mkdir -p $OUT
MEMEHIP_GEMM_VARIANT=8 timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "gemm or split" > $OUT/tests8.log 2>&1 || { tail -30 $OUT/tests8.log; exit 1; }
tail -2 $OUT/tests8.log
timeout -k 10 300 python tools/gemm_ab.py 4 8 > $OUT/ab8.log 2>&1 || { tail -20 $OUT/ab8.log; exit 1; }
cat $OUT/ab8.log
B="--no-extras --no-cpu-baseline"
MEMEHIP_GEMM_VARIANT=4 timeout -k 10 200 python bench.py $B > $OUT/b4.log 2>&1; echo "v4: $(tail -1 $OUT/b4.log | cut -c100-200)"
MEMEHIP_GEMM_VARIANT=8 timeout -k 10 200 python bench.py $B > $OUT/b8.log 2>&1; echo "v8: $(tail -1 $OUT/b8.log | cut -c100-200)"
