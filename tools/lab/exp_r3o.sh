#!/bin/bash
# round 3: stream-K GEMM -- test, then time against the tile-per-workgroup kernel (one process), forced and by its own estimate
set -e
OUT=gpurun_out/r3o
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "gemm" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
SK_MODE=2 timeout -k 10 240 python tools/gemm_sk_check.py > $OUT/sk_check_forced.log 2>&1 || { tail -30 $OUT/sk_check_forced.log; exit 1; }
grep -v amdgpu.ids $OUT/sk_check_forced.log
