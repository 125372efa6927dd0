#!/bin/bash
# first run of the 256x256 kernel: exact tests, then config-5 shapes with and without it
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4x
timeout -k 10 400 python -m pytest tests/test_gemm_exact_gpu.py -m gpu -x -q -k big > gpurun_out/r4x/big_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r4x/big_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/gemm_shapes.py --config 5 --cands product,torch --dtype fp16 --csv gpurun_out/r4x/c5_big.csv > gpurun_out/r4x/c5_big.log 2>&1 || exit 1
MEMEHIP_GEMM_BIG_MIN=0 timeout -k 10 300 python tools/gemm_shapes.py --config 5 --cands product --dtype fp16 --csv gpurun_out/r4x/c5_small.csv > gpurun_out/r4x/c5_small.log 2>&1 || exit 1
paste -d, gpurun_out/r4x/c5_big.csv gpurun_out/r4x/c5_small.csv | head -40
