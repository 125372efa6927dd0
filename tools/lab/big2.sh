#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4x
timeout -k 10 400 python -m pytest tests/test_gemm_exact_gpu.py -m gpu -x -q -k big > gpurun_out/r4x/big_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4x/big_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/gemm_big_timeline.py > gpurun_out/r4x/big_timeline.log 2>&1 || { tail -20 gpurun_out/r4x/big_timeline.log; exit 1; }
MEMEHIP_GEMM_BIG_MIN=0 timeout -k 10 300 python tools/gemm_big_timeline.py > gpurun_out/r4x/small_timeline.log 2>&1 || exit 1
cat gpurun_out/r4x/big_timeline.log; echo; cat gpurun_out/r4x/small_timeline.log
