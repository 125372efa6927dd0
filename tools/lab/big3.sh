#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4x
timeout -k 10 600 python -m pytest tests/test_gemm_exact_gpu.py tests/test_kernels_gpu.py -m gpu -x -q > gpurun_out/r4x/gemm_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4x/gemm_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python tools/gemm_big_timeline.py > gpurun_out/r4x/big_timeline.log 2>&1 || { tail -20 gpurun_out/r4x/big_timeline.log; exit 1; }
MEMEHIP_GEMM_BIG_MIN=0 timeout -k 10 300 python tools/gemm_big_timeline.py > gpurun_out/r4x/small_timeline.log 2>&1 || exit 1
cat gpurun_out/r4x/big_timeline.log; echo; cat gpurun_out/r4x/small_timeline.log
bash tools/lab/ab.sh r4x_ab 3 --no-extras --no-cpu-baseline --steps 60 --warmup 10 -- baseline= -- head=
