#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4x
L=$PWD/multimodal_propaganda_meme_classification_amd
MEMEHIP_LIB=$L/libmemehip_lab.so MEMEHIP_LIB_F16=$L/libmemehip_lab_f16.so timeout -k 10 600 python -m pytest tests/test_gemm_exact_gpu.py -m gpu -x -q > gpurun_out/r4x/gemm_tests_lab.log 2>&1; rc=$?
tail -3 gpurun_out/r4x/gemm_tests_lab.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 600 python -m pytest tests/test_gemm_exact_gpu.py tests/test_kernels_gpu.py tests/test_model_gpu.py -m gpu -x -q > gpurun_out/r4x/gemm_tests.log 2>&1; rc=$?
tail -3 gpurun_out/r4x/gemm_tests.log
[ $rc -ne 0 ] && exit $rc
MEMEHIP_LIB=$L/libmemehip_lab.so MEMEHIP_LIB_F16=$L/libmemehip_lab_f16.so timeout -k 10 300 python tools/gemm_big_timeline.py > gpurun_out/r4x/big_timeline.log 2>&1 || { tail -20 gpurun_out/r4x/big_timeline.log; exit 1; }
timeout -k 10 300 python tools/gemm_big_timeline.py > gpurun_out/r4x/small_timeline.log 2>&1 || exit 1
cat gpurun_out/r4x/big_timeline.log gpurun_out/r4x/small_timeline.log
