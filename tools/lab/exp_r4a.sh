#!/bin/bash
# round 4, experiment a: the direct-epilogue GEMM (variant 10) -- exactness, then per-shape A/B against variant 4 and torch.matmul
set -o pipefail
mkdir -p gpurun_out/r4a
for v in 4 10; do
  MEMEHIP_GEMM_VARIANT=$v timeout -k 10 300 python -m pytest tests/test_gemm_exact_gpu.py tests/test_kernels_gpu.py -k "gemm" -x -q > gpurun_out/r4a/tests_v$v.log 2>&1 || { echo "tests failed for variant $v"; tail -30 gpurun_out/r4a/tests_v$v.log; exit 1; }
  tail -3 gpurun_out/r4a/tests_v$v.log
done
timeout -k 10 300 python tools/gemm_shapes.py --variants 4,10 --torch --csv gpurun_out/r4a/gemm_shapes.csv > gpurun_out/r4a/shapes.log 2>&1 || { tail -30 gpurun_out/r4a/shapes.log; exit 1; }
cat gpurun_out/r4a/shapes.log
for v in 4 10; do
  GEMM_VARIANT=$v timeout -k 10 120 python tools/gemm_timeline.py > gpurun_out/r4a/timeline_v$v.log 2>&1 || { tail -20 gpurun_out/r4a/timeline_v$v.log; exit 1; }
  grep -v "resident /" gpurun_out/r4a/timeline_v$v.log
done
