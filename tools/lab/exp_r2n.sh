#!/bin/bash
set -e
mkdir -p gpurun_out/r2n
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_model_gpu.py tests/test_resnet_gpu.py tests/test_config5_gpu.py -x -q > gpurun_out/r2n/tests.log 2>&1 || { tail -30 gpurun_out/r2n/tests.log; exit 1; }
tail -2 gpurun_out/r2n/tests.log
