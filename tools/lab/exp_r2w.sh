#!/bin/bash
# correctness of the in-tree attention build, then the same-box step-level A/B against an alternative build
set -e
OUT=gpurun_out/r2w
mkdir -p $OUT
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py tests/test_dropout_gpu.py tests/test_packed_gpu.py tests/test_model_gpu.py tests/test_config5_gpu.py tests/test_round2_gpu.py -x -q > $OUT/tests.log 2>&1 || { tail -40 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
bash tools/exp_r2t.sh ${1:-build_ab/libmemehip_slots1.so} ${2:-slots1}
