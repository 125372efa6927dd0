#!/bin/bash
# round 4, i: deterministic guarded Adam (ordinals), clip_scaled_gradients against the reference's default-branch fixture
mkdir -p gpurun_out/r4i
timeout -k 10 600 python -m pytest tests/test_scaler_gpu.py tests/test_kernels_gpu.py "tests/test_reference_run_gpu.py::test_kevin_default_fp16_branch_clips_the_scaled_gradients" "tests/test_reference_run_gpu.py::test_kevin_train_test_evaluate_match_the_reference_run" -x -q -s > gpurun_out/r4i/tests.log 2>&1; rc=$?
grep -E "^\[Kevin default|passed|failed|Error|error|assert" gpurun_out/r4i/tests.log | tail -30
exit $rc
