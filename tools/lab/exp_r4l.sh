#!/bin/bash
mkdir -p gpurun_out/r4l
timeout -k 10 700 python -m pytest tests/test_model_gpu.py tests/test_scaler_gpu.py tests/test_reference_run_gpu.py -m gpu -x -q -s > gpurun_out/r4l/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r4l/tests.log)"
grep -n "capture under the watchdog\|PG-DESTROY\|organizers.*bf16\|Kevin train loop" gpurun_out/r4l/tests.log | cut -c1-220
bash tools/ddp2_bench_rehearsal.sh gpurun_out/r4l
tail -5 gpurun_out/r4l/bench_w2_bf16.log | cut -c1-300
