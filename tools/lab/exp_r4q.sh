#!/bin/bash
mkdir -p gpurun_out/r4q
timeout -k 10 600 python -m pytest tests/test_fused_adam_gpu.py -m gpu -x -q > gpurun_out/r4q/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r4q/tests.log)"; grep -n "^E  \|Error" gpurun_out/r4q/tests.log | head -10 | cut -c1-400
