#!/bin/bash
# full GPU suite on HEAD (LDS-DMA attention staging + branch-free LayerNorm row loads), then the step A/B against the round's baseline
mkdir -p gpurun_out/s2e
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s2e/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/s2e/tests.log)"; grep -n "^E  \|FAILED" gpurun_out/s2e/tests.log | head -10 | cut -c1-300
bash tools/lab/ab.sh s2e_ab 3 --no-extras --no-cpu-baseline --steps 60 --warmup 10 -- baseline= -- head=
