#!/bin/bash
# conv dgrad epilogue / split-K finish with batched operand loads, Adam with the four streams requested together:
# tests, config-2 A/B, config-3 A/B against the round's baseline
mkdir -p gpurun_out/s2h
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "resnet or convgemm or scaler or optim or adam or train or model_fp16 or reference_run" > gpurun_out/s2h/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/s2h/tests.log)"; grep -n "^E  \|FAILED" gpurun_out/s2h/tests.log | head -10 | cut -c1-300
bash tools/lab/ab.sh s2h_c2 3 --config 2 --steps 100 --warmup 10 -- baseline= -- head=
bash tools/lab/ab.sh s2h_c3 2 --no-extras --no-cpu-baseline --steps 60 --warmup 10 -- baseline= -- head=
