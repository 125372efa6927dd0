#!/bin/bash
# dropout index tables in LDS (attention) + BatchNorm finish reductions with 8 loads in flight: full GPU suite, reference-dropout A/B, config-2 A/B
mkdir -p gpurun_out/s2j
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s2j/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/s2j/tests.log)"; grep -n "^E  \|FAILED" gpurun_out/s2j/tests.log | head -10 | cut -c1-300
bash tools/lab/ab.sh s2j_drop 2 --no-extras --no-cpu-baseline --reference-dropout --steps 60 --warmup 10 -- baseline= -- head=
bash tools/lab/ab.sh s2j_c2 2 --config 2 --steps 100 --warmup 10 -- baseline= -- head=
