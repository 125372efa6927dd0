#!/bin/bash
# head kernels / colsum / exact-f32 GEMM with more loads in flight: full GPU suite, config 2 and config 3 A/B against the round's baseline
mkdir -p gpurun_out/s2i
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s2i/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/s2i/tests.log)"; grep -n "^E  \|FAILED" gpurun_out/s2i/tests.log | head -10 | cut -c1-300
bash tools/lab/ab.sh s2i_c2 2 --config 2 --steps 100 --warmup 10 -- baseline= -- head=
bash tools/lab/ab.sh s2i_c3 3 --no-extras --no-cpu-baseline --steps 60 --warmup 10 -- baseline= -- head=
