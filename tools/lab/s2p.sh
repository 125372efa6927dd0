#!/bin/bash
# the remaining hot __shfl_xor uses (attention row maximum / sums, conv BatchNorm epilogue, f64 BatchNorm finish) through the register-file partners: suite + A/B
mkdir -p gpurun_out/s2p
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s2p/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/s2p/tests.log)"; grep -n "^E  \|FAILED" gpurun_out/s2p/tests.log | head -10 | cut -c1-300
bash tools/lab/ab.sh s2p_c3 2 --no-extras --no-cpu-baseline --steps 60 --warmup 10 -- baseline= -- head=
bash tools/lab/ab.sh s2p_c2 2 --config 2 --steps 100 --warmup 10 -- baseline= -- head=
