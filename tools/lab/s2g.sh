#!/bin/bash
# GEMM epilogue without the per-iteration s_waitcnt vmcnt(0): exactness tests, per-shape table vs the round's baseline, step A/B
mkdir -p gpurun_out/s2g
timeout -k 10 400 python -m pytest tests -m gpu -x -q -k "gemm or kernels or convgemm or round2" > gpurun_out/s2g/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/s2g/tests.log)"; grep -n "^E  \|FAILED" gpurun_out/s2g/tests.log | head -10 | cut -c1-300
timeout -k 10 300 python tools/gemm_shapes.py --cands product --dtype fp16 > gpurun_out/s2g/shapes_head.csv 2>&1
(cd .ab_baseline && timeout -k 10 300 python tools/gemm_shapes.py --cands product --dtype fp16 > ../gpurun_out/s2g/shapes_base.csv 2>&1)
paste -d' ' <(grep product gpurun_out/s2g/shapes_base.csv | cut -d, -f1,3) <(grep product gpurun_out/s2g/shapes_head.csv | cut -d, -f3)
bash tools/lab/ab.sh s2g_ab 3 --no-extras --no-cpu-baseline --steps 60 --warmup 10 -- baseline= -- head=
