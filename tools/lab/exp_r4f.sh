#!/bin/bash
# round 4, experiment f: rolled-loop epilogue (11 KB kernels instead of 42 / 78 KB): exactness, per-shape table, step A/B against the round's start
mkdir -p gpurun_out/r4f
P=multimodal_propaganda_meme_classification_amd
timeout -k 10 300 python -m pytest tests/test_gemm_exact_gpu.py tests/test_kernels_gpu.py -k "gemm" -x -q > gpurun_out/r4f/tests.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/r4f/tests.log; exit 1; }
tail -1 gpurun_out/r4f/tests.log
MEMEHIP_LIB_F16=$PWD/$P/libmemehip_lab_f16.so timeout -k 10 400 python tools/gemm_shapes.py --cands narrow,wide,v4 > gpurun_out/r4f/shapes.log 2>&1 || { tail -30 gpurun_out/r4f/shapes.log; exit 1; }
cat gpurun_out/r4f/shapes.log
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
show() { tail -1 $1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], r["avg_launch_us"], {k:v["ms_per_step"] for k,v in r["all_gemm_kernels"].items()})'; }
for rep in 1 2 3; do
  (cd .ab_baseline && timeout -k 10 200 python bench.py $B > ../gpurun_out/r4f/base_$rep.log 2>&1); echo "baseline $rep: $(show gpurun_out/r4f/base_$rep.log)"
  MEMEHIP_GEMM_WIDE=0 timeout -k 10 200 python bench.py $B > gpurun_out/r4f/w0_$rep.log 2>&1; echo "HEAD w=0  $rep: $(show gpurun_out/r4f/w0_$rep.log)"
  MEMEHIP_GEMM_WIDE=1 timeout -k 10 200 python bench.py $B > gpurun_out/r4f/w1_$rep.log 2>&1; echo "HEAD w=1  $rep: $(show gpurun_out/r4f/w1_$rep.log)"
done
