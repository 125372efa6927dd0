#!/bin/bash
# round 4, experiment e: wave shape of the direct GEMM: 2x4 waves of 64x32 (nj2) against 4x2 waves of 32x64 (nj4: whole 128-B lines per row in the epilogue)
mkdir -p gpurun_out/r4e
P=multimodal_propaganda_meme_classification_amd
for nj in 2 4; do
MEMEHIP_GEMM_NJ=$nj timeout -k 10 300 python -m pytest tests/test_gemm_exact_gpu.py tests/test_kernels_gpu.py -k "gemm" -x -q > gpurun_out/r4e/tests_$nj.log 2>&1 || { echo "tests failed nj=$nj"; tail -40 gpurun_out/r4e/tests_$nj.log; exit 1; }
tail -1 gpurun_out/r4e/tests_$nj.log
done
MEMEHIP_LIB_F16=$PWD/$P/libmemehip_lab_f16.so timeout -k 10 400 python tools/gemm_shapes.py --cands nj2,nj4,v4 > gpurun_out/r4e/shapes.log 2>&1 || { tail -30 gpurun_out/r4e/shapes.log; exit 1; }
cat gpurun_out/r4e/shapes.log
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
show() { tail -1 $1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], r["avg_launch_us"], {k:v["ms_per_step"] for k,v in r["all_gemm_kernels"].items()})'; }
for rep in 1 2; do
  (cd .ab_baseline && timeout -k 10 200 python bench.py $B > ../gpurun_out/r4e/base_$rep.log 2>&1); echo "baseline $rep: $(show gpurun_out/r4e/base_$rep.log)"
  MEMEHIP_GEMM_WIDE=0 MEMEHIP_GEMM_NJ=2 timeout -k 10 200 python bench.py $B > gpurun_out/r4e/nj2_$rep.log 2>&1; echo "HEAD nj2  $rep: $(show gpurun_out/r4e/nj2_$rep.log)"
  MEMEHIP_GEMM_WIDE=0 MEMEHIP_GEMM_NJ=4 timeout -k 10 200 python bench.py $B > gpurun_out/r4e/nj4_$rep.log 2>&1; echo "HEAD nj4  $rep: $(show gpurun_out/r4e/nj4_$rep.log)"
done
