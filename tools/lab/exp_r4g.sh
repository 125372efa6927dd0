#!/bin/bash
# round 4, experiment g: hybrid dispatch -- round 3's LDS-staged-epilogue kernel for forward / dgrad, the direct kernel (4 x 2 waves of 32x64) for the weight gradients
mkdir -p gpurun_out/r4g
P=$PWD/multimodal_propaganda_meme_classification_amd
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
show() { tail -1 $1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], r["avg_launch_us"], {k:v["ms_per_step"] for k,v in r["all_gemm_kernels"].items()})'; }
for rep in 1 2 3; do
  (cd .ab_baseline && timeout -k 10 200 python bench.py $B > ../gpurun_out/r4g/base_$rep.log 2>&1); echo "baseline $rep: $(show gpurun_out/r4g/base_$rep.log)"
  MEMEHIP_LIB_F16=$P/libmemehip_lab_f16.so MEMEHIP_GEMM_VARIANT=11 timeout -k 10 200 python bench.py $B > gpurun_out/r4g/hyb_$rep.log 2>&1; echo "hybrid   $rep: $(show gpurun_out/r4g/hyb_$rep.log)"
  MEMEHIP_LIB_F16=$P/libmemehip_lab_f16.so MEMEHIP_GEMM_VARIANT=4 timeout -k 10 200 python bench.py $B > gpurun_out/r4g/v4_$rep.log 2>&1; echo "lab v4   $rep: $(show gpurun_out/r4g/v4_$rep.log)"
done
