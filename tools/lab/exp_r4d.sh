#!/bin/bash
# round 4, experiment d: 16-bit epilogue operand of the direct GEMM by LDS-DMA into the idle stage (narrow) or by register loads after the K loop (opregs)
mkdir -p gpurun_out/r4d
P=multimodal_propaganda_meme_classification_amd
MEMEHIP_LIB_F16=$PWD/$P/libmemehip_lab_f16.so timeout -k 10 400 python tools/gemm_shapes.py --cands narrow,opregs,v4 > gpurun_out/r4d/shapes.log 2>&1 || { tail -30 gpurun_out/r4d/shapes.log; exit 1; }
cat gpurun_out/r4d/shapes.log
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
show() { tail -1 $1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], r["avg_launch_us"], {k:v["ms_per_step"] for k,v in r["all_gemm_kernels"].items()})'; }
for rep in 1 2; do
  (cd .ab_baseline && timeout -k 10 200 python bench.py $B > ../gpurun_out/r4d/base_$rep.log 2>&1); echo "baseline $rep: $(show gpurun_out/r4d/base_$rep.log)"
  MEMEHIP_GEMM_WIDE=0 MEMEHIP_GEMM_OPDMA=1 timeout -k 10 200 python bench.py $B > gpurun_out/r4d/dma_$rep.log 2>&1; echo "HEAD dma  $rep: $(show gpurun_out/r4d/dma_$rep.log)"
  MEMEHIP_GEMM_WIDE=0 MEMEHIP_GEMM_OPDMA=0 timeout -k 10 200 python bench.py $B > gpurun_out/r4d/regs_$rep.log 2>&1; echo "HEAD regs $rep: $(show gpurun_out/r4d/regs_$rep.log)"
done
