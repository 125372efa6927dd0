#!/bin/bash
# round 4, experiment c: HEAD (direct-epilogue GEMM; wide tile off / by rule) against the round's starting point on ONE box, alternating.
# Needs the baseline next to the tree (git-ignored):  git worktree add -f .ab_baseline cd4bd55 && make -C .ab_baseline/multimodal_propaganda_meme_classification_amd/csrc -j8
OUT=gpurun_out/r4c
mkdir -p $OUT
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
show() { tail -1 $1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], r["avg_launch_us"], {k:v["ms_per_step"] for k,v in r["all_gemm_kernels"].items()})'; }
for rep in 1 2 3; do
  (cd .ab_baseline && timeout -k 10 200 python bench.py $B > ../$OUT/base_$rep.log 2>&1)
  echo "baseline  rep $rep: $(show $OUT/base_$rep.log)"
  MEMEHIP_GEMM_WIDE=0 timeout -k 10 200 python bench.py $B > $OUT/head_w0_$rep.log 2>&1
  echo "HEAD w=0  rep $rep: $(show $OUT/head_w0_$rep.log)"
  MEMEHIP_GEMM_WIDE=1 timeout -k 10 200 python bench.py $B > $OUT/head_w1_$rep.log 2>&1
  echo "HEAD w=1  rep $rep: $(show $OUT/head_w1_$rep.log)"
done
