#!/bin/bash
# round 3: config 5 (CLIP ViT-L/14@336 + BERT-large) under the 256x128 GEMM variants (2 = ring, 3 = ping-pong) vs the default 128x128 (4)
OUT=gpurun_out/r3l
mkdir -p $OUT
for v in 4 3 2; do
  MEMEHIP_GEMM_VARIANT=$v timeout -k 10 280 python bench.py --config 5 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $OUT/bench5_v$v.log 2>&1
  echo "variant $v: $(tail -1 $OUT/bench5_v$v.log | cut -c1-260)"
done
