#!/bin/bash
set -e
OUT=gpurun_out/r2r
mkdir -p $OUT
B="--no-extras --no-cpu-baseline --force-ddp"
for g in 2 3 4 6; do
  MEMEHIP_DDP_GROUP=$g timeout -k 10 200 python bench.py $B > $OUT/g$g.log 2>&1 || { tail -20 $OUT/g$g.log; exit 1; }
  echo "group $g: $(tail -1 $OUT/g$g.log | cut -c100-200)"
done
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline > $OUT/single.log 2>&1; echo "single: $(tail -1 $OUT/single.log | cut -c100-200)"
