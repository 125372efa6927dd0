#!/bin/bash
set -e
OUT=gpurun_out/r2g
mkdir -p $OUT
for t in 1024 512 256 128; do
  MEMEHIP_WGRAD_TILES=$t timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > $OUT/c2_$t.log 2>&1 || { tail -30 $OUT/c2_$t.log; exit 1; }
  echo "tiles $t: $(tail -1 $OUT/c2_$t.log | cut -c100-230)"
done
