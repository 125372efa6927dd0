#!/bin/bash
# round-2 experiments: config-5 GEMM variants; kernel stats of configs 2 and 5
set -e
OUT=gpurun_out/r2a
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline"
for v in 4 3 2; do
  MEMEHIP_GEMM_VARIANT=$v timeout -k 10 240 python bench.py --config 5 --steps 5 --warmup 2 $B > $OUT/c5_v$v.log 2>&1
  echo "config5 variant $v: $(tail -1 $OUT/c5_v$v.log | cut -c1-160)"
done
MEMEHIP_GEMM_WIDE=2 timeout -k 10 240 python bench.py --config 5 --steps 5 --warmup 2 $B > $OUT/c5_wide.log 2>&1
echo "config5 wide: $(tail -1 $OUT/c5_wide.log | cut -c1-160)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c5stats/run -- python bench.py --config 5 --steps 3 --warmup 1 --no-graph --no-overlap-wgrad --no-overlap-opt $B > $OUT/c5stats.log 2>&1
echo "c5 stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2stats/run -- python bench.py --config 2 --steps 5 --warmup 2 $B > $OUT/c2stats.log 2>&1
echo "c2 stats done"
