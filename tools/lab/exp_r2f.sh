#!/bin/bash
set -e
OUT=gpurun_out/r2f
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline > $OUT/c2.log 2>&1 || { tail -30 $OUT/c2.log; exit 1; }
echo "config2: $(tail -1 $OUT/c2.log | cut -c1-1200)"
timeout -k 10 300 python bench.py --config 2 --no-cpu-baseline --no-graph > $OUT/c2e.log 2>&1 || { tail -30 $OUT/c2e.log; exit 1; }
echo "config2 eager: $(tail -1 $OUT/c2e.log | cut -c1-300)"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2stats/run -- python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > $OUT/c2stats.log 2>&1
echo "c2 stats done"
