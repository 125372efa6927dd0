#!/bin/bash
set -e
OUT=gpurun_out/r2d
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline --steps 6 --warmup 2"
for m in segments stream; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$m/run -- python bench.py $B --force-ddp --ddp-mode $m > $OUT/$m.log 2>&1
  python tools/trace_idle.py $OUT/$m 5 > $OUT/idle_$m.txt 2>&1 || true
  cat $OUT/idle_$m.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/single/run -- python bench.py $B > $OUT/single.log 2>&1
python tools/trace_idle.py $OUT/single 5 > $OUT/idle_single.txt 2>&1 || true
cat $OUT/idle_single.txt
