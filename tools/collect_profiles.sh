#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench command, FETCH_SIZE / WRITE_SIZE PMC passes (own runs, eager
#   launches so every kernel is a separate dispatch), and the default bench line.
# Outputs under gpurun_out/$1/ ; tools/traffic_from_pmc.py + tools/pmc_summary.py turn them into profiles/.
set -e
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats/run -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch/run -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write/run -- python bench.py --steps 3 --warmup 1 --no-graph --no-cpu-baseline > $OUT/write.log 2>&1
echo "write done"
python bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log | cut -c1-300
