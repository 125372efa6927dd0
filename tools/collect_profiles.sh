#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench command (graphed, overlapped) and of the sequential eager step the roofline
#   replay matches; FETCH_SIZE / WRITE_SIZE / SQ counter PMC passes (own runs, --kernel-trace only, eager launches so every
#   kernel is a separate dispatch); the default bench line.
# Outputs under gpurun_out/$1/ ; tools/traffic_from_pmc.py, tools/sq_summary.py turn them into profiles/.
set -e
TAG=${1:-prof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats/run -- python bench.py --steps 5 --warmup 2 $B > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_seq/run -- python bench.py --steps 5 --warmup 2 --no-graph --no-overlap-wgrad --no-overlap-opt $B > $OUT/stats_seq.log 2>&1
echo "sequential stats done"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch/run -- python bench.py --steps 3 --warmup 1 --no-graph $B > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write/run -- python bench.py --steps 3 --warmup 1 --no-graph $B > $OUT/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq/run -- python bench.py --steps 2 --warmup 1 --no-graph --no-overlap-wgrad --no-overlap-opt $B > $OUT/sq.log 2>&1
echo "sq done"
python bench.py > $OUT/bench_default.log 2>&1
tail -1 $OUT/bench_default.log | cut -c1-300
