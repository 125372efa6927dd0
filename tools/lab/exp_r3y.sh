#!/bin/bash
# round 3: HBM-side traffic of the config-2 step's kernels (FETCH_SIZE / WRITE_SIZE passes, eager launches) + SQ counters
OUT=gpurun_out/r3y
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/fetch/run -- python bench.py --config 2 --steps 3 --warmup 1 --no-graph > $OUT/fetch.log 2>&1
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/write/run -- python bench.py --config 2 --steps 3 --warmup 1 --no-graph > $OUT/write.log 2>&1
echo "write done"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq/run -- python bench.py --config 2 --steps 2 --warmup 1 --no-graph > $OUT/sq.log 2>&1
echo "sq done"
