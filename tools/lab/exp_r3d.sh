#!/bin/bash
# round 3: ResNet-50 on the implicit-GEMM convolutions -- parity tests, config-2 bench, kernel trace
set -e
OUT=gpurun_out/r3d
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_resnet_gpu.py tests/test_convgemm_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
timeout -k 10 200 python bench.py --config 2 --steps 20 --warmup 5 > $OUT/bench2.log 2>&1
tail -1 $OUT/bench2.log | cut -c1-250
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2/run -- python bench.py --config 2 --steps 5 --warmup 2 > $OUT/stats2.log 2>&1
echo "stats done"
