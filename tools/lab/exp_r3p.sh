#!/bin/bash
# round 3: conv weight gradients of consecutive layers in one launch (group size)
set -e
OUT=gpurun_out/r3p
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 400 python -m pytest tests/test_resnet_gpu.py tests/test_convgemm_gpu.py -m gpu -x -q > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for G in 1 2 3 4 6; do
  MEMEHIP_WGRAD_GROUP=$G timeout -k 10 200 python bench.py --config 2 --steps 30 --warmup 5 > $OUT/bench2_$G.log 2>&1
  echo "group $G: $(tail -1 $OUT/bench2_$G.log | cut -c100-200)"
done
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2/run -- python bench.py --config 2 --steps 5 --warmup 2 > $OUT/stats2.log 2>&1
echo "stats done"
