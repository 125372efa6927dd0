#!/bin/bash
# round 3: final collection -- GPU suite, config-3 profiles (tools/collect_profiles.sh), config-2 kernel stats
OUT=gpurun_out/r3q
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $OUT/gpu_suite.log 2>&1
tail -2 $OUT/gpu_suite.log
bash tools/collect_profiles.sh r3q
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats2/run -- python bench.py --config 2 --steps 5 --warmup 2 > $OUT/stats2.log 2>&1
echo "config 2 stats done"
