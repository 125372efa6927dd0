#!/bin/bash
# round 3: LayerNorm -- forward 4 rows per wave vs 2, backward partial-sum swizzle; kernel tests first
set -e
OUT=gpurun_out/r3g
mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "layernorm or ln" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
MEMEHIP_LN_FWD_RPW=2 timeout -k 10 200 python tools/ln_probe.py > $OUT/probe_rpw2.log 2>&1
timeout -k 10 200 python tools/ln_probe.py > $OUT/probe_rpw4.log 2>&1
echo "--- rpw2"; grep -E "fwd|n_part   512" $OUT/probe_rpw2.log
echo "--- rpw4"; grep -E "fwd|n_part   512" $OUT/probe_rpw4.log
