#!/bin/bash
set -e
OUT=gpurun_out/r2b
mkdir -p $OUT
B="--no-extras --no-cpu-baseline"
timeout -k 10 600 python -m pytest tests/test_model_gpu.py -x -q -k "ddp" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -3 $OUT/tests.log
timeout -k 10 200 python bench.py $B > $OUT/single.log 2>&1
echo "single: $(tail -1 $OUT/single.log | cut -c1-170)"
timeout -k 10 200 python bench.py $B --force-ddp --ddp-mode stream > $OUT/ddp_stream.log 2>&1
echo "ddp stream: $(tail -1 $OUT/ddp_stream.log | cut -c1-170)"
timeout -k 10 200 python bench.py $B --force-ddp --ddp-mode segments > $OUT/ddp_seg.log 2>&1
echo "ddp segments: $(tail -1 $OUT/ddp_seg.log | cut -c1-170)"
timeout -k 10 200 python bench.py $B --force-ddp --ddp-mode stream --ddp-compress bf16 > $OUT/ddp_stream_bf16.log 2>&1
echo "ddp stream bf16: $(tail -1 $OUT/ddp_stream_bf16.log | cut -c1-170)"
