#!/bin/bash
# Round 3: the two halves of commit c0cc090 taken apart, ONE run each of the combination that used to crash
# (config-5 tests + test_model_gpu.py + the graph-replay test), at HEAD.
OUT=gpurun_out/r3a
mkdir -p $OUT
T="tests/test_packed_gpu.py::test_one_graph_serves_batches_with_different_masks"
run() { name=$1; shift; timeout -k 10 400 python -X faulthandler -m pytest tests/test_config5_gpu.py tests/test_model_gpu.py $T -m gpu -x -q > $OUT/$name.log 2>&1; echo "$name: rc=$? $(tail -1 $OUT/$name.log | cut -c1-80)"; }
MEMEHIP_DEBUG_PIN64=1 MEMEHIP_DEBUG_PG_PER_TEST=1 run both_reverted
MEMEHIP_DEBUG_PG_PER_TEST=1 run only_pg_per_test
MEMEHIP_DEBUG_PIN64=1 run only_pin64
run head
