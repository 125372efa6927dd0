#!/bin/bash
# Round 3: a process group PER TEST (the configuration that crashes a later hipGraphLaunch), torn down (a) with a bare
# destroy_process_group() as before, (b) through ddp.shutdown() = explicit close of GraphedSteps / reducers + gc first.  One run each.
OUT=gpurun_out/r3b
mkdir -p $OUT
T="tests/test_packed_gpu.py::test_one_graph_serves_batches_with_different_masks"
run() { name=$1; shift; timeout -k 10 400 python -X faulthandler -m pytest tests/test_config5_gpu.py tests/test_model_gpu.py $T -m gpu -x -q --deselect tests/test_model_gpu.py::test_ddp_two_ranks_on_one_gpu_match_the_global_batch > $OUT/$name.log 2>&1; echo "$name: rc=$? $(tail -1 $OUT/$name.log | cut -c1-80)"; }
MEMEHIP_DEBUG_PG_PER_TEST=1 run pg_per_test_with_shutdown
MEMEHIP_DEBUG_PG_PER_TEST=1 MEMEHIP_DEBUG_RAW_DESTROY=1 run pg_per_test_raw_destroy
