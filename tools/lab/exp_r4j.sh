#!/bin/bash
# round 4, j: (1) the recorded crashing configuration of the round-2 hipGraphLaunch segfault (a process group per test), ONCE, at a HEAD whose
# captures are all thread-local (commit c392575) -- the test of that fix; (2) the parity measurements for profiles/
OUT=gpurun_out/r4j
mkdir -p $OUT
T="tests/test_packed_gpu.py::test_one_graph_serves_batches_with_different_masks"
MEMEHIP_DEBUG_PG_PER_TEST=1 timeout -k 10 500 python -X faulthandler -m pytest tests/test_config5_gpu.py tests/test_model_gpu.py $T -m gpu -x -q --deselect tests/test_model_gpu.py::test_ddp_two_ranks_on_one_gpu_match_the_global_batch > $OUT/pg_per_test.log 2>&1
echo "pg_per_test (thread-local captures): rc=$? $(tail -1 $OUT/pg_per_test.log | cut -c1-100)"
grep -n "Segmentation\|Fatal" $OUT/pg_per_test.log | head -3
timeout -k 10 900 python tools/publish_parity.py $OUT/parity.txt > $OUT/publish.log 2>&1
echo "publish_parity rc=$?"
grep -c "" $OUT/parity.txt
