#!/bin/bash
OUT=gpurun_out/r2k
mkdir -p $OUT
T="tests/test_packed_gpu.py::test_one_graph_serves_batches_with_different_masks"
run() { name=$1; shift; timeout -k 10 400 python -X faulthandler -m pytest "$@" -m gpu -x -q > $OUT/$name.log 2>&1; echo "$name: rc=$? $(tail -1 $OUT/$name.log | cut -c1-60)"; }
run nostream tests/test_config5_gpu.py tests/test_model_gpu.py $T -k "not stream"
run nosegments tests/test_config5_gpu.py tests/test_model_gpu.py $T -k "not segments"
