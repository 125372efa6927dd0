#!/bin/bash
# is test_ddp_segmented_graph_path_world1[stream-None] timing-dependent at the baseline build too?  3 runs each, same box
mkdir -p gpurun_out/s2d
B=$PWD/.ab_baseline/multimodal_propaganda_meme_classification_amd
for i in 1 2 3; do
  timeout -k 10 120 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "test_ddp_segmented_graph_path_world1" > gpurun_out/s2d/head_$i.log 2>&1; echo "head $i rc=$? $(tail -1 gpurun_out/s2d/head_$i.log)"
  MEMEHIP_LIB=$B/libmemehip.so MEMEHIP_LIB_F16=$B/libmemehip_f16.so timeout -k 10 120 python -m pytest tests/test_model_gpu.py -m gpu -x -q -k "test_ddp_segmented_graph_path_world1" > gpurun_out/s2d/base_$i.log 2>&1; echo "base $i rc=$? $(tail -1 gpurun_out/s2d/base_$i.log)"
done
timeout -k 10 200 python tools/ln_probe.py > gpurun_out/s2d/ln_probe.log 2>&1; tail -12 gpurun_out/s2d/ln_probe.log
