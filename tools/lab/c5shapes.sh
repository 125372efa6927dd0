#!/bin/bash
mkdir -p gpurun_out/r4w
P=$PWD/multimodal_propaganda_meme_classification_amd
MEMEHIP_LIB_F16=$P/libmemehip_lab_f16.so timeout -k 10 600 python tools/gemm_shapes.py --config 5 --rounds 5 --cands product,v2,v3,v12,torch --csv gpurun_out/r4w/gemm_shapes_config5.csv > gpurun_out/r4w/shapes5.log 2>&1; echo rc=$?
grep -v amdgpu.ids gpurun_out/r4w/shapes5.log | cut -c1-260
timeout -k 10 300 python -m pytest tests/test_config5_gpu.py -m gpu -x -q -s 2>&1 | grep -E "config 5|passed|failed" | cut -c1-200
