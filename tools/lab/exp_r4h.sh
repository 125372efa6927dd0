#!/bin/bash
# round 4, experiment h: the per-shape GEMM table for profiles/ (product kernel, the direct / wide / ping-pong lab kernels, torch.matmul),
# then the whole GPU suite on the product build
mkdir -p gpurun_out/r4h
P=$PWD/multimodal_propaganda_meme_classification_amd
MEMEHIP_LIB_F16=$P/libmemehip_lab_f16.so timeout -k 10 500 python tools/gemm_shapes.py --cands product,v10,v12,v3,torch --csv gpurun_out/r4h/gemm_shapes.csv > gpurun_out/r4h/shapes.log 2>&1 || { tail -30 gpurun_out/r4h/shapes.log; exit 1; }
cat gpurun_out/r4h/shapes.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4h/tests.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/r4h/tests.log; exit 1; }
tail -3 gpurun_out/r4h/tests.log
