#!/bin/bash
# round 4, experiment b: product GEMM = direct-epilogue kernels (128x128 + 128x256 dispatch); exactness, per-shape table incl. the
# round-3 default (lab v4) and torch.matmul; then the whole GPU suite's GEMM users and a bench line
set -o pipefail
mkdir -p gpurun_out/r4b
P=multimodal_propaganda_meme_classification_amd
timeout -k 10 300 python -m pytest tests/test_gemm_exact_gpu.py tests/test_kernels_gpu.py -k "gemm" -x -q > gpurun_out/r4b/tests.log 2>&1 || { echo "tests failed"; tail -40 gpurun_out/r4b/tests.log; exit 1; }
tail -2 gpurun_out/r4b/tests.log
MEMEHIP_LIB_F16=$PWD/$P/libmemehip_lab_f16.so timeout -k 10 400 python tools/gemm_shapes.py --cands narrow,wide,auto,v4,torch --csv gpurun_out/r4b/gemm_shapes.csv > gpurun_out/r4b/shapes.log 2>&1 || { tail -30 gpurun_out/r4b/shapes.log; exit 1; }
cat gpurun_out/r4b/shapes.log
timeout -k 10 600 python bench.py --steps 30 --warmup 10 > gpurun_out/r4b/bench.log 2>&1 || { tail -30 gpurun_out/r4b/bench.log; exit 1; }
tail -1 gpurun_out/r4b/bench.log | python -c "import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','value_bf16','value_config2','value_config5') if k in d}); print(d.get('roofline'))"
