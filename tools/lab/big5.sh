#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4x
timeout -k 10 300 python tools/gemm_big_timeline.py > gpurun_out/r4x/big_timeline_nt.log 2>&1 || { tail -20 gpurun_out/r4x/big_timeline_nt.log; exit 1; }
cat gpurun_out/r4x/big_timeline_nt.log
