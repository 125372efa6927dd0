#!/bin/bash
set -e
mkdir -p gpurun_out/r2j
timeout -k 10 300 python tools/gemm_epi_probe.py > gpurun_out/r2j/probe.log 2>&1 || { tail -20 gpurun_out/r2j/probe.log; exit 1; }
cat gpurun_out/r2j/probe.log
