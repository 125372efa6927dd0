#!/bin/bash
mkdir -p gpurun_out/r4n
timeout -k 10 300 python tools/gemm_cold_probe.py > gpurun_out/r4n/cold.log 2>&1; echo rc=$?; grep -v amdgpu.ids gpurun_out/r4n/cold.log
