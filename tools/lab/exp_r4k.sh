#!/bin/bash
mkdir -p gpurun_out/r4k
timeout -k 10 900 python tools/publish_parity.py gpurun_out/r4k/parity.txt > gpurun_out/r4k/publish.log 2>&1
echo "publish_parity rc=$?"
grep -n "^E \|FAILED" gpurun_out/r4k/publish.log | head -20
