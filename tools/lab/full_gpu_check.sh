#!/bin/bash
mkdir -p gpurun_out/r4full
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/r4full/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/r4full/tests.log)"; grep -n "^E  \|Error\|FAILED" gpurun_out/r4full/tests.log | head -10 | cut -c1-300
timeout -k 10 100 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
