#!/bin/bash
# final check at HEAD: full GPU suite, smoke, default bench line
mkdir -p gpurun_out/s2n
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/s2n/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/s2n/tests.log)"; grep -n "^E  \|FAILED" gpurun_out/s2n/tests.log | head -10 | cut -c1-300
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout -k 10 900 python bench.py > gpurun_out/s2n/bench_default.log 2>&1; tail -1 gpurun_out/s2n/bench_default.log | cut -c1-250
