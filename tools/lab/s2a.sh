#!/bin/bash
# session-2 first call: GPU suite sanity on the rebuilt tree, baseline rates of configs 3 and 2 on this box, copy sources of config 2
mkdir -p gpurun_out/s2a
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/s2a/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/s2a/tests.log)"
timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 60 --warmup 10 > gpurun_out/s2a/bench3.log 2>&1 && tail -1 gpurun_out/s2a/bench3.log | cut -c1-200
timeout -k 10 200 python bench.py --config 2 --steps 100 --warmup 10 > gpurun_out/s2a/bench2.log 2>&1 && tail -1 gpurun_out/s2a/bench2.log | cut -c1-200
timeout -k 10 200 python tools/find_copies.py > gpurun_out/s2a/copies.log 2>&1
tail -45 gpurun_out/s2a/copies.log
