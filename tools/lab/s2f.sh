#!/bin/bash
# HEAD kernel stats (graphed run) + idle-gap analysis; A/B of the text heads' backward as its own launch (MEMEHIP_ATTN_ONEPASS=1) now that their staging is one round trip
mkdir -p gpurun_out/s2f
export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s2f/stats/run -- python3 bench.py --steps 16 --warmup 2 --no-extras --no-cpu-baseline > gpurun_out/s2f/stats.log 2>&1
echo "stats rc=$?"
python3 tools/trace_gaps.py gpurun_out/s2f/stats/run > gpurun_out/s2f/gaps.txt 2>&1; cat gpurun_out/s2f/gaps.txt | head -30
f=$(ls gpurun_out/s2f/stats/run/*/*kernel_stats.csv | head -1); cp $f gpurun_out/s2f/kernel_stats.csv; head -14 $f | cut -c1-160
rm -rf gpurun_out/s2f/stats
bash tools/lab/ab.sh s2f_ab 2 --no-extras --no-cpu-baseline --steps 60 --warmup 10 -- head= -- onepass1=MEMEHIP_ATTN_ONEPASS=1
