#!/bin/bash
# round-4 evidence in one call: rocprofv3 passes of the default bench command (collect_profiles.sh), the parity measurements, the config-2 kernel stats
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4s
bash tools/collect_profiles.sh r4s > gpurun_out/r4s/collect.log 2>&1; rc=$?
tail -3 gpurun_out/r4s/collect.log | cut -c1-400
[ $rc -ne 0 ] && exit $rc
timeout -k 10 900 python tools/publish_parity.py gpurun_out/r4s/parity.txt > gpurun_out/r4s/parity.log 2>&1; rc=$?
grep "pytest:" gpurun_out/r4s/parity.txt
[ $rc -ne 0 ] && exit $rc
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4s/c2stats/run -- python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r4s/c2stats.log 2>&1 || { tail -5 gpurun_out/r4s/c2stats.log; exit 1; }
tail -1 gpurun_out/r4s/c2stats.log | cut -c1-300
