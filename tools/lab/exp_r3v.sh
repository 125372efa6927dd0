#!/bin/bash
# round 3: end-of-round records -- the 1-rank RCCL data-parallel tax (segments / graph) and config 5 kernel stats
OUT=gpurun_out/r3v
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline --steps 40 --warmup 8"
timeout -k 10 200 python bench.py $B > $OUT/single.log 2>&1; echo "single graph: $(tail -1 $OUT/single.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
timeout -k 10 200 python bench.py $B --force-ddp > $OUT/ddp_segments.log 2>&1; echo "ddp segments: $(tail -1 $OUT/ddp_segments.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
timeout -k 10 200 python bench.py $B --force-ddp --ddp-mode graph > $OUT/ddp_graph.log 2>&1; echo "ddp graph: $(tail -1 $OUT/ddp_graph.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
timeout -k 10 200 python bench.py $B > $OUT/single2.log 2>&1; echo "single graph again: $(tail -1 $OUT/single2.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"])')"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats5/run -- python bench.py --config 5 --steps 4 --warmup 2 --no-graph --no-overlap-wgrad --no-overlap-opt --no-extras --no-cpu-baseline > $OUT/stats5.log 2>&1
echo "config 5 stats done"
