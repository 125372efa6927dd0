#!/bin/bash
# round 4, u: which layers take the fixed-point accumulators (by row-tile count)
mkdir -p gpurun_out/r4u
show() { tail -1 $1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["config"].get("final_loss"))'; }
for rep in 1 2; do
  for t in 0 16 64 256 100000; do
    MEMEHIP_BN_ATOMIC_MAXTILES=$t timeout -k 10 200 python bench.py --config 2 --steps 100 --warmup 10 --no-extras --no-cpu-baseline > gpurun_out/r4u/c2_t${t}_$rep.log 2>&1; echo "max tiles $t rep $rep: $(show gpurun_out/r4u/c2_t${t}_$rep.log)"
  done
done
