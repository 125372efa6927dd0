#!/bin/bash
# round 4, o: weight prefetch one layer ahead on a third stream (MEMEHIP_PREFETCH = workgroups per prefetch launch)
mkdir -p gpurun_out/r4o
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
show() { tail -1 $1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); r=d["roofline"]; print(d["ms_per_step"], d["value"], d["config"]["final_loss"], {k:v["ms_per_step"] for k,v in r["all_gemm_kernels"].items()})'; }
for rep in 1 2 3; do
  for pf in 0 32 128 512; do
    MEMEHIP_PREFETCH=$pf timeout -k 10 200 python bench.py $B > gpurun_out/r4o/pf${pf}_$rep.log 2>&1; echo "prefetch $pf rep $rep: $(show gpurun_out/r4o/pf${pf}_$rep.log)"
  done
done
