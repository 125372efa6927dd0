#!/bin/bash
# Rehearsal of the N > 1 bench path at CONFIG-3 SIZE on a 1-GPU box: two ranks on ONE device over gloo
# (MEMEHIP_BENCH_SHARE_DEVICE=1, MEMEHIP_DIST_BACKEND=gloo).  Exercises, at the real bucket sizes (57 MB layer-pair slices): the parameter
# broadcast, the per-segment gradient exchange in both wire formats (fp32 all-reduce; bf16 all-to-all + fp32 shard sum + all-gather), the
# gathered embedding-table gradient, 1/world in Adam, the Adam slices behind each bucket.  NOT exercisable here: --ddp-mode graph (gloo
# collectives run on host threads and cannot be captured into a hipGraph; two RCCL ranks cannot share one device).
OUT=${1:-gpurun_out/ddp2}
mkdir -p $OUT
export MEMEHIP_BENCH_SHARE_DEVICE=1 MEMEHIP_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0
rc=0
for wire in none bf16; do
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29600 + RANDOM % 200)) \
      bench.py --gpus 2 --steps 4 --warmup 2 --no-extras --no-cpu-baseline --ddp-compress $wire > $OUT/bench_w2_$wire.log 2>&1
  r=$?
  [ $r -ne 0 ] && rc=$r
  echo "world 2, wire $wire: rc=$r $(tail -1 $OUT/bench_w2_$wire.log | python -c 'import json,sys
try:
    d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "value", d["value"], "wire", d["ddp_wire"], "final loss", d["config"]["final_loss"])
except Exception as e:
    print("no JSON line")')"
done
exit $rc
