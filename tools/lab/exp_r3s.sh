#!/bin/bash
# round 3: rehearse the driver's N > 1 bench launch on the one-GPU box: two ranks sharing the device over gloo (the RCCL path itself needs two GPUs)
OUT=gpurun_out/r3s2
mkdir -p $OUT
export MEMEHIP_DIST_BACKEND=gloo MEMEHIP_BENCH_SHARE_DEVICE=1
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 > $OUT/bench_n2.log 2>&1
echo "rc=$?"
tail -3 $OUT/bench_n2.log | cut -c1-600
