#!/bin/bash
set -e
OUT=gpurun_out/r2l
mkdir -p $OUT
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1 || { tail -20 $OUT/smoke.log; exit 1; }
echo "smoke: $(tail -1 $OUT/smoke.log | cut -c1-200)"
MEMEHIP_DIST_BACKEND=gloo MEMEHIP_BENCH_SHARE_DEVICE=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 5 --warmup 2 > $OUT/n2.log 2>&1 || { tail -30 $OUT/n2.log; exit 1; }
echo "n2 gloo: $(tail -1 $OUT/n2.log | cut -c1-260)"
MEMEHIP_DIST_BACKEND=gloo MEMEHIP_BENCH_SHARE_DEVICE=1 timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 5 --warmup 2 --ddp-mode stream --ddp-compress bf16 > $OUT/n2s.log 2>&1 || { tail -30 $OUT/n2s.log; exit 1; }
echo "n2 gloo stream bf16: $(tail -1 $OUT/n2s.log | cut -c1-260)"
