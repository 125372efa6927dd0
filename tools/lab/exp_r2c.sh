#!/bin/bash
set -e
OUT=gpurun_out/r2c
mkdir -p $OUT
for m in none eager1 stream segments; do
  timeout -k 10 200 python tools/ddp_host_probe.py $m > $OUT/$m.log 2>&1 || { tail -20 $OUT/$m.log; exit 1; }
  tail -1 $OUT/$m.log
done
