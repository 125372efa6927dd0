#!/bin/bash
# same-box, step-level A/B of environment switches (config 3, two repetitions, interleaved)
set -e
OUT=gpurun_out/r2u
mkdir -p $OUT
B="--no-extras --no-cpu-baseline"
run() { name=$1; shift; env "$@" timeout -k 10 200 python bench.py $B > $OUT/$name.log 2>&1; echo "$name: $(tail -1 $OUT/$name.log | cut -c150-200)"; }
for rep in 1 2; do
  run default_$rep X=1
  run gemm7_$rep MEMEHIP_GEMM_VARIANT=7
  run attn_res256_$rep MEMEHIP_ATTN_BWD_RESIDENT_MAX=256
  run attn_merged0_$rep MEMEHIP_ATTN_BWD_MERGED=0
done
