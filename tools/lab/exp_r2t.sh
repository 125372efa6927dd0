#!/bin/bash
# same-box, step-level A/B of an alternative bf16 build of the library (MEMEHIP_LIB) against the in-tree one:
#   tools/exp_r2t.sh build_ab/libmemehip_<variant>.so <label>
set -e
ALT=${1:-build_ab/libmemehip_exact.so}
LABEL=${2:-alt}
OUT=gpurun_out/r2t_$LABEL
mkdir -p $OUT
B="--no-extras --no-cpu-baseline --dtype bf16"
for rep in 1 2; do
  MEMEHIP_LIB=$ALT timeout -k 10 300 python bench.py --config 5 --steps 5 --warmup 2 $B > $OUT/c5_alt_$rep.log 2>&1
  echo "config5 $LABEL   ($rep): $(tail -1 $OUT/c5_alt_$rep.log | cut -c100-230)"
  timeout -k 10 300 python bench.py --config 5 --steps 5 --warmup 2 $B > $OUT/c5_tree_$rep.log 2>&1
  echo "config5 in-tree ($rep): $(tail -1 $OUT/c5_tree_$rep.log | cut -c100-230)"
done
for rep in 1 2; do
  MEMEHIP_LIB=$ALT timeout -k 10 200 python bench.py $B > $OUT/c3_alt_$rep.log 2>&1
  echo "config3 $LABEL   ($rep): $(tail -1 $OUT/c3_alt_$rep.log | cut -c100-200)"
  timeout -k 10 200 python bench.py $B > $OUT/c3_tree_$rep.log 2>&1
  echo "config3 in-tree ($rep): $(tail -1 $OUT/c3_tree_$rep.log | cut -c100-200)"
done
