#!/bin/bash
# One parameterised A/B script (replaces the exp_*.sh family; their commands are recorded in tools/lab/RESULTS.md).
#   tools/lab/ab.sh <tag> <reps> [bench.py args ...] -- NAME=ENV[,ENV...] [-- NAME=ENV...] ...
# Alternates `python bench.py <args>` runs under the given environments on ONE box, <reps> times, and prints ms/step, value and the
# sequential-replay GEMM times of each.  NAME `baseline` runs the same command inside the same-box baseline worktree instead
#   (git worktree add -f .ab_baseline <commit> && make -C .ab_baseline/multimodal_propaganda_meme_classification_amd/csrc -j8).
# Example:  tools/lab/ab.sh r4x 3 --no-extras --no-cpu-baseline --steps 60 --warmup 10 -- baseline= -- head= -- nograph=MEMEHIP_X=1
TAG=$1; REPS=$2; shift 2
ARGS=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do ARGS+=("$1"); shift; done
VARIANTS=(); while [ $# -gt 0 ]; do [ "$1" = "--" ] || VARIANTS+=("$1"); shift; done
OUT=gpurun_out/$TAG; mkdir -p $OUT
show() { tail -1 $1 | python -c 'import json,sys
try:
    d=json.loads(sys.stdin.read()); r=d.get("roofline") or {}
    print(d["ms_per_step"], d["value"], {k: v["ms_per_step"] for k, v in (r.get("all_gemm_kernels") or {}).items()})
except Exception as e:
    print("no JSON line:", e)'; }
for rep in $(seq 1 $REPS); do
  for v in "${VARIANTS[@]}"; do
    name=${v%%=*}; envs=${v#*=}
    log=$OUT/${name}_$rep.log
    if [ "$name" = "baseline" ]; then
      (cd .ab_baseline && env ${envs//,/ } timeout -k 10 300 python bench.py "${ARGS[@]}" > ../$log 2>&1)
    else
      env ${envs//,/ } timeout -k 10 300 python bench.py "${ARGS[@]}" > $log 2>&1
    fi
    echo "$name rep $rep: $(show $log)"
  done
done
