#!/bin/bash
# Round 3: the data-parallel schedules on a 1-rank RCCL group against the single-GPU graph, one box, two repetitions
for rep in 1 2; do
  for m in single segments stream graph; do
    if [ $m = single ]; then extra=""; else extra="--force-ddp --ddp-mode $m"; fi
    python bench.py $extra --no-extras --no-cpu-baseline --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['ms_per_step'], d['value'])"
  done
done
