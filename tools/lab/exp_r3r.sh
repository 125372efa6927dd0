#!/bin/bash
# round 3: HEAD against the round's starting point on ONE box, alternating.  Needs the baseline next to the tree first (it is git-ignored):
#   git worktree add -f .ab_baseline e1e1e6f && make -C .ab_baseline/multimodal_propaganda_meme_classification_amd/csrc -j6   (remove with: git worktree remove --force .ab_baseline)
OUT=gpurun_out/r3r
mkdir -p $OUT
B="--no-extras --no-cpu-baseline --steps 60 --warmup 10"
for rep in 1 2 3; do
  (cd .ab_baseline && timeout -k 10 200 python bench.py $B > ../$OUT/base_$rep.log 2>&1)
  echo "baseline rep $rep: $(tail -1 $OUT/base_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"])')"
  timeout -k 10 200 python bench.py $B > $OUT/head_$rep.log 2>&1
  echo "HEAD     rep $rep: $(tail -1 $OUT/head_$rep.log | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["value"], d["roofline"]["avg_launch_us"])')"
done
