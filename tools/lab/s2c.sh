#!/bin/bash
# session-2 third call: LDS-DMA staging of the resident attention operands -- kernel A/B against the round's baseline build in one
# process, the attention / model GPU tests, then the step A/B (baseline worktree vs HEAD, 3 alternations)
mkdir -p gpurun_out/s2c
B=.ab_baseline/multimodal_propaganda_meme_classification_amd
H=multimodal_propaganda_meme_classification_amd
timeout -k 10 200 python tools/attn_ab.py $B/libmemehip_f16.so $H/libmemehip_f16.so > gpurun_out/s2c/attn_ab.log 2>&1; cat gpurun_out/s2c/attn_ab.log | tail -8
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "attn or attention or model or round2 or packed or properties or config5 or dropout" > gpurun_out/s2c/tests.log 2>&1
echo "tests rc=$? $(tail -1 gpurun_out/s2c/tests.log)"; grep -n "^E  \|FAILED" gpurun_out/s2c/tests.log | head -10 | cut -c1-300
bash tools/lab/ab.sh s2c_ab 3 --no-extras --no-cpu-baseline --steps 60 --warmup 10 -- baseline= -- head=
