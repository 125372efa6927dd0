#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r4x
export MEMEHIP_LIB=$PWD/multimodal_propaganda_meme_classification_amd/libmemehip_lab.so MEMEHIP_LIB_F16=$PWD/multimodal_propaganda_meme_classification_amd/libmemehip_lab_f16.so MEMEHIP_GEMM_BIG_MIN=0
timeout -k 10 500 python tools/gemm_shapes.py --config 5 --cands product,v1,v5,v6,v7,v8,v9 --dtype fp16 --csv gpurun_out/r4x/c5_lab.csv > gpurun_out/r4x/c5_lab.log 2>&1 || { tail -20 gpurun_out/r4x/c5_lab.log; exit 1; }
python - <<'PY'
import csv, collections
rows = [r for r in csv.reader(open("gpurun_out/r4x/c5_lab.csv")) if r and not r[0].startswith("#")][1:]
t = collections.OrderedDict()
for shape, cand, med, mn, tf in rows:
    t.setdefault(shape, {})[cand] = float(med)
cands = list(next(iter(t.values())).keys())
print("shape".ljust(28), " ".join(c.rjust(8) for c in cands))
for s_, d in t.items():
    print(s_.ljust(28), " ".join(f"{d.get(c, 0):8.1f}" for c in cands))
PY
