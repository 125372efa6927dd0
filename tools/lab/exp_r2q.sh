#!/bin/bash
set -e
OUT=gpurun_out/r2q
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-extras --no-cpu-baseline --config 5 --steps 1 --warmup 1 --no-graph --no-overlap-wgrad --no-overlap-opt"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $OUT/sq/run -- python bench.py $B > $OUT/sq.log 2>&1
python tools/sq_summary.py $OUT/sq/run $OUT/sq_summary.csv > /dev/null
head -12 $OUT/sq_summary.csv | cut -c1-160
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $OUT/inst/run -- python bench.py $B > $OUT/inst.log 2>&1 || true
python - <<'PY'
import csv,glob,collections
f=glob.glob("gpurun_out/r2q/inst/run/*/*counter_collection.csv")
if f:
    acc=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        if 'attn' in k:
            print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
