#!/bin/bash
# round 4, r: config 2 kernel statistics (what the ResNet-50 step spends where, launches per step)
OUT=gpurun_out/r4r
mkdir -p $OUT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/c2/run -- python bench.py --config 2 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $OUT/c2.log 2>&1
echo "rc=$?"; tail -1 $OUT/c2.log | cut -c1-200
f=$(ls -t $OUT/c2/run/*/*_kernel_stats.csv | head -1); cp $f $OUT/c2_kernel_stats.csv
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/r4r/c2_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows); calls=sum(int(r['Calls']) for r in rows)
print('total ms', tot/1e6, 'calls', calls)
for r in rows[:26]:
    print(f"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
PY
