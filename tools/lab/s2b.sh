#!/bin/bash
# session-2 second call: ConvNeXt GPU tests; config-2 kernel trace to count the per-step copyBuffer launches in steady state
mkdir -p gpurun_out/s2b
timeout -k 10 400 python -m pytest tests/test_convnext.py -m gpu -x -q -s > gpurun_out/s2b/convnext.log 2>&1
echo "convnext rc=$? $(tail -1 gpurun_out/s2b/convnext.log)"; grep -n "convnext_tiny pooled\|^E  \|Error" gpurun_out/s2b/convnext.log | head -20 | cut -c1-300
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s2b/trace2 -- python3 bench.py --config 2 --steps 8 --warmup 2 > gpurun_out/s2b/trace2.log 2>&1
echo "trace rc=$?"
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/s2b/trace2/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
ce = [i for i, (s, n) in enumerate(rows) if 'ce_kernel' in n]
print("steps seen:", len(ce), "kernels total:", len(rows))
for a, b in list(zip(ce[:-1], ce[1:]))[-3:]:
    c = collections.Counter(n.split('(')[0][-40:] for s, n in rows[a:b])
    print("launches in step:", b - a, "copyBuffer:", sum(v for k, v in c.items() if 'copyBuffer' in k), "fill:", sum(v for k, v in c.items() if 'fill' in k.lower()))
first = ce[0]
c0 = collections.Counter('copyBuffer' in n for s, n in rows[:first])
print("before the first step's loss kernel: copyBuffer launches =", c0[True], "of", first)
PY
rm -rf gpurun_out/s2b/trace2
