"""Analyse a rocprofv3 kernel trace of bench.py: per-step makespan, busy union, sum of durations, idle gaps."""
import csv, glob, sys, collections, re
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# steps are delimited by the loss kernel (one per step)
ends = [e for s, e, n in rows if 'ce_kernel' in n]
steps = []
for a, b in zip(ends[:-1], ends[1:]):
    steps.append([(s, e, n) for s, e, n in rows if s >= a and e <= b])
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    return n.split('(')[0].replace('void ', '').strip()[:40]
for st in steps[-4:-1]:
    t0, t1 = min(s for s, e, n in st), max(e for s, e, n in st)
    span = (t1 - t0) / 1e6
    total = sum(e - s for s, e, n in st) / 1e6
    # union of busy intervals
    cur_s, cur_e, busy = None, None, 0
    for s, e, n in sorted(st):
        if cur_e is None or s > cur_e:
            if cur_e is not None: busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"step: {len(st)} kernels, makespan {span:.3f} ms, busy-union {busy/1e6:.3f} ms, sum of durations {total:.3f} ms, idle {span-busy/1e6:.3f} ms")
st = steps[-2]
agg = collections.defaultdict(float)
for s, e, n in st: agg[short(n)] += (e - s) / 1e6
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:16]:
    print(f"   {k:42s} {v:7.3f} ms")
