"""Idle time of the GPU inside one step of a rocprofv3 kernel trace: span (loss kernel to loss kernel), union of the
kernels' busy intervals, and the idle gaps by size.  usage: trace_idle.py <dir> [step index]"""
import sys, glob, csv, collections
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
ce = [i for i, r in enumerate(rows) if 'ce_kernel' in r[2]]
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
st = rows[ce[k]:ce[k + 1]]
t0, t1 = st[0][0], max(e for _, e, _ in st)
cur_e, busy, gaps = st[0][1], 0, []
cur_s = st[0][0]
for s, e, n in st[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, n))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("span %.3f ms  busy-union %.3f ms  idle %.3f ms in %d gaps" % ((t1 - t0) / 1e6, busy / 1e6, sum(g for g, _ in gaps) / 1e6, len(gaps)))
hist = collections.Counter()
for g, _ in gaps:
    hist["<1us" if g < 1000 else "<2us" if g < 2000 else "<5us" if g < 5000 else "<20us" if g < 20000 else ">=20us"] += 1
print(dict(hist))
for g, n in sorted(gaps, reverse=True)[:6]:
    print("  %.1f us before %s" % (g / 1e3, n[:60]))
