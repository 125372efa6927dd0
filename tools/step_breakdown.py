"""Per-kernel time of ONE graphed step from a rocprofv3 kernel trace (csv dir or rocpd .db): steps are delimited by
the dense Adam kernel of the tail. usage: step_breakdown.py <dir_with_kernel_trace_csv | results.db>"""
import sys, glob, csv, sqlite3, collections, re
src = sys.argv[1]
if src.endswith('.db'):
    rows = sqlite3.connect(src).execute("select name,start,end from kernels order by start").fetchall()
else:
    f = glob.glob(src + '/**/*kernel_trace.csv', recursive=True)[0]
    rows = sorted((r['Kernel_Name'], int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in csv.DictReader(open(f)))
    rows.sort(key=lambda r: r[1])
def short(n):
    n = re.sub(r'\(anonymous namespace\)::', '', n)
    n = re.sub(r'_ZN12_GLOBAL__N_1\d+', '', n)
    return n.split('(')[0].replace('void ', '').strip()[:46]
ce = [i for i, r in enumerate(rows) if 'ce_kernel' in r[0]]
# a graphed step = ce .. next ce; take the third from the end of the timed region (before the eager replays)
k = int(sys.argv[2]) if len(sys.argv) > 2 else 4
a, b = ce[k], ce[k + 1]
st = rows[a:b]
agg = collections.defaultdict(float); cnt = collections.Counter()
for n, s, e in st:
    agg[short(n)] += (e - s) / 1e6; cnt[short(n)] += 1
t0, t1 = min(s for _, s, _ in st), max(e for _, _, e in st)
print("step span %.3f ms, sum of kernel durations %.3f ms, %d kernels" % ((t1 - t0) / 1e6, sum(agg.values()), len(st)))
for kk, v in sorted(agg.items(), key=lambda kv: -kv[1])[:26]:
    print(f"   {kk:48s} {cnt[kk]:4d} {v:7.3f} ms")
