import csv, collections, sys, glob
d = sys.argv[1]
cc = glob.glob(d + '/*/*counter_collection.csv')[0]
kt = glob.glob(d + '/*/*kernel_trace.csv')
rows = list(csv.DictReader(open(cc)))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r['Kernel_Name']
    if len(sys.argv) > 2 and sys.argv[2] not in k: continue
    agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
dur = collections.defaultdict(list)
if kt:
    for r in csv.DictReader(open(kt[0])):
        dur[r['Kernel_Name']].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, c in agg.items():
    ds = sorted(dur.get(k, [0]))
    print(k[:70], 'median us %.1f' % ds[len(ds) // 2])
    for n, v in sorted(c.items()):
        v = sorted(v)
        print('   %-28s %.4e' % (n, v[len(v) // 2]))
