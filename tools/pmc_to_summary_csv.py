"""rocprofv3 --pmc counter_collection.csv -> per-kernel summary CSV (launches, mean and sum of the counter).
usage: pmc_to_summary_csv.py <pmc_dir> <COUNTER> <out.csv>"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if r['Counter_Name'] == sys.argv[2]:
        acc[r['Kernel_Name']].append(float(r['Counter_Value']))
with open(sys.argv[3], 'w', newline='') as o:
    w = csv.writer(o)
    w.writerow(["Kernel_Name", "Counter_Name", "Launches", "MeanValue_KiB", "SumValue_KiB"])
    for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
        w.writerow([k, sys.argv[2], len(v), round(sum(v) / len(v), 1), round(sum(v), 1)])
