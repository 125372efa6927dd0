"""rocprofv3 SQ counter pass -> per-kernel summary CSV: launches, mean duration, MFMA-busy share and where the wave-cycles go.
usage: sq_summary.py <pmc_dir> <out.csv>
Columns: mfma_busy_frac = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines): matrix-pipe busy cycles per
SIMD over the cycles the kernel was resident (rocprofv3 sums MFMA-busy over the chip's 1024 SIMDs and SQ_BUSY_CYCLES over its 32
shader engines; check: clock_ghz = SQ_BUSY_CYCLES / 32 / duration comes out at the 1.7-2.0 GHz the chip holds under a profiled
MFMA load); wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES (wave parked in s_waitcnt / barrier); issue_stall_frac =
SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES; active_frac = SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES; lds_conflict_frac = SQ_LDS_BANK_CONFLICT /
SQ_LDS_IDX_ACTIVE."""
import collections
import csv
import glob
import sys

d, out = sys.argv[1], sys.argv[2]
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(lambda: collections.defaultdict(int))
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[k][r["Counter_Name"]] += 1
dur = collections.defaultdict(list)
for f in glob.glob(d + "/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = []
for k, c in agg.items():
    n = max(cnt[k].values())
    busy, wave = c.get("SQ_BUSY_CYCLES", 0.0), c.get("SQ_WAVE_CYCLES", 0.0)
    short = k.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].strip()
    ds = dur.get(k, [])
    rows.append([short, n, round(sum(ds) / len(ds), 2) if ds else "", round(sum(ds) / 1e3, 3) if ds else "",
                 round((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024.0) / (busy / 32.0), 4) if busy else "",
                 round(busy / 32.0 / n / (sum(ds) / len(ds) * 1e3), 3) if (busy and ds) else "",
                 round(c.get("SQ_WAIT_ANY", 0.0) / wave, 4) if wave else "", round(c.get("SQ_WAIT_INST_ANY", 0.0) / wave, 4) if wave else "",
                 round(c.get("SQ_ACTIVE_INST_ANY", 0.0) / wave, 4) if wave else "",
                 round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4) if c.get("SQ_LDS_IDX_ACTIVE") else ""])
rows.sort(key=lambda r: -(r[3] or 0))
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "mean_us", "total_ms", "mfma_busy_frac", "clock_ghz", "wait_frac", "issue_stall_frac", "active_frac", "lds_conflict_frac"])
    w.writerows(rows)
for r in rows[:14]:
    print(r)
