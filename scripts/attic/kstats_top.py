"""Prints the top kernels of a rocprofv3 --stats kernel_stats.csv found under a directory.  python scripts/kstats_top.py DIR [N]"""
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f"{r['Name'][:84]:84s} {int(r['Calls']):5d} {float(r['AverageNs']) / 1e3:9.1f} us {float(r['TotalDurationNs']) / 1e6:8.1f} ms {float(r['TotalDurationNs']) / tot * 100:5.1f}%")
