"""rocprofv3 --kernel-trace CSV -> one line per (kernel, grid) in first-appearance order: launches, mean / min us.
    python scripts/trace_runs.py <dir or kernel_trace.csv> [min_count] [name filter]"""
import csv, glob, os, sys
path = sys.argv[1]
if os.path.isdir(path):
    path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 3
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
groups = {}
for r in rows:
    name = r["Kernel_Name"]
    if flt and flt not in name: continue
    key = (name, r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "")))
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    groups.setdefault(key, []).append(d)
for (name, g, wg), ds in groups.items():
    if len(ds) >= minc:
        print(f"{len(ds):4d} x mean {sum(ds) / len(ds):8.1f} us  min {min(ds):8.1f}  grid {g}/{wg}  {name[:100]}")
if len(sys.argv) > 4:      # chunk N: the filtered kernel's dispatches in order, N at a time (one probe shape each)
    n = int(sys.argv[4]); seq = []
    for r in rows:
        if flt in r["Kernel_Name"]: seq.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for i in range(0, len(seq), n):
        c = seq[i:i + n]
        print(f"chunk {i // n:2d}: {len(c)} x mean {sum(c) / len(c):8.1f} us  min {min(c):8.1f}  max {max(c):8.1f}")
