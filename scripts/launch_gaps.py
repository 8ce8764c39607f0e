"""Idle time between consecutive kernels of one timed pass, from a rocprofv3 kernel trace (scripts/attic usage: see below).
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/rp_trace -o t -- python3 $REPO/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-profile
    python3 scripts/launch_gaps.py /tmp/rp_trace
Takes the LAST 50-step pass of the trace (the last 50 launches of dmx_sched_ddim_kernel delimit it), lists: wall time of the pass, sum of kernel durations, idle
time, the gap distribution, and the gaps in front of each step's first kernel (graph-to-graph transitions)."""
import csv
import glob
import os
import sys

f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
sched = [i for i, r in enumerate(rows) if "dmx_sched_ddim_kernel" in r[2]]
assert len(sched) >= 51, f"only {len(sched)} scheduler steps in the trace"
lo, hi = sched[-51] + 1, sched[-1]                 # kernels of the last pass: behind the previous pass's last scheduler step .. this pass's last one
ks = rows[lo:hi + 1]
wall = (ks[-1][1] - ks[0][0]) * 1e-6
busy = sum(e - s for s, e, _ in ks) * 1e-6
gaps = [(ks[i + 1][0] - ks[i][1]) * 1e-3 for i in range(len(ks) - 1)]       # us (negative = overlap)
pos = [g for g in gaps if g > 0]
print(f"{len(ks)} kernels; pass wall {wall:.2f} ms, sum of kernel durations {busy:.2f} ms, idle between kernels {sum(pos) * 1e-3:.2f} ms ({100 * sum(pos) * 1e-3 / wall:.1f} %)")
srt = sorted(pos)
print(f"gaps (us): median {srt[len(srt) // 2]:.2f}  mean {sum(pos) / len(pos):.2f}  p90 {srt[int(0.9 * len(srt))]:.2f}  p99 {srt[int(0.99 * len(srt))]:.2f}  max {srt[-1]:.1f}")
step_first = [i for i, r in enumerate(ks) if "dmx_temb_row_kernel" in r[2] or "dmx_im2col_small_kernel" in r[2]]
big = sorted(((g, ks[i][2][:50], ks[i + 1][2][:50]) for i, g in enumerate(gaps)), reverse=True)[:8]
print("largest gaps (us, after kernel -> before kernel):")
for g, a, b in big:
    print(f"  {g:8.1f}  {a}  ->  {b}")
after_sched = [gaps[i] for i in range(len(ks) - 1) if "dmx_sched_ddim_kernel" in ks[i][2]]
before_sched = [gaps[i] for i in range(len(ks) - 1) if "dmx_sched_ddim_kernel" in ks[i + 1][2]]
if after_sched:
    print(f"step transitions: gap behind the scheduler kernel (graph launch of the next step) mean {sum(after_sched) / len(after_sched):.1f} us; in front of it mean {sum(before_sched) / len(before_sched):.1f} us; x {len(after_sched)} steps")
