"""Idle gaps of one step from a rocprofv3 kernel trace (csv with Start_Timestamp / End_Timestamp).
python tools/timeline.py <kernel_trace.csv> [marker]   - prints the kernels of the last step in launch order with
the gap in front of each, and the busy / idle split (union over streams).  A step starts at every launch of the
marker kernel (default ms_summary_k; vx_minmax_k for the voxel stage)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
name = lambda r: r["Kernel_Name"].split("(")[0].replace("pch::", "").replace("void ", "")[:28]
# a step starts at every ms_summary_k launch
marker = sys.argv[2] if len(sys.argv) > 2 else "ms_summary_k"
starts = [i for i, r in enumerate(rows) if marker in r["Kernel_Name"]]
if len(starts) < 3:
    sys.exit("fewer than 3 steps in the trace")
a, b = starts[-2], starts[-1]
step = rows[a:b]
t0 = int(step[0]["Start_Timestamp"])
end = t0
busy = 0
print(f"{'kernel':30s} {'start_us':>9s} {'dur_us':>8s} {'gap_us':>8s}")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = max(0, s - end)
    print(f"{name(r):30s} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:8.1f}")
    if e > end:
        busy += e - max(s, end)
        end = e
total = int(rows[b]["Start_Timestamp"]) - t0
print(f"step {total / 1e3:.1f} us: busy {busy / 1e3:.1f} us, idle {(total - busy) / 1e3:.1f} us, kernels {len(step)}")
