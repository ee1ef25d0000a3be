"""Timeline of ONE training step from a rocprofv3 kernel trace (csv): every launch between two Adam kernels with its
start offset, duration, queue and the idle gap on its queue before it.  Usage: step_timeline.py kernel_trace.csv"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
adam = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
i0, i1 = adam[-3] + 1, adam[-2] + 1          # a step well inside the timed region
step = rows[i0:i1]
t0 = int(rows[adam[-3]]["End_Timestamp"])
last_end = {}
print(f"{'start':>8s} {'dur':>7s} {'gap':>6s} q   kernel  (us; gap = idle time on the same queue before the launch)")
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    q = r.get("Queue_Id", "?")
    gap = (s - last_end[q]) / 1e3 if q in last_end else float("nan")
    last_end[q] = e
    wg = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} {gap:6.1f} {q[-2:]:>2s}  {r['Kernel_Name'][:70]} [{wg}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']}]")
print("step wall:", (int(step[-1]["End_Timestamp"]) - t0) / 1e3, "us;  sum of kernel durations:", sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in step) / 1e3)
