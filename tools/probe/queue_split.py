"""Per-queue composition of a captured step from a rocprofv3 kernel trace: python queue_split.py <kernel_trace.csv> [steps]
Takes the last `steps` repetitions (default 5) of the periodic kernel sequence and prints, per hardware queue, the busy time
per step and the kernels' (count, mean us, total us per step)."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the optimiser launch ends a step
marks = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
marks = marks[-(nsteps + 1):]
sel = rows[marks[0] + 1: marks[-1] + 1]
t0, t1 = int(sel[0]["Start_Timestamp"]), int(sel[-1]["End_Timestamp"])
print(f"{nsteps} steps, {1e-3 * (t1 - t0) / nsteps:.1f} us per step (trace clock), {len(sel) / nsteps:.0f} kernels per step")
byq = defaultdict(lambda: defaultdict(list))
for r in sel:
    byq[r.get("Queue_Id", "?")][r["Kernel_Name"][:70]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for q, ks in byq.items():
    tot = sum(sum(v) for v in ks.values())
    print(f"queue {q}: busy {1e-3 * tot / nsteps:.1f} us per step")
    for k, v in sorted(ks.items(), key=lambda kv: -sum(kv[1]))[:18]:
        print(f"   {len(v) / nsteps:6.1f} x {1e-3 * sum(v) / len(v):7.1f} us = {1e-3 * sum(v) / nsteps:7.1f}  {k}")
