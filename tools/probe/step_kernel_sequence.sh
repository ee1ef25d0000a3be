#!/bin/bash
# The kernels of ONE replayed step of a workload, in start order (rocprofv3 --kernel-trace; the step between the last two
# optimiser launches): shows which nodes of the captured graph are not the path's own kernels and where they sit.
#   bash tools/probe/step_kernel_sequence.sh cdsprites_shipped > gpurun_out/seq_cdsprites_shipped.txt
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
c=${1:-cdsprites_shipped}
rm -rf gpurun_out/seq_$c
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/seq_$c -- python3 bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/seq_$c.err
f=$(find gpurun_out/seq_$c -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
ends = [i for i, r in enumerate(rows) if "adam" in r["Kernel_Name"]]
a, b = ends[-2] + 1, ends[-1] + 1
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f}  q{r.get('Queue_Id', '?')}  {r['Kernel_Name'][:110]}")
print("kernels in the step:", b - a)
P
rm -rf gpurun_out/seq_$c
