#!/bin/bash
# kernel durations (without launch gaps) of tools/probe/rc_time.py, grouped by kernel and grid
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/rc_trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/rc_trace -- python3 tools/probe/rc_time.py ${1:-24} > gpurun_out/rc_trace.log 2>&1
f=$(find gpurun_out/rc_trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if not name.startswith("rc_"):
        continue
    key = (name, r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg.setdefault(key, []).append(d)
for k, v in agg.items():
    v = sorted(v)
    print(f"{k[0]:34s} grid {int(k[1])//256:5d} x {k[2]:>4s} x {k[3]:>3s}  n {len(v):4d}  median {v[len(v)//2]:7.1f} us  min {v[0]:7.1f}")
PY
rm -rf gpurun_out/rc_trace
