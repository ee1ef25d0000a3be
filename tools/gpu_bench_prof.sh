#!/bin/bash
# bench + rocprof kernel trace on the GPU box; results under gpurun_out/
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python3 bench.py --steps 200 --warmup 20 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "bench rc=$?"; cat gpurun_out/bench.json; tail -5 gpurun_out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/prof_bench.json 2> gpurun_out/prof.err; echo "prof rc=$?"
find gpurun_out/prof -name "*kernel_stats.csv" | head -2
f=$(find gpurun_out/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -45 "$f" | cut -c1-220
