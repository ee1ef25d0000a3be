#!/bin/bash
# per-kernel durations of the cfg2 step with both towers on ONE stream (every kernel alone on the chip) beside the shipped
# two-stream step: rocprofv3 kernel stats for batch $1 -> gpurun_out/r04_{alone,instep}_b$1_kernel_stats.csv
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
B=${1:-128}
for mode in alone instep; do
  rm -rf gpurun_out/prof_$mode
  if [ $mode = alone ]; then export MMVAE_STREAMS=0; else unset MMVAE_STREAMS; fi
  python3 bench.py --batch $B --steps 100 --warmup 10 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | cut -c1-200
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$mode -- python3 bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/prof_$mode.err; echo "$mode prof rc=$?"
  f=$(find gpurun_out/prof_$mode -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r04_${mode}_b${B}_kernel_stats.csv
  rm -rf gpurun_out/prof_$mode
done
