#!/bin/bash
# End-of-round evidence on the shipped kernels: one bench line per workload + rocprofv3 kernel stats of the same command.
# -> gpurun_out/r03_bench_<cfg>.json, gpurun_out/r03_<cfg>_kernel_stats.csv   (copy into profiles/ afterwards)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
CFGS=${CFGS:-"cfg1 cfg3 cfg3_elbo cfg4 cfg5 mnistsvhn cdsprites_shipped"}
for c in $CFGS; do
  python3 bench.py --config $c --steps 50 --warmup 10 > gpurun_out/r03_bench_$c.json 2> gpurun_out/r03_bench_$c.err; echo "$c bench rc=$?"
  cut -c1-260 gpurun_out/r03_bench_$c.json
  rm -rf gpurun_out/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/prof_$c.err; echo "$c prof rc=$?"
  f=$(find gpurun_out/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r03_${c}_kernel_stats.csv
  rm -rf gpurun_out/prof_$c
done
