#!/bin/bash
# Round-6 evidence on the shipped kernels (copy gpurun_out/r06_* into profiles/ afterwards):
#   r06_bench.json                       the default bench line (cfg2, B=128, extras, cpu_baseline)
#   r06_cfg2_b{128,512,1000}_kernel_stats.csv   rocprofv3 --kernel-trace --stats of the same command at that batch
#   r06_step_trace.txt                   start times of the step's kernels inside the real captured step (trace build)
#   r06_bench_<cfg>.json + r06_<cfg>_kernel_stats.csv   the other BASELINE workloads (CFGS=... to choose)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
mkdir -p gpurun_out; export TMPDIR=/tmp
python3 bench.py --steps 200 --warmup 20 > gpurun_out/r06_bench.json 2> gpurun_out/r06_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r06_bench.json
for B in ${BATCHES:-128 512 1000}; do
  rm -rf gpurun_out/prof_b$B
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b$B -- python3 bench.py --batch $B --steps 30 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/prof_b$B.err; echo "prof B=$B rc=$?"
  f=$(find gpurun_out/prof_b$B -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_cfg2_b${B}_kernel_stats.csv
  rm -rf gpurun_out/prof_b$B
done
[ -f tools/probe/libmmvae_trace.so ] && MMVAE_HIP_LIB=$PWD/tools/probe/libmmvae_trace.so python3 tools/probe/trace_step.py > gpurun_out/r06_step_trace.txt 2>/dev/null; cat gpurun_out/r06_step_trace.txt
for c in ${CFGS:-}; do
  python3 bench.py --config $c --steps 50 --warmup 10 > gpurun_out/r06_bench_$c.json 2> gpurun_out/r06_bench_$c.err; echo "$c bench rc=$?"
  cut -c1-260 gpurun_out/r06_bench_$c.json
  rm -rf gpurun_out/prof_$c
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-extras > /dev/null 2> gpurun_out/prof_$c.err; echo "$c prof rc=$?"
  f=$(find gpurun_out/prof_$c -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r06_${c}_kernel_stats.csv
  rm -rf gpurun_out/prof_$c
done
