mkdir -p gpurun_out; export TMPDIR=/tmp
python3 bench.py --steps 200 --warmup 20 > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; echo "bench rc=$?"; cut -c1-600 gpurun_out/r03a_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03a_prof -- python3 bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/r03a_prof.json 2> gpurun_out/r03a_prof.err; echo "prof rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03a_prof1000 -- python3 bench.py --batch 1000 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r03a_prof1000.json 2> gpurun_out/r03a_prof1000.err; echo "prof1000 rc=$?"
PMC_CFG=cfg2 bash tools/gpu_pmc_step.sh b128
PMC_CFG=cfg2 PMC_B=1000 bash tools/gpu_pmc_step.sh b1000 > /dev/null
