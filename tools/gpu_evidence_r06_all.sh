#!/bin/bash
# everything the round-6 docs cite, on one box: GPU suite, bench lines + kernel statistics of every workload, step trace,
# PMC passes (conv kernels at batch 128, whole step at batch 128 / 1000), split-bf16 gather probe
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
bash tools/gpu_tests_all.sh > gpurun_out/r06_suite.log 2>&1; tail -3 gpurun_out/r06_suite.log
CFGS="cfg1 cfg3 cfg3_elbo cfg4 cfg5 mnistsvhn cdsprites_shipped" bash tools/gpu_evidence_r06.sh > gpurun_out/r06_evidence.log 2>&1; grep -c "rc=0" gpurun_out/r06_evidence.log
python bench.py --config cdsprites_shipped --batch 128 --steps 30 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r06_bench_cdsprites_shipped_b128.json 2>/dev/null
python tools/probe/t3_time.py 128 512 1000 > gpurun_out/r06_convT3_alone.txt 2>&1
bash tools/gpu_pmc.sh > gpurun_out/r06_pmc_conv_b128.txt 2>&1
PMC_CFG=cfg2 PMC_B=128 bash tools/gpu_pmc_step.sh b128 > /dev/null 2>&1
PMC_CFG=cfg2 PMC_B=1000 bash tools/gpu_pmc_step.sh b1000 > /dev/null 2>&1
ls -la gpurun_out/pmcstep_b128_summary.txt gpurun_out/pmcstep_b1000_summary.txt
