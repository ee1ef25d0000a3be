#!/bin/bash
# PMC passes over the kernels of one eager training step (tools/pmc_step.py); counters in separate passes.
# usage (GPU box): PMC_CFG=cfg2 PMC_B=128 bash tools/gpu_pmc_step.sh <tag>    -> gpurun_out/pmcstep_<tag>_<pass>/
TAG=${1:-step}
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcstep_${TAG}_fetch -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmcstep_${TAG}_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmcstep_${TAG}_write -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmcstep_${TAG}_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmcstep_${TAG}_sq -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmcstep_${TAG}_sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/pmcstep_${TAG}_sq2 -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmcstep_${TAG}_sq2.log 2>&1; echo "sq2 rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmcstep_${TAG}_l2 -- python3 $R/tools/pmc_step.py > $R/gpurun_out/pmcstep_${TAG}_l2.log 2>&1; echo "l2 rc=$?"
cd $R
python3 tools/pmc_step_summary.py $TAG gpurun_out/pmcstep_${TAG}_summary.txt
