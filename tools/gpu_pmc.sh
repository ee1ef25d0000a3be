#!/bin/bash
# PMC passes over the conv kernels at one batch size (default: the bench batch, 128); counters in separate passes
# (FETCH_SIZE and WRITE_SIZE do not fit one pass on gfx950).  Results: gpurun_out/pmc_<pass>/
mkdir -p gpurun_out
export TMPDIR=/tmp
export PMC_B=${PMC_B:-128}
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_fetch -- python3 $R/tools/pmc_conv.py > $R/gpurun_out/pmc_fetch.log 2>&1; echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_write -- python3 $R/tools/pmc_conv.py > $R/gpurun_out/pmc_write.log 2>&1; echo "write rc=$?"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq -- python3 $R/tools/pmc_conv.py > $R/gpurun_out/pmc_sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $R/gpurun_out/pmc_l2 -- python3 $R/tools/pmc_conv.py > $R/gpurun_out/pmc_l2.log 2>&1; echo "l2 rc=$?"
cd $R
python3 - <<'PY'
import csv, glob, collections
for p in ("fetch", "write", "sq", "l2"):
    for f in glob.glob(f"gpurun_out/pmc_{p}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            if "conv" not in k: continue
            print(p, k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
