#!/bin/bash
# PMC passes over the fused feed-forward kernels (tools/probe/ffn_time.py): VALU vs MFMA issue
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAVES SQ_INSTS_VALU --kernel-trace --output-format csv -d $R/gpurun_out/pmcffn_sq -- python3 $R/tools/probe/ffn_time.py > $R/gpurun_out/pmcffn_sq.log 2>&1; echo "sq rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $R/gpurun_out/pmcffn_sq2 -- python3 $R/tools/probe/ffn_time.py > $R/gpurun_out/pmcffn_sq2.log 2>&1; echo "sq2 rc=$?"
cd $R
python3 - <<'PY'
import csv, glob, collections
for p in ("sq", "sq2"):
    for f in glob.glob(f"gpurun_out/pmcffn_{p}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in agg.items():
            if "ffn32" not in k: continue
            print(p, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
