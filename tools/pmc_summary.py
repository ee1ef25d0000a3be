"""gpurun_out/pmc_{fetch,write,sq,l2}/ (tools/gpu_pmc.sh) -> profiles/<name> in the format bench.py's _pmc_traffic reads.
    python tools/pmc_summary.py profiles/r02_pmc_conv_b128.txt"""
import collections
import csv
import glob
import sys

out = [
    "# rocprofv3 --pmc passes (separate runs) over tools/pmc_conv.py, PMC_B=128 (bench batch), MI355X; tools/gpu_pmc.sh, the shipped kernels",
    "# FETCH_SIZE / WRITE_SIZE are KB; gfx950 reports half the bytes of 16-B/lane reads (MI355X_MICROARCH.md, HBM): read bytes = 2 x FETCH_SIZE",
    "# SQ_* cycle counters are summed over all waves, in quad-cycles; lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE",
    "# L2 pass: TCC_REQ_sum = requests the CUs sent to L2 (one <=128-B line each): the L2->LDS/VGPR traffic, weights re-staged per workgroup included",
]
for p in ("fetch", "write", "sq", "l2"):
    for f in glob.glob(f"gpurun_out/pmc_{p}/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in sorted(agg.items()):
            if "conv_" not in k:
                continue
            vals = {c: round(sum(v) / len(v), 1) for c, v in d.items()}
            line = f"{p} | {k} | {vals} | launches {len(next(iter(d.values())))}"
            if p == "sq" and vals.get("SQ_LDS_IDX_ACTIVE"):
                line += f" | lds_conflict_share {vals['SQ_LDS_BANK_CONFLICT'] / vals['SQ_LDS_IDX_ACTIVE']:.4f}"
            if p == "l2":
                hit = vals["TCC_HIT_sum"] / (vals["TCC_HIT_sum"] + vals["TCC_MISS_sum"])
                line += (f" | l2_hit_rate {hit:.3f} | l2_to_cu_bytes <= {vals['TCC_REQ_sum'] * 128 / 1e6:.1f} MB (128-B lines), "
                         f">= {vals['TCC_REQ_sum'] * 64 / 1e6:.1f} MB (64-B)")
            out.append(line)
open(sys.argv[1], "w").write("\n".join(out) + "\n")
print("\n".join(out[4:]))
